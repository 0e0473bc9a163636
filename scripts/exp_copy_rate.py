import torch, time
dev="cuda:0"
for mb in (320, 640):
    n = mb*1024*1024//4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f"copy {mb} MB: {ms*1e3:.1f} us, {2*mb*1.048576/ms:.0f} GB/s read+write")
    e0.record()
    for _ in range(20): b.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f"fill {mb} MB: {ms*1e3:.1f} us, {mb*1.048576/ms:.0f} GB/s write")
    e0.record()
    for _ in range(20): s = a.sum()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f"sum  {mb} MB: {ms*1e3:.1f} us, {mb*1.048576/ms:.0f} GB/s read")

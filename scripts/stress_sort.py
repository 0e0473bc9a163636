"""Many sorts of random sizes and key distributions against torch's stable sort, and the sort's error word
(a look-back that gave up) after each: a soak of the decoupled look-back, which leans on workgroups
starting in index order. Prints one line per 50 sorts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import _lib

dev = "cuda:0"
l = _lib.lib()
gen = torch.Generator(device=dev); gen.manual_seed(1234)
rng = np.random.default_rng(99)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
# "load": other kernels of this process keep the GPU busy on a second stream while the sorts run
load = len(sys.argv) > 2 and sys.argv[2] == "load"
side = torch.cuda.Stream(dev) if load else None
ballast = torch.rand(64 << 20, device=dev) if load else None
bad = 0
t0 = time.time()
for it in range(rounds):
    n = int(rng.choice([rng.integers(1, 5000), rng.integers(5000, 300_000), rng.integers(300_000, 12_000_000)]))
    kind = it % 3
    if kind == 0:
        keys = torch.randint(0, 1 << 31, (n,), device=dev, generator=gen, dtype=torch.int64)
    elif kind == 1:
        keys = torch.randint(0, 5, (n,), device=dev, generator=gen, dtype=torch.int64) * 0x01010101
    else:
        keys = torch.randint(0, 1 << 30, (n,), device=dev, generator=gen, dtype=torch.int64)
    ka = keys.to(torch.int32).contiguous()
    want_k, want_v = torch.sort(keys, stable=True)
    kb, va, vb = torch.empty_like(ka), torch.empty_like(ka), torch.empty_like(ka)
    ws = torch.zeros(int(l.kk_sort_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    if load:
        with torch.cuda.stream(side):
            for _ in range(3):
                ballast.mul_(1.0000001).add_(1e-9)
    _lib.check(l.kk_sort_pairs_u32(ka.data_ptr(), kb.data_ptr(), va.data_ptr(), vb.data_ptr(), 1, n,
                                   ws.data_ptr(), ws.numel(), _lib.current_stream_ptr()), "sort")
    torch.cuda.synchronize()
    err = int(ws[int(l.kk_sort_workspace_error_offset(n)):][:4].view(torch.int32)[0])
    ok = torch.equal(ka.to(torch.int64) & 0xFFFFFFFF, want_k & 0xFFFFFFFF) and torch.equal(va.to(torch.int64), want_v)
    if err or not ok:
        bad += 1
        print(f"sort {it}: n = {n}, kind {kind}, error word {err}, equal {ok}", flush=True)
    if it % 50 == 49:
        print(f"{it + 1} sorts, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)

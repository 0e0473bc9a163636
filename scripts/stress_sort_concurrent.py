"""Two radix sorts at a time on two streams (the case in which a look-back that leans on the start order
of workgroups could starve: both grids fill the GPU with tiles that wait for tiles in front). Every sort's
error word and result is checked. usage: stress_sort_concurrent.py [rounds] [plain|ticketed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import _lib

dev = "cuda:0"
l = _lib.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
l.kk_sort_set_ticketed(1 if mode == "ticketed" else 0)
gen = torch.Generator(device=dev); gen.manual_seed(7)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
sizes = [10_000_000, 7_300_001]
keys = [torch.randint(0, 1 << 30, (n,), device=dev, generator=gen, dtype=torch.int64) for n in sizes]
want = [torch.sort(k, stable=True) for k in keys]
bufs = []
for n, k in zip(sizes, keys):
    ka = k.to(torch.int32).contiguous()
    bufs.append({"src": ka, "ka": ka.clone(), "kb": torch.empty_like(ka), "va": torch.empty_like(ka),
                 "vb": torch.empty_like(ka),
                 "ws": torch.zeros(int(l.kk_sort_workspace_bytes(n)), dtype=torch.uint8, device=dev)})
torch.cuda.synchronize()
bad = 0
t0 = time.time()
for it in range(rounds):
    for s in range(2):
        b, n = bufs[s], sizes[s]
        with torch.cuda.stream(streams[s]):
            b["ka"].copy_(b["src"])
            _lib.check(l.kk_sort_pairs_u32(b["ka"].data_ptr(), b["kb"].data_ptr(), b["va"].data_ptr(),
                                           b["vb"].data_ptr(), 1, n, b["ws"].data_ptr(), b["ws"].numel(),
                                           int(streams[s].cuda_stream)), "sort")
    torch.cuda.synchronize()
    for s in range(2):
        b, n = bufs[s], sizes[s]
        err = int(b["ws"][int(l.kk_sort_workspace_error_offset(n)):][:4].view(torch.int32)[0])
        ok = torch.equal(b["ka"].to(torch.int64), want[s][0]) and torch.equal(b["va"].to(torch.int64), want[s][1])
        if err or not ok:
            bad += 1
            print(f"round {it} stream {s}: error word {err}, equal {ok}", flush=True)
    if it % 25 == 24:
        print(f"{mode}: {it + 1} rounds of two sorts side by side, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print(f"{mode}: bad {bad}, {(time.time() - t0) / rounds * 1e3:.2f} ms per round (with the checks)")
sys.exit(1 if bad else 0)

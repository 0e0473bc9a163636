"""Per-kernel timeline of one LBVH build + trace-side repacking of the 10 M-triangle soup (run under rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.lbvh import LBVH
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pos, tri = syn.triangle_soup(N)
bvh = LBVH("cuda:0")
bvh.BuildFrom(torch.from_numpy(pos), torch.from_numpy(tri.view(np.int32)))
for _ in range(3):
    bvh.rebuild()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    bvh.rebuild()
e1.record(); torch.cuda.synchronize()
print("build ms:", e0.elapsed_time(e1) / 10)
bvh.trace_accel(); bvh.trace_accel(rebuild=True)
torch.cuda.synchronize()

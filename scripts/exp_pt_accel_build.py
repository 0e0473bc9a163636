"""Trace-side repacking of veach-mis (the scene with run nodes): time per build; run under rocprofv3
--kernel-trace --stats for the per-kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd.data.scene_setup import make_tracer

d = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "veach_mis.npz")))
tracer, arr, c, proj = make_tracer(d, 64, 64, "cuda:0")
bvh = tracer.lbvh
for _ in range(3):
    bvh.trace_accel(rebuild=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    bvh.trace_accel(rebuild=True)
e1.record()
torch.cuda.synchronize()
print("veach-mis trace_accel build: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))

"""Profiling driver: LBVH build of the 10 M-triangle soup + 2 M incoherent closest-hit rays."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.lbvh import LBVH
from raytracer_amd.ray_tracer import closest_hit

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
r = int(sys.argv[2]) if len(sys.argv) > 2 else 2_097_152
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
pos, tri = syn.triangle_soup(n)
rays = torch.from_numpy(syn.incoherent_rays(r)).cuda()
bvh = LBVH("cuda:0")
bvh.BuildFrom(torch.from_numpy(pos), torch.from_numpy(tri.view(np.int32)))
print("M", bvh.GetDeviceData()["compactedFaceNum"])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(reps):
    ev[0].record(); bvh.rebuild(); ev[1].record(); torch.cuda.synchronize()
    print("build ms", ev[0].elapsed_time(ev[1]))
out = closest_hit(bvh, rays, count=True)
print(out[3], "hit frac", float((out[2] >= 0).float().mean()))
for i in range(reps):
    ev[0].record(); closest_hit(bvh, rays); ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]); print("trace ms", ms, "Mrays/s", r / ms / 1e3)

"""Build time of the 10 M-triangle soup (median of 20 rebuilds) and a byte check against a fresh build."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import _lib
if len(sys.argv) > 2 and sys.argv[2] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from raytracer_amd import synthetic as syn
from raytracer_amd.lbvh import LBVH
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pos, tri = syn.triangle_soup(N)
bvh = LBVH("cuda:0")
bvh.BuildFrom(torch.from_numpy(pos), torch.from_numpy(tri.view(np.int32)))
assert bvh.IsComplete()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
times = []
for i in range(25):
    ev[0].record(); bvh.rebuild(); ev[1].record(); torch.cuda.synchronize()
    if i >= 5:
        times.append(ev[0].elapsed_time(ev[1]))
print("build ms median %.4f min %.4f" % (float(np.median(times)), min(times)), flush=True)
h = bvh.to_host()
for k in ("faces", "compactedFaces", "internalNodes", "leafAABBs", "uniqueMortons", "sortedMortons"):
    print(k, hashlib.sha1(h[k].tobytes()).hexdigest()[:16])

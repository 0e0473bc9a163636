"""Distribution of used slots per wide node (200 K-triangle soup), per tree level."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.lbvh import LBVH
import test_gpu_trace_accel as T
pos, tri = syn.triangle_soup(2_000_000)
bvh = LBVH("cuda:0"); bvh.BuildFrom(pos, tri)
wide, tris, parents, nodes, N, M = T._read(bvh)
imask = wide["imask"].astype(np.uint32); leaf = wide["leafMask"] & 0xFF
used = np.array([bin(int(x)).count("1") for x in (imask | leaf)])
ni = np.array([bin(int(x)).count("1") for x in imask])
print("wide nodes", len(wide), "mean used", used.mean())
print("hist used slots:", np.bincount(used, minlength=9) / len(used))
print("nodes with no subtree slot:", (ni == 0).mean(), " their used hist:", np.bincount(used[ni == 0], minlength=9) / max((ni == 0).sum(), 1))
print("nodes with subtree slots: used hist:", np.bincount(used[ni > 0], minlength=9) / max((ni > 0).sum(), 1))

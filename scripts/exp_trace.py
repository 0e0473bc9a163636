"""Times the closest-hit variants against each other on the benchmark workload and checks that they agree."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.lbvh import LBVH
from raytracer_amd.ray_tracer import closest_hit

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2_097_152
dev = "cuda:0"
pos, tri = syn.triangle_soup(N)
bvh = LBVH(dev)
bvh.BuildFrom(torch.from_numpy(pos), torch.from_numpy(tri.view(np.int32)))
rays = torch.from_numpy(syn.incoherent_rays(R)).to(dev)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

ref = closest_hit(bvh, rays, layout="build", count=True)
print("build-layout: %.3f ms" % timeit(lambda: closest_hit(bvh, rays, layout="build")), ref[3], flush=True)
print("accel build: %.3f ms" % timeit(lambda: bvh.trace_accel(rebuild=True)), flush=True)
for sort in (False, True):
    got = closest_hit(bvh, rays, layout="trace", sort=sort, count=True)
    ms = timeit(lambda: closest_hit(bvh, rays, layout="trace", sort=sort))
    same_p = bool((got[2] == ref[2]).all()); same_t = bool((got[0] == ref[0]).all()); same_b = bool((got[1] == ref[1]).all())
    print(f"trace-layout sort={sort}: {ms:.3f} ms  {R/ms/1e3:.1f} Mrays/s  prim_eq={same_p} t_eq={same_t} bary_eq={same_b}",
          {k: (round(v / R, 2) if k != 'dbg' else v) for k, v in got[3].items()}, flush=True)
from raytracer_amd import _lib
l = _lib.lib()
ws = torch.zeros(l.kk_closest_hit_accel_workspace_bytes(R), dtype=torch.uint8, device=dev)
t = torch.empty(R, dtype=torch.float32, device=dev); b = torch.empty((R, 2), dtype=torch.float32, device=dev); pr = torch.empty(R, dtype=torch.int32, device=dev)
closest_hit(bvh, rays, out=(t, b, pr, ws)); torch.cuda.synchronize()
print("exact-list length:", int(ws[:256].view(torch.int32)[16]), "queue heads:", ws[:256].view(torch.int32)[[0, 32]].tolist())

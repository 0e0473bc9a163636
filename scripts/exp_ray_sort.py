"""What the ray ordering (keys + sort) costs beside the trace kernel, and the LBVH build time, for a given
build of the library: python scripts/exp_ray_sort.py [path/to/libkouek_hip.so ...] (alternating, three rounds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess

if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for rnd in range(3):
        for lib in sys.argv[1:]:
            out = subprocess.run([sys.executable, __file__, "--one", lib], capture_output=True, text=True)
            print(lib, out.stdout.strip() or out.stderr[-400:], flush=True)
    sys.exit(0)
from raytracer_amd import _lib
if len(sys.argv) > 2 and sys.argv[2] != "default":
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.lbvh import LBVH
from raytracer_amd.ray_tracer import closest_hit

dev = "cuda:0"
R = 2_097_152
pos, tri = syn.triangle_soup(10_000_000)
bvh = LBVH(dev)
bvh.BuildFrom(torch.from_numpy(pos), torch.from_numpy(tri.view(np.int32)))
rays = torch.from_numpy(syn.incoherent_rays(R)).to(dev)
bvh.trace_accel()
t, b, p = closest_hit(bvh, rays)
ws = torch.empty(_lib.lib().kk_closest_hit_accel_workspace_bytes(R), dtype=torch.uint8, device=dev)
out = (t, b, p, ws)


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


step = timeit(lambda: closest_hit(bvh, rays, out=out))
kern = timeit(lambda: closest_hit(bvh, rays, out=out, reuse_order=True))
build = timeit(lambda: bvh.rebuild(), 20)
print(f"step {step:.4f} ms, kernel {kern:.4f}, order {1e3 * (step - kern):.1f} us, build {build:.4f} ms")

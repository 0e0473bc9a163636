"""Closest hit on the benchmark workload: time, work counters and per-wave utilisation figures."""
import sys, os
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


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("accel build: %.3f ms" % timeit(lambda: bvh.trace_accel(rebuild=True), 3), flush=True)
hdr = bvh.trace_accel()[:64].view(torch.int32).cpu().numpy()
print("wide nodes:", int(hdr[10]), "of", int(hdr[7]), "binary nodes")
got = closest_hit(bvh, rays, count=True)
c = got[3]
print({k: round(v / R, 2) for k, v in c.items() if not k.endswith("_us")})
for _ in range(2):
    c2 = closest_hit(bvh, rays, count=True)[3]
    print(f"counting launch again: {c2['launch_us']:.0f} us, queues used up {c2['tail_us']:.0f} us before the last exit, "
          f"mean life of a wave {c2['wave_life_sum_us'] / 5120:.0f} us")
print(f"counting launch: {c['launch_us']:.0f} us from first start to last exit; the first wave found the queues used up "
      f"{c['tail_us']:.0f} us before the last exit; mean life of a wave {c['wave_life_sum_us'] / 5120:.0f} us (5120 waves)")
it = max(c["wave_passes"], 1)
print(f"visit lanes/pass {c['visit_lanes'] / it:.1f}, tri phases/pass {c['tri_phases'] / it:.3f}, "
      f"lanes/tri phase {c['tri_lanes'] / max(c['tri_phases'], 1):.1f}")
ms = timeit(lambda: closest_hit(bvh, rays))
ms2 = timeit(lambda: closest_hit(bvh, rays, reuse_order=True))
print(f"sorted: {ms:.3f} ms {R / ms / 1e3:.1f} Mrays/s; kernel only {ms2:.3f} ms {R / ms2 / 1e3:.1f} Mrays/s")
ms3 = timeit(lambda: closest_hit(bvh, rays, sort=False))
print(f"unsorted: {ms3:.3f} ms {R / ms3 / 1e3:.1f} Mrays/s")

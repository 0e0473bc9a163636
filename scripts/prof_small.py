"""Short profiling driver (for rocprofv3 --pmc passes): one LBVH build of the 10 M-triangle soup, three
closest-hit launches of 2 M rays, and optionally DVR frames of the Kingsnake-shape volume at 4K ("dvr"), of the Head-shape volume at 1080p ("dvr_head") or path-tracer sample passes of
veach-mis at 1080p ("pt")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.lbvh import LBVH
from raytracer_amd.ray_tracer import closest_hit

what = sys.argv[1] if len(sys.argv) > 1 else "path1"
if what in ("path1", "all"):
    pos, tri = syn.triangle_soup(10_000_000)
    rays = torch.from_numpy(syn.incoherent_rays(2_097_152)).cuda()
    bvh = LBVH("cuda:0")
    bvh.BuildFrom(torch.from_numpy(pos), torch.from_numpy(tri.view(np.int32)))
    bvh.rebuild()
    torch.cuda.synchronize()
    for _ in range(3):
        closest_hit(bvh, rays)
    torch.cuda.synchronize()
if what in ("dvr", "all"):
    from benchlib import bench_dvr
    print(bench_dvr.run("cuda:0", 1, 0, None, frames=4, warmup=2, with_cpu=False))
if what in ("dvr_head",):
    from benchlib import bench_dvr
    print(bench_dvr.run("cuda:0", 1, 0, None, frames=4, warmup=2, with_cpu=False, workload="head",
                        dim=(128, 256, 256), size=(1920, 1080)))
if what in ("pt",):
    from benchlib import bench_pt
    print(bench_pt.run("cuda:0", spp=4, frames=1, with_cpu=False))

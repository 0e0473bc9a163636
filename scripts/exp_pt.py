"""Times the path-tracing leg alone (veach-mis 1920x1080, 16 spp) and prints its parity figures."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib import bench_pt  # noqa: E402

if __name__ == "__main__":
    import ctypes
    import torch
    from raytracer_amd import _lib
    with_cpu = "--cpu" in sys.argv
    print(json.dumps(bench_pt.run("cuda:0", with_cpu=with_cpu)))
    if "--count" in sys.argv:
        c = torch.zeros(16, dtype=torch.int64, device="cuda:0")
        _lib.lib().kk_pt_set_work_counters(ctypes.c_void_p(c.data_ptr()))
        bench_pt.run("cuda:0", with_cpu=False, frames=1)
        torch.cuda.synchronize()
        _lib.lib().kk_pt_set_work_counters(None)
        v = [int(x) for x in c.cpu()]
        spp = 80  # bench_pt.run renders five frames of 16 samples: two per timing form and a last one
        print(json.dumps({"walk_passes_per_wave_sum": v[0], "lanes_per_walk_pass": round(v[1] / max(v[0], 1), 1),
                          "shade_phases": v[2], "lanes_per_shade": round(v[3] / max(v[2], 1), 1),
                          "rays_per_path": round(v[4] / (1920 * 1080 * spp), 2),
                          "visits_per_ray": round(v[5] / max(v[4], 1), 2),
                          "tris_per_ray": round(v[6] / max(v[4], 1), 2),
                          "tris_of_hit_slots_per_ray": round(v[14] / max(v[4], 1), 2),
                          "tri_rounds_per_walk_pass": round(v[7] / max(v[0], 1), 2),
                          "pairs_per_tri_round": round(v[15] / max(v[7], 1), 1),
                          "walk_passes_per_shade": round(v[0] / max(v[2], 1), 1),
                          "cycles_frac": {k: round(v[8 + i] / max(sum(v[8:14]), 1), 3) for i, k in
                                          enumerate(("walk_loop", "visit", "tri", "lights", "bounce", "end"))},
                          "cycles_per_walk_pass": round(v[9] / max(v[0], 1), 1),
                          "cycles_per_shade_section": round(sum(v[11:14]) / max(v[2], 1), 1)}))

"""Marcher timing only (Kingsnake 4K and Head 1080p orbits: fps with the tile order, in launch order, with two
frames in flight), without the CPU baseline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib import bench_dvr

for wl, kw in (("kingsnake", {}), ("head", {"dim": (128, 256, 256), "size": (1920, 1080), "cpu_rows": 16})):
    r = bench_dvr.run("cuda:0", 1, 0, with_cpu=False, workload=wl, **kw)
    print(wl, {k: r[k] for k in ("fps", "fps_launch_order", "fps_two_frames_in_flight") if k in r}, flush=True)

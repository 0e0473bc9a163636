"""DVR frame times on the two benchmark volumes (no CPU legs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib import bench_dvr
r = bench_dvr.run("cuda:0", 1, 0, None, with_cpu=False)
print({k: r[k] for k in ("fps", "ms_per_frame", "Gsamples_per_s", "samples_per_frame", "gpu_ms_per_frame")})
r = bench_dvr.run("cuda:0", 1, 0, None, with_cpu=False, workload="head", dim=(128, 256, 256), size=(1920, 1080))
print({k: r[k] for k in ("fps", "ms_per_frame", "Gsamples_per_s", "samples_per_frame", "gpu_ms_per_frame")})

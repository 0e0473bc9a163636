"""Path tracer timing only (veach-mis 1080p, 16 spp as 16 calls and as one call), without the CPU baseline."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib import bench_pt  # noqa: E402

r = bench_pt.run("cuda:0", frames=5, with_cpu=False)
print(json.dumps({k: r[k] for k in ("ms_per_frame", "ms_per_sample_pass", "ms_per_frame_one_call")}))

import sys, json
sys.path.insert(0, "/root/repo")
from benchlib import bench_pt
r = bench_pt.run("cuda:0", frames=5, with_cpu=False)
print(json.dumps({k: r[k] for k in ("ms_per_frame", "ms_per_sample_pass", "ms_per_frame_one_call")}))

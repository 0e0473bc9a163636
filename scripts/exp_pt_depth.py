"""Path tracer on veach-mis: time per 16-spp frame against the maximum path depth."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd.data.scene_setup import make_tracer
from raytracer_amd.ray_tracer import FrameBuffer

d = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/veach_mis.npz")))
W, H, spp = 1920, 1080, 16
print("faces", d["face_position_indices"].shape[0])
for depth in (1, 2, 3, 4):
    tracer, arr, c, proj = make_tracer(d, W, H, "cuda:0", max_path_depth=depth)
    fb = FrameBuffer(W, H, "cuda:0")
    def frame():
        tracer.curr_accu_spp = 1.0
        for _ in range(spp):
            tracer.Render(fb)
    frame(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        frame()
    e1.record(); torch.cuda.synchronize()
    print(f"depth {depth}: {e0.elapsed_time(e1) / 3:.2f} ms", flush=True)

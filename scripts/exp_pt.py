"""Path tracer on veach-mis: the 8-wide tree against the binary tree (time, identical accumulation)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd.data.scene_setup import make_tracer
from raytracer_amd.ray_tracer import FrameBuffer

d = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/veach_mis.npz")))
W, H, spp = 1920, 1080, 16
res = {}
for wide in (False, True):
    tracer, arr, c, proj = make_tracer(d, W, H, "cuda:0")
    tracer.use_trace_tree = wide
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
    ms = e0.elapsed_time(e1) / 3
    res[wide] = tracer.d_accum.clone()
    print(f"wide={wide}: {ms:.2f} ms per {spp}-spp frame, {W * H * spp / ms / 1e3:.1f} Mpaths/s", flush=True)
diff = (res[True] - res[False]).abs()
print("max abs diff", float(diff.max()), "pixels differing", int((diff.max(dim=2).values > 0).sum()))

"""One GPU renders rank r's share of the path-traced 1080p frame (round-robin 16-row stripes of N ranks,
compact, 16 calls of one sample each) for N = 1, 2, 4, 8: what a rank of an N-GPU job takes per frame
without the gather. The slowest rank's time is what the job's frame takes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd.data.scene_setup import make_tracer
from raytracer_amd.ray_tracer import FrameBuffer
from raytracer_amd import _lib

dev = "cuda:0"
d = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "veach_mis.npz")))
W, H, spp = 1920, 1080, 16
tracer, arr, c, proj = make_tracer(d, W, H, dev)
fb = FrameBuffer(W, H, dev)
for n in (1, 2, 4, 8):
    per = int(_lib.lib().kk_stripes_per_rank(H, n))
    buf = torch.zeros((per, 16, W, 4), dtype=torch.uint8, device=dev)
    times, once = [], []
    for r in range(n):
        def frame(one_call=False):
            tracer.curr_accu_spp = 1.0
            if tracer.d_accum is not None:
                tracer.d_accum.zero_()
            if one_call:
                tracer.Render(fb, stripes=(r, n), compact_out=buf, samples=spp)
                return
            for _ in range(spp):
                tracer.Render(fb, stripes=(r, n), compact_out=buf)
        frame(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            frame()
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 3)
        frame(True); torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            frame(True)
        e1.record(); torch.cuda.synchronize()
        once.append(e0.elapsed_time(e1) / 3)
    print(f"N={n}: ms per 16-spp frame of each rank's share: " + " ".join(f"{t:.2f}" for t in times) +
          f"  -> slowest {max(times):.2f} ms; as one call of {spp} samples: slowest {max(once):.2f} ms", flush=True)

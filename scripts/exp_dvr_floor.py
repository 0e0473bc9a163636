"""What a DVR frame costs when no ray meets the volume (the camera turned away): the floor that launching
32,400 workgroups and setting up their rays puts under a 4K frame, beside the frame itself in launch order
and in the adaptive tile order. The set-up is the first part of exp_dvr_inflight.py."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
exec(open(os.path.join(HERE, "exp_dvr_inflight.py")).read().split("for k in (1, 2, 3):")[0])
import numpy as np
fb = FrameBuffer(W, H, dev)
away = np.diag([-1.0, 1.0, -1.0]).astype(np.float32)
def timeit(turn, n=300):
    def frame(i):
        pos, rot = poses[i]
        caster.SetCameraToWorld(pos, (rot @ away) if turn else rot)
        caster.RenderDepthBoxVDB(fb)
    for i in range(8): frame(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(8, 8 + n): frame(i)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 4)
for adaptive in (False, True):
    caster.SetAdaptiveTileOrder(adaptive)
    print("adaptive" if adaptive else "launch order", "frame", timeit(False), "turned away", timeit(True),
          "lit pixels when turned away", int((fb.rgba[..., 3] != 0).sum()))

"""Where a DVR frame's time goes: the frame, row bands and subsets of its 16-row stripes timed alone.
The set-up is the first part of exp_dvr_inflight.py."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
exec(open(os.path.join(HERE, "exp_dvr_inflight.py")).read().split("for k in (1, 2, 3):")[0])
fb = FrameBuffer(W, H, dev)
def timeit(rows, stripes=None, n=200):
    def frame(i):
        caster.SetCameraToWorld(*poses[i])
        caster.RenderDepthBoxVDB(fb, rows=rows, stripes=stripes)
    for i in range(8): frame(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(8, 8 + n): frame(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("full", timeit(None))
for a, b in ((0, 540), (540, 1080), (1080, 1620), (1620, 2160), (800, 1360), (1000, 1160), (1072, 1088)):
    print("rows", a, b, round(timeit((a, b)), 4))
for st in ((0, 2), (0, 4), (0, 8), (0, 16), (0, 64)):
    print("stripes", st, round(timeit(None, st), 4))

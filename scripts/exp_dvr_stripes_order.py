"""One GPU's share of a 4K frame dealt out to N GPUs in 16-row stripes (rank 0's compact stripes), in launch
order and with the adaptive tile order: ms per frame over part of the orbit, the tile sort included.
The set-up is the first part of exp_dvr_inflight.py."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
exec(open(os.path.join(HERE, "exp_dvr_inflight.py")).read().split("for k in (1, 2, 3):")[0])
from raytracer_amd import _lib
fb = FrameBuffer(W, H, dev)
for world in (2, 4, 8):
    per = int(_lib.lib().kk_stripes_per_rank(H, world))
    buf = torch.zeros(per, 16, W, 4, dtype=torch.uint8, device=dev)
    out = []
    for adaptive in (False, True):
        caster.SetAdaptiveTileOrder(adaptive)
        def frame(i):
            caster.SetCameraToWorld(*poses[i])
            caster.RenderDepthBoxVDB(fb, stripes=(0, world), compact_out=buf)
        for i in range(8): frame(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(8, 308): frame(i)
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / 300 * 1e3, 4))
    print(f"{world} GPUs: rank 0's stripes {out[0]} ms in launch order, {out[1]} ms with the tile order", flush=True)

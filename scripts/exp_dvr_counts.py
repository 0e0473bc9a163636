"""Work counters of one 4K Kingsnake frame with a diagnostic build of the marcher (KK_LIB): counter 0 = samples,
counter 1 = what that build counts instead of brick entries (cell steps, depth steps, ...)."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from raytracer_amd import _lib
if os.environ.get("KK_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["KK_LIB"])
exec(open(os.path.join(HERE, "exp_dvr_inflight.py")).read().split("for k in (1, 2, 3):")[0])
fb = FrameBuffer(W, H, dev)
for i in (8, 180):
    counters = torch.zeros(2, dtype=torch.int64, device=dev)
    caster.SetCameraToWorld(*poses[i])
    caster.RenderDepthBoxVDB(fb, counters=counters)
    torch.cuda.synchronize()
    print(os.environ.get("KK_LIB", "default"), "pose", i, "samples", int(counters[0]), "counter 1:", int(counters[1]), flush=True)

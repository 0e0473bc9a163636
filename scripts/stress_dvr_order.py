"""Every frame of a stretch of the 4K Kingsnake orbit rendered in launch order and with the adaptive tile
order (and, every tenth frame, as rank 0's compact stripes of four GPUs in both ways): the bytes must be
equal frame by frame. The set-up is the first part of exp_dvr_inflight.py."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
exec(open(os.path.join(HERE, "exp_dvr_inflight.py")).read().split("for k in (1, 2, 3):")[0])
from raytracer_amd import _lib
from raytracer_amd.ray_caster import RayCaster
n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
plain, ordered = FrameBuffer(W, H, dev), FrameBuffer(W, H, dev)
other = RayCaster(dev)
other._params, other.vdb, other.tf = caster._params, caster.vdb, caster.tf
other.SetAdaptiveTileOrder(True)
per = int(_lib.lib().kk_stripes_per_rank(H, 4))
s_plain = torch.zeros(per, 16, W, 4, dtype=torch.uint8, device=dev)
s_ord = torch.zeros_like(s_plain)
stripes = RayCaster(dev)
stripes._params, stripes.vdb, stripes.tf = caster._params, caster.vdb, caster.tf
stripes.SetAdaptiveTileOrder(True)
bad = 0
for i in range(n):
    pose = poses[(3 * i) % len(poses)]
    caster.SetCameraToWorld(*pose)
    other.SetCameraToWorld(*pose)
    caster.RenderDepthBoxVDB(plain)
    other.RenderDepthBoxVDB(ordered)
    same = torch.equal(plain.rgba, ordered.rgba)
    if i % 10 == 0:
        stripes.SetCameraToWorld(*pose)
        caster.RenderDepthBoxVDB(plain, stripes=(0, 4), compact_out=s_plain)
        stripes.RenderDepthBoxVDB(plain, stripes=(0, 4), compact_out=s_ord)
        same = same and torch.equal(s_plain, s_ord)
    if not same:
        bad += 1
        print("frame", i, "differs", flush=True)
print(f"{n} frames, {bad} bad; lit pixels in the last frame: {int((plain.rgba[..., 3] != 0).sum())}")
sys.exit(1 if bad else 0)

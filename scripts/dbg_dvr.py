"""Wave-level iteration counts of the DVR kernel's loops (library built with -DKK_DBG_STEPS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.data import camera as cam
from raytracer_amd.data.transfer_function import README_KINGSNAKE_TF, TransferFunction
from raytracer_amd.dpbx_vdb import DepthBoxVDB, Volume
from raytracer_amd.ray_caster import FrameBuffer, RayCaster, TransferFunctionTable
DEV = "cuda:0"
tf = TransferFunction(README_KINGSNAKE_TF)
dim = (1024, 1024, 795)
vol = syn.kingsnake_volume_torch(dim, DEV)
vdb = DepthBoxVDB(DEV)
vdb.BuildFrom(Volume(vol, DEV), tf.empty_scalar_ranges, apron_wid=1, log2_dims=(5, 4, 3), use_depth_box=True)
scales = cam.volume_scales((1, 1, 1), dim)
W, H = 3840, 2160
caster = RayCaster(DEV)
caster.SetWorldToScene(cam.world_to_scene(scales)); caster.SetDepthBoxVDB(vdb)
caster.SetTransferFunctionTexture(TransferFunctionTable(tf.dat, DEV))
caster.SetProjection(cam.perspective_fov(cam.radians(90.0), W, H, 1.0, 3.0))
c = cam.FPSCamera((0, 0, float(max(scales))), (0, 0, 0))
for _ in range(9):
    c.Revolve(cam._c("sqrtf", cam._dot(c.pos, c.pos)), 0.5, 0.0)
caster.SetCameraToWorld(c.pos, c.rotation())
fb = FrameBuffer(W, H, DEV)
counters = torch.zeros(16, dtype=torch.int64, device=DEV)
caster.RenderDepthBoxVDB(fb, counters=counters); torch.cuda.synchronize()
cs = counters.tolist()
print("samples", cs[0], "bricks", cs[1])
for name, k in (("dda steps", 2), ("events", 4), ("leaf phases", 6), ("depth steps", 8), ("sample iters", 10)):
    print(f"{name}: wave passes {cs[k]}, lanes {cs[k+1]}, lanes/pass {cs[k+1]/max(cs[k],1):.1f}")
tot = cs[12]
print(f"wave clocks: kernel {tot}, empty runs {cs[13]/tot:.3f}, sample loops {cs[14]/tot:.3f}, leaf phases (incl. samples) {cs[15]/tot:.3f}")

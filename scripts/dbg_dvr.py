import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.data import camera as cam
from raytracer_amd.data.transfer_function import README_HEAD_TF, TransferFunction
from raytracer_amd.dpbx_vdb import DepthBoxVDB, Volume
from raytracer_amd.ray_caster import FrameBuffer, RayCaster, TransferFunctionTable
DEV="cuda:0"
tf = TransferFunction(README_HEAD_TF)
dim=(64,128,128)
vol = syn.head_volume(dim)
for udb in (False, True):
    vdb = DepthBoxVDB(DEV)
    vdb.BuildFrom(Volume(torch.from_numpy(vol).to(DEV), DEV), tf.empty_scalar_ranges, apron_wid=1, log2_dims=(4,3,3), use_depth_box=udb)
    print("complete", vdb.IsComplete(), "root_lev", vdb.plan.param.root_lev, "leafs", vdb.plan.leaf_num)
    scales = cam.volume_scales((2,1,1), dim)
    W,H=200,150
    caster = RayCaster(DEV)
    caster.SetWorldToScene(cam.world_to_scene(scales)); caster.SetDepthBoxVDB(vdb)
    caster.SetTransferFunctionTexture(TransferFunctionTable(tf.dat, DEV))
    caster.SetProjection(cam.perspective_fov(cam.radians(90.0), W, H, 1.0, 3.0))
    c = cam.FPSCamera((0,0,float(max(scales))),(0,0,0))
    caster.SetCameraToWorld(c.pos, c.rotation())
    fb = FrameBuffer(W,H,DEV)
    counters = torch.zeros(2, dtype=torch.int64, device=DEV)
    caster.RenderDepthBoxVDB(fb, counters=counters); torch.cuda.synchronize()
    print(udb, counters.tolist(), fb.ReadBack()[75,100])

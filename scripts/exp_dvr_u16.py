"""The marcher on 16-bit voxels: the Head-shape volume scaled to uint16 (value * 257), 1080p orbit, launch
order: fps of 300 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import synthetic as syn
from raytracer_amd.data import camera as cam
from raytracer_amd.dpbx_vdb import DepthBoxVDB, Volume
from raytracer_amd.ray_caster import FrameBuffer, RayCaster, TransferFunctionTable
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from vol_util import HEAD_TF, dvr_setup

dev = "cuda:0"
dim = (128, 256, 256)
W, H = 1920, 1080
vol = syn.head_volume(dim).astype(np.uint16) * 257
rng = (np.asarray(HEAD_TF.empty_scalar_ranges).astype(np.uint32) * 257).astype(np.uint16)
vdb = DepthBoxVDB(dev)
vdb.BuildFrom(Volume(vol, dev), rng, apron_wid=1, log2_dims=(5, 2, 2), use_depth_box=True)
assert vdb.IsComplete()
caster = RayCaster(dev)
caster.SetDepthBoxVDB(vdb)
caster.SetTransferFunctionTexture(TransferFunctionTable(HEAD_TF.dat, dev))
fb = FrameBuffer(W, H, dev)
poses = []
for i in range(308):
    w2s, c, proj = dvr_setup((256, 256, 128), (2, 1, 1), W, H, orbit_steps=i)
    poses.append((c.pos, c.rotation()))
caster.SetWorldToScene(w2s)
caster.SetProjection(proj)
for adaptive in (False, True):
    caster.SetAdaptiveTileOrder(adaptive)
    for i in range(8):
        caster.SetCameraToWorld(*poses[i]); caster.RenderDepthBoxVDB(fb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(8, 308):
        caster.SetCameraToWorld(*poses[i]); caster.RenderDepthBoxVDB(fb)
    torch.cuda.synchronize()
    print(("tile order" if adaptive else "launch order"), f"{300 / (time.perf_counter() - t0):.1f} fps; lit pixels {int((fb.rgba[..., 3] > 0).sum())}", flush=True)

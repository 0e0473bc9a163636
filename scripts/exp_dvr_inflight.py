"""DVR frame rate with one, two and three frames in flight (frames alternate over as many HIP streams and
framebuffers): what a double- or triple-buffered presentation loop would see."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import synthetic as syn
from raytracer_amd.data import camera as cam
from raytracer_amd.data.transfer_function import README_KINGSNAKE_TF, TransferFunction
from raytracer_amd.dpbx_vdb import DepthBoxVDB, Volume
from raytracer_amd.ray_caster import FrameBuffer, RayCaster, TransferFunctionTable

dev = "cuda:0"
dim, (W, H) = (1024, 1024, 795), (3840, 2160)
tf = TransferFunction(README_KINGSNAKE_TF)
vdb = DepthBoxVDB(dev)
vdb.BuildFrom(Volume(syn.kingsnake_volume_torch(dim, dev), dev), tf.empty_scalar_ranges, apron_wid=1,
              log2_dims=(5, 4, 3), use_depth_box=True)
scales = cam.volume_scales((1, 1, 1), dim)
caster = RayCaster(dev)
caster.SetWorldToScene(cam.world_to_scene(scales))
caster.SetDepthBoxVDB(vdb)
caster.SetTransferFunctionTexture(TransferFunctionTable(tf.dat, dev))
caster.SetProjection(cam.perspective_fov(cam.radians(90.0), W, H, 1.0, 3.0))
c = cam.FPSCamera((0, 0, float(max(scales))), (0, 0, 0))
poses = []
for _ in range(728):
    c.Revolve(cam._c("sqrtf", cam._dot(c.pos, c.pos)), 0.5, 0.0)
    poses.append((c.pos.copy(), c.rotation().copy()))
for k in (1, 2, 3):
    streams = [torch.cuda.Stream(dev) for _ in range(k)]
    fbs = [FrameBuffer(W, H, dev) for _ in range(k)]
    def frame(i):
        caster.SetCameraToWorld(*poses[i])
        with torch.cuda.stream(streams[i % k]):
            caster.RenderDepthBoxVDB(fbs[i % k])
    for i in range(8):
        frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8, 728):
        frame(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{k} in flight: {720 / dt:.1f} fps, {dt / 720 * 1e3:.4f} ms per frame", flush=True)

# one frame as two launches on two streams: even / odd 16-row stripes
streams = [torch.cuda.Stream(dev) for _ in range(2)]
fb = FrameBuffer(W, H, dev)
for label, parts in (("even/odd stripes on two streams", [(0, 2), (1, 2)]),
                     ("four stripe sets on two streams", [(0, 4), (1, 4), (2, 4), (3, 4)]),
                     ("even/odd stripes on one stream", None)):
    def frame(i):
        caster.SetCameraToWorld(*poses[i])
        if parts is None:
            caster.RenderDepthBoxVDB(fb, stripes=(0, 2))
            caster.RenderDepthBoxVDB(fb, stripes=(1, 2))
            return
        ev = torch.cuda.Event()
        ev.record()
        for j, st in enumerate(parts):
            s = streams[j % 2]
            s.wait_event(ev)
            with torch.cuda.stream(s):
                caster.RenderDepthBoxVDB(fb, stripes=st)
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
    for i in range(8):
        frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8, 728):
        frame(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {720 / dt:.1f} fps, {dt / 720 * 1e3:.4f} ms per frame", flush=True)

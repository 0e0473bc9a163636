"""Path-tracing leg of the benchmark: the veach-mis scene (the reference's only complete scene, arrays
from tests/golden/veach_mis.npz) at 1920x1080, 16 calls of RayTracer.Render (16 samples per pixel,
maxPathDepth 4), timed with the scene resident; the CPU checker renders a few row bands of the same
frame for the baseline and the full-size parity figure."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from raytracer_amd.data import camera as cam
from raytracer_amd.data.scene_setup import make_tracer
from raytracer_amd.ray_tracer import FrameBuffer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(dev: str, size=(1920, 1080), spp: int = 16, frames: int = 5, with_cpu: bool = True,
        cpu_bands: int = 4, band_rows: int = 4) -> dict:
    path = os.path.join(ROOT, "tests", "golden", "veach_mis.npz")
    if not os.path.exists(path):
        return {}
    d = dict(np.load(path))
    W, H = size
    tracer, arr, c, proj = make_tracer(d, W, H, dev)
    fb = FrameBuffer(W, H, dev)

    def frame(batched=False):
        tracer.curr_accu_spp = 1.0  # restart the accumulation
        if batched:
            tracer.Render(fb, samples=spp)
        else:
            for _ in range(spp):
                tracer.Render(fb)

    def timed(batched):
        frame(batched)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(frames):
            frame(batched)
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / frames

    # the trace-side repacking of this scene's tree (it has run nodes, which the soup's has not)
    bvh = tracer.lbvh
    for _ in range(2):
        bvh.trace_accel(rebuild=True)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        bvh.trace_accel(rebuild=True)
    ev1.record()
    torch.cuda.synchronize()
    accel_ms = ev0.elapsed_time(ev1) / 10

    ms = timed(False)           # the reference's loop: one sample per Render call
    ms_batched = timed(True)    # the same image from one call (Render(fb, samples=spp))
    frame(False)
    out = {"workload": f"veach-mis {W}x{H}, {spp} spp, maxPathDepth 4",
           "ms_per_frame": round(ms, 3), "ms_per_sample_pass": round(ms / spp, 4),
           "ms_per_frame_one_call": round(ms_batched, 3),
           "trace_accel_build_ms": round(accel_ms, 4),
           "Mpaths_per_s": round(W * H * spp / (ms * 1e-3) / 1e6, 1)}
    # the path-tracing kernel is bound by instruction issue and the CUs' memory path, not by HBM: the frame
    # against the time its wave instructions need on all SIMDs (one dispatch of the profiled kernel is one
    # sample pass of the 1920 x 1080 frame)
    if (W, H) == (1920, 1080):
        from benchlib.roofline import valu_issue
        rl = valu_issue(["r03_pt_pmc.json", "r02_pt_pmc.json"], ms, dispatches_per_unit=spp,
                        kernel_contains="<false, 3,")
        if rl:
            rl["kernel"] = "path_trace_persistent_kernel<false, 3, false>"
            rl["frac_one_call"] = round(rl["peak"] / ms_batched, 4)
            out["roofline"] = rl
    if with_cpu:
        from oracle import kkoracle as ko
        cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
        ref = ko.lbvh_build(d["positions"], d["face_position_indices"])
        acc = np.zeros((H, W, 3), np.float32)
        img = np.zeros((H, W, 4), np.uint8)
        bands = [(int((i + 0.5) * H / cpu_bands), int((i + 0.5) * H / cpu_bands) + band_rows)
                 for i in range(cpu_bands)]
        inv_proj = cam.inverse_projective(proj)
        t0 = time.perf_counter()
        for s in range(1, spp + 1):
            for rows in bands:
                ko.pt_render_sample(ref, arr, np.eye(4, dtype=np.float32), inv_proj, c.pos,
                                    c.rotation(), acc, img, float(s), rows=rows)
        cpu_s = time.perf_counter() - t0
        n_rows = cpu_bands * band_rows
        got = tracer.d_accum.cpu().numpy()
        sel = np.concatenate([np.arange(a, b) for a, b in bands])
        err = np.abs(got[sel] - acc[sel]).max(axis=2)
        out["cpu_baseline"] = {"Mpaths_per_s": round(n_rows * W * spp / cpu_s / 1e6, 3), "cores": cores,
                               "kind": "port", "sample": f"{n_rows} rows of the frame, {spp} spp"}
        out["parity_vs_cpu"] = {"rows": n_rows, "tolerance": 1e-3,
                                "frac_pixels_beyond": float((err > 1e-3).mean()),
                                "mean_abs_err": float(np.abs(got[sel] - acc[sel]).mean())}
    return out


def run_multi(dev: str, world: int, rank: int, dist, comm=None, size=(1920, 1080), spp: int = 16,
              frames: int = 5) -> dict:
    """The same frame shared between `world` ranks: rank 0 builds the tree and uploads the scene, both are
    broadcast as one allocation each, every Render(Scene) call adds a sample to each rank's round-robin
    16-row stripes (compact, accumulated per rank) and the RGBA8 stripes are gathered to rank 0 two calls
    deep. Reports ms per frame of `spp` calls (max over ranks), the same without the gather, and whether
    the gathered frame equals the frame rank 0 renders alone."""
    from raytracer_amd import ray_tracer as rt
    from raytracer_amd.lbvh import LBVH
    from raytracer_amd.multi_gpu import StripeGatherer, broadcast_lbvh, broadcast_scene
    path = os.path.join(ROOT, "tests", "golden", "veach_mis.npz")
    if not os.path.exists(path):
        return {}
    d = dict(np.load(path))
    W, H = size
    if rank == 0:
        tracer, arr, c, proj = make_tracer(d, W, H, dev)
    else:  # an empty tracer and tree: both are filled by the broadcasts
        from raytracer_amd.data.scene_setup import veach_camera
        tracer = rt.RayTracer(dev)
        tracer.SetWorldToScene(np.eye(4, dtype=np.float32))
        tracer.SetLBVH(LBVH(dev))
        tracer.SetMaxPathDepth(4)
        c, proj = veach_camera(d, W, H)
        tracer.SetProjection(proj)
        tracer.SetCameraToWorld(c.pos, c.rotation())
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    broadcast_lbvh(tracer.lbvh, src=0, comm=comm)
    broadcast_scene(tracer, src=0, comm=comm)
    torch.cuda.synchronize()
    dist.barrier()
    bcast_ms = (time.perf_counter() - t0) * 1e3
    fb = FrameBuffer(W, H, dev)
    g = StripeGatherer(W, H, world, rank, dev, dst=0, comm=comm)

    def frame(gather=True):
        tracer.curr_accu_spp = 1.0  # restart the accumulation
        if tracer.d_accum is not None:
            tracer.d_accum.zero_()  # the first sample's mean is prev + (c - prev) / 1: exact from zero only
        for _ in range(spp):
            if gather:
                tracer.RenderGathered(fb, g)
            else:
                tracer.Render(fb, stripes=(rank, world), compact_out=g.send[0])

    def timed(gather):
        frame(gather)
        g.flush(fb.rgba)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(frames):
            frame(gather)
        if gather:
            g.flush(fb.rgba)  # the last calls' gathers and their assembly on rank 0 are inside the time
        torch.cuda.synchronize()
        dist.barrier()
        tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()) * 1e3 / frames

    ms = timed(True)
    ms_render = timed(False)
    frame(True)
    g.flush(fb.rgba)
    torch.cuda.synchronize()
    dist.barrier()
    same = None
    if rank == 0:
        got = fb.rgba.clone()
        alone, _, _, _ = make_tracer(d, W, H, dev)
        for _ in range(spp):
            alone.Render(fb)
        torch.cuda.synchronize()
        same = bool(torch.equal(got, fb.rgba))
    if rank != 0:
        return {}
    return {"workload": f"veach-mis {W}x{H}, {spp} spp, maxPathDepth 4, stripes of {world} GPUs gathered per call",
            "n_gpus": world, "ms_per_frame": round(ms, 3), "render_only_ms_per_frame": round(ms_render, 3),
            "gather_us_per_frame": round((ms - ms_render) * 1e3, 1),
            "Mpaths_per_s": round(W * H * spp / (ms * 1e-3) / 1e6, 1),
            "scene_and_tree_broadcast_ms": round(bcast_ms, 3),
            "broadcast_bytes": int(tracer.lbvh.d_blob.numel() + tracer.d_scene_blob.numel()),
            "gathered_frame_equals_single_gpu": same,
            "transport": "kk_comm (library RCCL)" if comm is not None else f"torch.distributed/{dist.get_backend()}"}


def config1_cpu(size=(512, 512), spp: int = 1) -> dict:
    """The first benchmark configuration as stated: the example scene, LBVH build + 512 x 512 path
    trace at 1 sample per pixel, the reference's logic (the CPU checker) on the host cores, no GPU."""
    path = os.path.join(ROOT, "tests", "golden", "veach_mis.npz")
    if not os.path.exists(path):
        return {}
    from oracle import kkoracle as ko
    from raytracer_amd.data.scene_setup import veach_arrays as scene_arrays, veach_camera as scene_camera
    d = dict(np.load(path))
    W, H = size
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    t0 = time.perf_counter()
    ref = ko.lbvh_build(d["positions"], d["face_position_indices"])
    build_ms = (time.perf_counter() - t0) * 1e3
    arr = scene_arrays(d)
    c, proj = scene_camera(d, W, H)
    acc = np.zeros((H, W, 3), np.float32)
    img = np.zeros((H, W, 4), np.uint8)
    t0 = time.perf_counter()
    for s in range(1, spp + 1):
        ko.pt_render_sample(ref, arr, np.eye(4, dtype=np.float32), cam.inverse_projective(proj), c.pos,
                            c.rotation(), acc, img, float(s))
    render_ms = (time.perf_counter() - t0) * 1e3
    return {"workload": f"veach-mis LBVH build + {W}x{H} {spp}-spp path trace, CPU only",
            "lbvh_build_ms": round(build_ms, 3), "render_ms": round(render_ms, 2), "cores": cores,
            "kind": "port", "triangles": int(d["face_position_indices"].shape[0]), "leaves": int(ref["M"]),
            "Mpaths_per_s": round(W * H * spp / (render_ms * 1e-3) / 1e6, 3),
            "lit_pixel_fraction": round(float((img[..., :3].max(axis=2) > 0).mean()), 4)}

"""DVR leg of the benchmark: Kingsnake-shaped 1024 x 1024 x 795 uint8 volume at 3840 x 2160, README
Kingsnake transfer function, log2Dims {5,4,3}, sampling rate 3, maxAlpha 0.9, depth boxes on; the 720
frames of the reference's own perf orbit (0.5 degrees per frame, app/dvr_dpbx_vdb.cpp:349-368). With several
ranks the VDB is built on rank 0 and broadcast, each rank renders its round-robin 16-row stripes and the
stripes are gathered to rank 0 once per frame."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from raytracer_amd import synthetic as syn
from raytracer_amd.data import camera as cam
from raytracer_amd.data.transfer_function import README_HEAD_TF, README_KINGSNAKE_TF, TransferFunction
from raytracer_amd.dpbx_vdb import DepthBoxVDB, Volume
from raytracer_amd.ray_caster import FrameBuffer, RayCaster, TransferFunctionTable

HBM_PEAK_GBPS = 8000.0


def run(dev: str, world: int, rank: int, dist=None, frames: int = 720, warmup: int = 8,
        dim=(1024, 1024, 795), size=(3840, 2160), cpu_rows: int = 24, with_cpu: bool = True,
        workload: str = "kingsnake", comm=None) -> dict:
    """workload "kingsnake": the 4K case above. "head": Head-shape 128x256x256 uint8 volume, README Head
    transfer function, spacing (2, 1, 1), 1920x1080 (pass dim and size accordingly)."""
    head = workload == "head"
    tf = TransferFunction(README_HEAD_TF if head else README_KINGSNAKE_TF)
    W, H = size
    vdb = DepthBoxVDB(dev)
    build = {}
    vol_t = None
    if rank == 0:
        vol_t = (torch.from_numpy(syn.head_volume(dim)).to(dev) if head
                 else syn.kingsnake_volume_torch(dim, dev))
        volume = Volume(vol_t, dev)
        cold_ms = None
        for attempt in range(2):  # the first build pays code-object loading and the allocator's growth
            vdb = DepthBoxVDB(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            vdb.BuildFrom(volume, tf.empty_scalar_ranges, apron_wid=1, log2_dims=(5, 4, 3),
                          use_depth_box=True)
            torch.cuda.synchronize()
            assert vdb.IsComplete(), "VDB build failed"
            if attempt == 0:
                cold_ms = (time.perf_counter() - t0) * 1e3
        build = {"total_ms": round((time.perf_counter() - t0) * 1e3, 2),
                 "first_build_ms": round(cold_ms, 2),
                 "classify_ms": round(vdb.timings["classify_ms"], 3),
                 "fill_atlas_ms": round(vdb.timings["fill_atlas_ms"], 3),
                 "leaf_bricks": int(vdb.plan.leaf_num),
                 "leaf_slots": int(np.prod(list(vdb.plan.leaf_slots)))}
        vol_bytes = dim[0] * dim[1] * dim[2]
        build["classify_GBps"] = round(vol_bytes / (vdb.timings["classify_ms"] * 1e-3) / 1e9, 1)
    bcast_ms = None
    if world > 1:
        from raytracer_amd.multi_gpu import broadcast_vdb
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        broadcast_vdb(vdb, src=0, comm=comm)
        torch.cuda.synchronize()
        dist.barrier()
        bcast_ms = (time.perf_counter() - t0) * 1e3

    scales = cam.volume_scales((2, 1, 1) if head else (1, 1, 1), dim)
    proj = cam.perspective_fov(cam.radians(90.0), W, H, 1.0, 3.0)
    tf_table = TransferFunctionTable(tf.dat, dev)

    def make_caster():
        rc = RayCaster(dev)
        rc.SetWorldToScene(cam.world_to_scene(scales))
        rc.SetDepthBoxVDB(vdb)
        rc.SetTransferFunctionTexture(tf_table)
        rc.SetProjection(proj)
        return rc

    caster = make_caster()
    c = cam.FPSCamera((0, 0, float(max(scales))), (0, 0, 0))
    # the orbit's camera poses, computed once (host work of the reference's frame loop)
    poses = []
    for _ in range(warmup + frames):
        c.Revolve(cam._c("sqrtf", cam._dot(c.pos, c.pos)), 0.5, 0.0)
        poses.append((c.pos.copy(), c.rotation().copy()))
    fb = FrameBuffer(W, H, dev)
    # Each frame's tiles (of the whole frame, or of this rank's stripes) are started in the order of what
    # they took in the frame before; the sort of the tile times is inside the timed loop. With one GPU
    # the same loop in plain launch order is timed beside it.
    caster.SetAdaptiveTileOrder(True)
    gatherer = None
    if world > 1:
        from raytracer_amd.multi_gpu import StripeGatherer
        gatherer = StripeGatherer(W, H, world, rank, dev, dst=0, comm=comm)

    def frame(i, gather=True):
        caster.SetCameraToWorld(*poses[i])
        if world > 1 and gather:
            # this rank's stripes go straight into the block that is sent; the gather of frame i runs
            # beside the rendering of frame i + 1, rank 0 puts frame i together when its slot comes round
            buf = gatherer.begin(fb.rgba)
            caster.RenderDepthBoxVDB(fb, stripes=(rank, world), compact_out=buf)
            gatherer.submit()
        elif world > 1:
            caster.RenderDepthBoxVDB(fb, stripes=(rank, world), compact_out=gatherer.send[0])
        else:
            caster.RenderDepthBoxVDB(fb)

    for i in range(warmup):
        frame(i)
    if gatherer:
        gatherer.flush(fb.rgba)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(warmup, warmup + frames):
        frame(i)
    if gatherer:
        gatherer.flush(fb.rgba)  # the last frames' gathers and their assembly on rank 0 are inside the time
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    wall = time.perf_counter() - t0
    render_only_ms = same_frame = None
    if dist:
        tt = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
        # the same frames without the gather: what the per-frame gather costs on top of the rendering
        n_short = min(frames, 60)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for i in range(warmup, warmup + n_short):
            frame(i, gather=False)
        torch.cuda.synchronize()
        dist.barrier()
        tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        render_only_ms = float(tt.item()) * 1e3 / n_short
        # one frame through the whole pipeline against the same frame rendered by rank 0 alone
        frame(warmup)
        gatherer.flush(fb.rgba)
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            got = fb.rgba.clone()
            caster.RenderDepthBoxVDB(fb)
            torch.cuda.synchronize()
            same_frame = bool(torch.equal(got, fb.rgba))
    fps2 = same_frame2 = None
    if dist:
        # Two frames in flight on every rank: a rank's share of a frame is bound by its longest rays (an
        # eighth of the stripes takes more than half the time of all of them), so the next frame's stripes
        # are rendered beside it on a second stream, with a caster (and thus a tile order), a frame buffer
        # and a slot of the gatherer per stream. The gathers stay in frame order on the gatherer's stream.
        from benchlib import two_streams
        from raytracer_amd.multi_gpu import StripeGatherer
        streams = two_streams(dev)
        from raytracer_amd import _lib as _lib2
        _lib2.lib().kk_sort_set_ticketed(1)  # tile-order sorts of the two streams may run side by side
        casters2 = [make_caster(), make_caster()]
        for rc in casters2:
            rc.SetAdaptiveTileOrder(True)
        fbs2 = [fb, FrameBuffer(W, H, dev)]
        g2 = StripeGatherer(W, H, world, rank, dev, dst=0, comm=comm)

        def frame2(i):
            k = i & 1
            casters2[k].SetCameraToWorld(*poses[i])
            with torch.cuda.stream(streams[k]):
                buf = g2.begin(fbs2[k].rgba)
                casters2[k].RenderDepthBoxVDB(fbs2[k], stripes=(rank, world), compact_out=buf)
                g2.submit()

        for i in range(warmup & ~1):  # an even count: slot and stream keep their pairing
            frame2(i)
        torch.cuda.synchronize()
        g2.flush(fb.rgba)
        torch.cuda.synchronize()
        dist.barrier()
        t2 = time.perf_counter()
        first = warmup & ~1
        for i in range(first, first + frames):
            frame2(i)
        torch.cuda.synchronize()
        g2.flush(fb.rgba)  # the last two frames' assembly on rank 0 is inside the time
        torch.cuda.synchronize()
        dist.barrier()
        tt = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        fps2 = frames / float(tt.item())
        if rank == 0:
            got = fb.rgba.clone()
            caster.SetCameraToWorld(*poses[first + frames - 1])
            caster.RenderDepthBoxVDB(fb)
            torch.cuda.synchronize()
            same_frame2 = bool(torch.equal(got, fb.rgba))
    if rank != 0:
        return {}
    out = {"workload": f"{'Head' if head else 'Kingsnake'}-shape {dim[0]}x{dim[1]}x{dim[2]} uint8, {W}x{H}, "
                       f"orbit 0.5 deg/frame",
           "fps": round(frames / wall, 2), "ms_per_frame": round(wall * 1e3 / frames, 4), "frames": frames,
           "n_gpus": world, "vdb_build": build}
    if bcast_ms is not None:
        out["vdb_broadcast_ms"] = round(bcast_ms, 3)
        out["vdb_broadcast_bytes"] = int(vdb.d_blob.numel())
        out["render_only_ms_per_frame"] = round(render_only_ms, 4)
        out["gather_us_per_frame"] = round((wall * 1e3 / frames - render_only_ms) * 1e3, 1)
        out["gathered_frame_equals_single_gpu"] = same_frame
        out["fps_two_frames_in_flight"] = round(fps2, 2)
        out["two_in_flight_gathered_frame_equals_single_gpu"] = same_frame2
        out["transport"] = "kk_comm (library RCCL)" if comm is not None else f"torch.distributed/{dist.get_backend()}"
    if world == 1:
        adaptive_last = fb.rgba.clone()
        caster.SetAdaptiveTileOrder(False)
        for i in range(warmup):
            frame(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(warmup, warmup + frames):
            frame(i)
        torch.cuda.synchronize()
        out["fps_launch_order"] = round(frames / (time.perf_counter() - t1), 2)
        out["adaptive_order_last_frame_equals_launch_order"] = bool(torch.equal(adaptive_last, fb.rgba))
        del adaptive_last
        # Two frames in flight: the orbit's frames alternate over two HIP streams and two frame buffers,
        # as a double-buffered presentation loop would submit them. A frame ends with only its longest
        # rays still marching (one 16-row stripe alone takes a quarter of a frame's time); the next frame
        # fills the GPU meanwhile. Every frame is the one the serial loop renders.
        from benchlib import two_streams
        streams = two_streams(dev)
        fbs = [FrameBuffer(W, H, dev), FrameBuffer(W, H, dev)]

        # a caster per stream: the tile order belongs to one stream's sequence of frames
        casters = [make_caster(), make_caster()]

        def frame2(i):
            casters[i & 1].SetCameraToWorld(*poses[i])
            with torch.cuda.stream(streams[i & 1]):
                casters[i & 1].RenderDepthBoxVDB(fbs[i & 1])

        last = warmup + frames - 1
        # (tile-order sorts of the two streams may run side by side: ticketed tiles, kk_sort_set_ticketed)
        from raytracer_amd import _lib
        _lib.lib().kk_sort_set_ticketed(1)
        for key, adaptive in (("fps_two_frames_in_flight_launch_order", False), ("fps_two_frames_in_flight", True)):
            for rc in casters:
                rc.SetAdaptiveTileOrder(adaptive)
            for i in range(warmup):
                frame2(i)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(warmup, warmup + frames):
                frame2(i)
            torch.cuda.synchronize()
            out[key] = round(frames / (time.perf_counter() - t2), 2)
            out["two_in_flight_last_frame_equals_serial"] = bool(
                out.get("two_in_flight_last_frame_equals_serial", True) and
                torch.equal(fbs[last & 1].rgba, fb.rgba))
        _lib.lib().kk_sort_set_ticketed(0)
        for rc in casters:
            rc.CheckTileOrder()
        caster.CheckTileOrder()
        # work counters of one frame: samples taken and bricks entered
        counters = torch.zeros(2, dtype=torch.int64, device=dev)
        caster.SetCameraToWorld(*poses[warmup])
        caster.RenderDepthBoxVDB(fb, counters=counters)
        torch.cuda.synchronize()
        ns, nb = int(counters[0]), int(counters[1])
        kernel_ms = ev0.elapsed_time(ev1) / frames
        out["gpu_ms_per_frame"] = round(kernel_ms, 4)  # device-side time between the first and the last frame
        # 8 voxel bytes + 2 table entries (32 B, LDS) per sample from cache; compulsory HBM traffic is
        # the distinct bricks touched plus the 4 B per pixel written
        out.update({"samples_per_frame": ns, "brick_entries_per_frame": nb,
                    "Gsamples_per_s": round(ns / (kernel_ms * 1e-3) / 1e9, 3),
                    "voxel_fetch_GBps": round(8 * ns / (kernel_ms * 1e-3) / 1e9, 1),
                    "compulsory_bytes_per_frame": int(vdb.plan.atlas_voxels) + 4 * W * H,
                    "compulsory_frac_of_hbm_peak": round(
                        (int(vdb.plan.atlas_voxels) + 4 * W * H) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)})
        # the marcher is bound by instruction issue, not by HBM (compulsory_frac_of_hbm_peak above): its
        # frame against the time its wave instructions need on all SIMDs
        from benchlib.roofline import valu_issue
        tag = "dvr_head" if head else "dvr"
        rl = valu_issue([f"r03_{tag}_pmc.json"] + ([] if head else ["r02_dvr_pmc.json"]), kernel_ms)
        if rl:
            rl["kernel"] = "dvr_kernel"
            rl["frac_two_frames_in_flight"] = round(rl["peak"] / (1e3 / out["fps_two_frames_in_flight"]), 4)
            rl["frac_launch_order"] = round(rl["peak"] / (1e3 / out["fps_launch_order"]), 4)
            rl["hbm_frac_of_peak"] = out["compulsory_frac_of_hbm_peak"]
            out["roofline"] = rl
        if with_cpu:
            from oracle import kkoracle as ko
            cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
            vol_h = vol_t.cpu().numpy()
            t0 = time.perf_counter()
            ref = ko.OracleVDB(vol_h, (5, 4, 3), 1, True, tf.empty_scalar_ranges)
            cpu_build_s = time.perf_counter() - t0
            y0 = H // 2 - cpu_rows // 2
            pos, rot = poses[warmup]
            t0 = time.perf_counter()
            want, cpu_ns = ko.dvr_render(ref, tf.dat, cam.world_to_scene(scales),
                                         cam.inverse_projective(proj), pos, rot, W, H,
                                         rows=(y0, y0 + cpu_rows))
            cpu_s = time.perf_counter() - t0
            got = fb.ReadBack()[y0:y0 + cpu_rows]
            d = np.abs(got.astype(int) - want[y0:y0 + cpu_rows].astype(int))
            out["cpu_baseline"] = {"fps_equivalent": round(1.0 / (cpu_s * H / cpu_rows), 4),
                                   "unit": "fps", "cores": cores, "kind": "port",
                                   "sample": f"{cpu_rows} centre rows of one {W}x{H} frame ({cpu_ns} samples)",
                                   "vdb_build_ms": round(cpu_build_s * 1e3, 1)}
            out["parity_vs_cpu"] = {"rows": cpu_rows, "max_level_diff": int(d.max()),
                                    "frac_pixels_differing": float((d.max(axis=2) > 0).mean()),
                                    "alpha_equal": bool((d[..., 3] == 0).all()),
                                    "vdb_bit_exact": bool(np.array_equal(vdb.atlas_to_host(), ref.atlas) and
                                                          np.array_equal(vdb.host_children, ref.child_pools))}
    return out

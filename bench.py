#!/usr/bin/env python3
"""Benchmark of the hot paths on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the closest-hit path over one batch of synthetic input: 2,097,152 incoherent
rays (seed 0x5EED0002) traced through the LBVH of the 10 M-triangle soup (seed 0x5EED0001); the rays
and the tree (with its trace-side repacking) are resident in HBM before the timed region. A step
orders the rays (Morton keys + radix sort) and traces them; the dominant kernel's own duration is
measured in a second loop that reuses the order. The same run also times the LBVH build of
that soup (ms per 10 M triangles) and, when the volume path is built, the DVR frame rate; they are
reported in the same JSON line beside the headline value.

N > 1 (launched with torch.distributed.run, one rank per GPU): the tree is built once on rank 0 and
broadcast over RCCL; every rank then traces its own 2 M-ray slice of the ray stream (weak scaling,
no collective on the data path). Timing is bracketed by barrier + synchronize and the maximum over
ranks is taken.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raytracer_amd import synthetic as syn  # noqa: E402
from raytracer_amd.lbvh import LBVH  # noqa: E402
from raytracer_amd.ray_tracer import closest_hit  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak
# algorithmic bytes of the build per triangle of a non-indexed soup: bounds 36, Morton 52, sort 68, run
# heads 8, leaf boxes 84, Karras links 16, refit 104
LBVH_BYTES_PER_TRI = 368


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # 0.25 s: one host hiccup of a few ms is not the result
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--rays", type=int, default=2_097_152)
    ap.add_argument("--cpu-rays", type=int, default=262_144, help="rays of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-dvr", action="store_true")
    ap.add_argument("--no-pt", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("KK_BENCH_REHEARSAL") == "1":
            # every rank on GPU 0 over gloo: runs the N > 1 control flow on a one-GPU box (RCCL refuses
            # two ranks on one device); its numbers mean nothing
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    N, R, K, W = args.tris, args.rays, args.steps, args.warmup

    # ---- scene: built once (rank 0), broadcast to the others ----------------------------------------
    bvh = LBVH(dev)
    build_ms = None
    pos = tri = None
    if rank == 0:
        pos, tri = syn.triangle_soup(N)
        bvh.BuildFrom(torch.from_numpy(pos), torch.from_numpy(tri.view(np.int32)))
        assert bvh.IsComplete(), "LBVH build failed"
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        times = []
        for i in range(3 + 10):
            ev[0].record()
            bvh.rebuild()
            ev[1].record()
            torch.cuda.synchronize()
            if i >= 3:
                times.append(ev[0].elapsed_time(ev[1]))
        build_ms = float(np.median(times))
    bcast_ms = None
    comm = None
    if world > 1:
        from raytracer_amd.multi_gpu import Communicator, broadcast_lbvh
        if os.environ.get("KK_BENCH_COMM") == "capi" and os.environ.get("KK_BENCH_REHEARSAL") != "1":
            # the library's own RCCL communicator (what a C++ host uses) instead of torch.distributed's
            comm = Communicator(rank, world, dev)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        broadcast_lbvh(bvh, src=0, comm=comm)
        torch.cuda.synchronize()
        dist.barrier()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    M = bvh.GetDeviceData()["compactedFaceNum"]

    # ---- rays: each rank owns rays [rank*R, (rank+1)*R) of the stream --------------------------------
    rays_h = syn.incoherent_rays(R, first=rank * R)
    rays = torch.from_numpy(rays_h).to(dev)
    accel_ms = None
    bvh.trace_accel()
    if rank == 0:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        times = []
        for i in range(2 + 5):
            ev[0].record()
            bvh.trace_accel(rebuild=True)
            ev[1].record()
            torch.cuda.synchronize()
            if i >= 2:
                times.append(ev[0].elapsed_time(ev[1]))
        accel_ms = float(np.median(times))
    t, bary, prim, counters = closest_hit(bvh, rays, count=True)
    torch.cuda.synchronize()
    from raytracer_amd import _lib
    ws = torch.empty(_lib.lib().kk_closest_hit_accel_workspace_bytes(R), dtype=torch.uint8, device=dev)
    l_out = (t, bary, prim, ws)

    for _ in range(W):
        closest_hit(bvh, rays, out=l_out)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    import gc
    gc.collect()
    gc.disable()  # (a collection inside the timed region would be host time, not the path's)
    t0 = time.perf_counter()
    for _ in range(K):
        closest_hit(bvh, rays, out=l_out)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    wall = time.perf_counter() - t0
    gc.enable()
    if dist:
        tt = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
    ms_per_step = wall * 1e3 / K
    value = world * R / (ms_per_step * 1e-3) / 1e6
    # the trace kernel alone (same rays, order kept from the last step): HIP events on the launch stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(K):
        closest_hit(bvh, rays, out=l_out, reuse_order=True)
    ev1.record()
    torch.cuda.synchronize()
    kernel_ms = ev0.elapsed_time(ev1) / K
    # two batches in flight: steps alternate over two streams, each with its own outputs and workspace, so
    # that one step's key + sort and the tail of its trace run beside the other's trace (what a renderer
    # with independent ray batches would do; `value` above stays one step at a time)
    in_flight = None
    if world == 1:
        from benchlib import two_streams
        streams = two_streams(dev)
        outs = [l_out, tuple(torch.empty_like(x) for x in l_out)]

        def step2(i):
            with torch.cuda.stream(streams[i & 1]):
                closest_hit(bvh, rays, out=outs[i & 1])
        torch.cuda.synchronize()
        # two ray sorts may now run side by side: their tiles take tickets instead of leaning on the
        # start order of workgroups (kk_sort_set_ticketed)
        _lib.lib().kk_sort_set_ticketed(1)
        for i in range(4):
            step2(i)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(K):
            step2(i)
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t2) * 1e3 / K
        _lib.lib().kk_sort_set_ticketed(0)
        in_flight = {"Mrays_per_s": round(R / (ms2 * 1e-3) / 1e6, 2), "ms_per_step": round(ms2, 4),
                     "sort_tiles": "ticketed",
                     "results_equal_one_at_a_time": bool(torch.equal(outs[0][0], outs[1][0]) and
                                                         torch.equal(outs[0][2], outs[1][2]))}
    # the optional watertight triangle test, same rays: throughput and how often its primitive differs
    prim_ref = prim.clone()
    t_ref = t.clone()
    bary_ref = bary.clone()
    ev0.record()
    for _ in range(K):
        closest_hit(bvh, rays, out=l_out, watertight=True)
    ev1.record()
    torch.cuda.synchronize()
    wt_ms = ev0.elapsed_time(ev1) / K
    wt_mism = prim != prim_ref
    wt_same = (~wt_mism) & (prim_ref >= 0)
    wt_rel = ((t[wt_same] - t_ref[wt_same]).abs() / t_ref[wt_same].abs().clamp_min(1e-30))
    watertight = {"Mrays_per_s": round(R / (wt_ms * 1e-3) / 1e6, 2),
                  "prim_mismatch_frac_vs_default": float(wt_mism.float().mean()),
                  "frac_same_prim_with_rel_t_err_above_1e-5": float((wt_rel > 1e-5).float().mean()),
                  "max_rel_t_err_same_prim": float(wt_rel.max()) if wt_rel.numel() else 0.0}
    # the entry-bound option (groups of pending slots dropped unvisited behind the best hit): same hits,
    # fewer record visits, more instructions per visit
    _, _, _, eb_counters = closest_hit(bvh, rays, count=True, entry_bound=True)
    closest_hit(bvh, rays, out=l_out, entry_bound=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(K):
        closest_hit(bvh, rays, out=l_out, entry_bound=True)
    ev1.record()
    torch.cuda.synchronize()
    eb_ms = ev0.elapsed_time(ev1) / K
    eb_equal = bool(torch.equal(prim, prim_ref) and torch.equal(t, t_ref) and torch.equal(bary, bary_ref))
    ev0.record()
    for _ in range(K):
        closest_hit(bvh, rays, out=l_out, entry_bound=True, reuse_order=True)
    ev1.record()
    torch.cuda.synchronize()
    entry_bound = {"Mrays_per_s": round(R / (eb_ms * 1e-3) / 1e6, 2),
                   "kernel_ms": round(ev0.elapsed_time(ev1) / K, 4),
                   "node_records_per_ray": round(eb_counters["node_fetches"] / R, 2),
                   "empty_visits_per_ray": round(eb_counters["empty_visits"] / R, 2),
                   "empty_visits_by_limit_per_ray": round(eb_counters["empty_visits_by_limit"] / R, 2),
                   "slots_dropped_unvisited_per_ray": round(eb_counters["slots_dropped_unvisited"] / R, 2),
                   "results_equal_default": eb_equal}
    closest_hit(bvh, rays, out=l_out)  # leave the default results in the output buffers
    torch.cuda.synchronize()
    # no step's ray sort gave up (the kernel would have traced in input order, at another rate)
    from raytracer_amd.ray_tracer import closest_hit_status
    closest_hit_status(ws, R)

    dvr = dvr_head = None
    if not args.no_dvr:
        from benchlib import bench_dvr
        dvr = bench_dvr.run(dev, world, rank, dist, with_cpu=not args.no_cpu, comm=comm)
        dvr_head = bench_dvr.run(dev, world, rank, dist, with_cpu=not args.no_cpu, workload="head",
                                 dim=(128, 256, 256), size=(1920, 1080), cpu_rows=16, comm=comm)

    pt_multi = None
    if world > 1 and not args.no_pt:
        from benchlib import bench_pt
        pt_multi = bench_pt.run_multi(dev, world, rank, dist, comm=comm)

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (closest_hit_accel_kernel) ---------------------------------
    # algorithmic bytes per launch: the 80 B a visit reads of a 128-B node record + 48 B per triangle
    # record tested + 4 B order index + 24 B ray in + 16 B hit out, from the kernel's own work counters
    alg_bytes = 80 * counters["node_fetches"] + 48 * counters["tri_tests"] + (4 + 24 + 16) * R
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    # the same counters priced as a walk over the reference's 32-byte nodes would be: 32 B per node
    # visited + 48 B per triangle tested + 24 B ray in + 16 B hit out
    survey_bytes = 32 * counters["node_fetches"] + 48 * counters["tri_tests"] + (24 + 16) * R
    passes = max(counters["wave_passes"], 1)
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                "kernel": "closest_hit_accel_kernel", "avg_launch_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "frac_survey_formula": round(survey_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "survey_formula": "32 B x node records + 48 B x triangles + 40 B per ray",
                # HBM is the roofline the fraction is taken against; what the counters of the committed
                # profile show the kernel waiting for is the CUs' own vector-memory path and VALU issue
                "limited_by": "CU vector-memory path (per-lane record fetches) and VALU issue, not HBM: see "
                              "td_busy_frac / valu_busy in the profile digest",
                "node_records_per_ray": round(counters["node_fetches"] / R, 2),
                "tri_tests_per_ray": round(counters["tri_tests"] / R, 2),
                "empty_visits_per_ray": round(counters["empty_visits"] / R, 2),
                "empty_visits_by_limit_per_ray": round(counters["empty_visits_by_limit"] / R, 2),
                "lanes_visiting_per_wave_pass": round(counters["visit_lanes"] / passes, 1),
                "lanes_per_triangle_phase": round(counters["tri_lanes"] / max(counters["tri_phases"], 1), 1)}
    # HBM traffic needs hardware counters (a rocprofv3 --pmc pass), which a plain run cannot read: the
    # figure of the committed counter digest of the same kernel and workload is quoted beside a null
    for name in ("r03_closest_hit_pmc.json", "r02_closest_hit_pmc.json", "r01_closest_hit_pmc.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            try:
                digest = json.load(open(pmc))
                roofline["traffic_from_profile"] = digest.get("hbm_bytes_per_launch")
                if "valu_instructions_per_launch" in digest:
                    # VALU issue: 0.6 G wave instructions per second and SIMD (4 cycles at 2.4 GHz) on 1,024
                    # SIMDs; a mixed stream was measured at 0.72 (benchlib/roofline.py says how)
                    roofline["valu_issue_floor_ms_from_profile"] = round(
                        4 * digest["valu_instructions_per_launch"] / (1024 * 2.4e9) * 1e3, 4)
                    roofline["valu_issue_floor_ms_at_measured_mix_rate"] = round(
                        digest["valu_instructions_per_launch"] / (1024 * 0.72e9) * 1e3, 4)
                    roofline["td_busy_frac_from_profile"] = round(
                        digest.get("texture_path", {}).get("td_busy_frac", 0.0), 3)
                roofline["traffic_source"] = (f"profiles/{name}: rocprofv3 --pmc passes of scripts/prof_small.py "
                                              "(same kernel, same workload), not measured in this run")
            except (OSError, ValueError):
                pass
            break

    out = {
        "metric": "Mrays/s closest-hit", "value": round(value, 2), "unit": "Mrays/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "10M-triangle soup LBVH + 2M incoherent closest-hit rays per GPU",
                   "triangles": N, "leaves": M, "rays_per_gpu": R, "parallelism": f"replicated-bvh x{world}"},
        "roofline": roofline,
        "hit_fraction": round(float((prim >= 0).float().mean()), 4),
        "two_batches_in_flight": in_flight,
        "watertight_option": watertight,
        "entry_bound_option": entry_bound,
    }
    if build_ms is not None:
        gbps = LBVH_BYTES_PER_TRI * N / (build_ms * 1e-3) / 1e9
        out["trace_accel_build_ms"] = round(accel_ms, 4)
        out["lbvh_build"] = {"ms": round(build_ms, 4), "ms_per_10Mtri": round(build_ms * 1e7 / N, 4),
                             "bytes_per_tri": LBVH_BYTES_PER_TRI, "achieved_GBps": round(gbps, 1),
                             "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4)}
    if bcast_ms is not None:
        out["lbvh_broadcast_ms"] = round(bcast_ms, 3)
        out["lbvh_broadcast_bytes"] = int(bvh.d_blob.numel())

    # ---- CPU baseline: the reference's logic (oracle port) on the host cores, bounded sample --------
    if world == 1 and not args.no_cpu:
        from oracle import kkoracle as ko
        cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
        if pos is None:
            pos, tri = syn.triangle_soup(N)
        t0 = time.perf_counter()
        ref = ko.lbvh_build(pos, tri, refit_mode=1)
        cpu_build_s = time.perf_counter() - t0
        S = min(args.cpu_rays, R)
        t0 = time.perf_counter()
        rt, rbary, rprim, rstats = ko.closest_hit(ref, pos, tri, rays_h[:S])
        cpu_trace_s = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": round(S / cpu_trace_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"first {S} of the {R} rays through the same 10M-triangle LBVH "
                      f"(reference traversal order, no distance culling)",
            "lbvh_build_ms_per_10Mtri": round(cpu_build_s * 1e3 * 1e7 / N, 1),
            "node_fetches_per_ray": round(rstats["node_fetches"] / S, 1),
        }
        # the sample doubles as a full-size parity check
        gp = prim[:S].cpu().numpy().view(np.uint32)
        gt = t[:S].cpu().numpy()
        mism = gp != rprim
        hit = ~mism & (rprim != 0xFFFFFFFF)
        rel = np.abs(gt[hit] - rt[hit]) / np.maximum(np.abs(rt[hit]), 1e-30)
        out["parity_vs_cpu"] = {"rays": int(S), "prim_mismatch_frac": float(mism.mean()),
                                "max_rel_t_err": float(rel.max()) if rel.size else 0.0,
                                "tree_bit_exact": bool(
                                    ref["M"] == M and
                                    np.array_equal(bvh.d_faces.cpu().numpy().view(np.uint32), ref["faces"]) and
                                    bvh.d_nodes[:(M - 1) * 32].cpu().numpy().tobytes() ==
                                    ref["internalNodes"].tobytes())}

    # ---- path tracer on the reference's scene (1920x1080, 16 spp), one GPU ---------------------------
    if world == 1 and not args.no_pt:
        from benchlib import bench_pt
        pt = bench_pt.run(dev, with_cpu=not args.no_cpu)
        if pt:
            out["path_trace"] = pt
        if not args.no_cpu:
            c1 = bench_pt.config1_cpu()
            if c1:
                out["config1_cpu_only"] = c1

    if pt_multi:
        out["path_trace"] = pt_multi

    # ---- volume path (reported beside the headline value) -------------------------------------------
    if dvr:
        out["dvr"] = dvr
        out["dvr_fps_4k"] = dvr["fps"]
        out["dvr_fps_4k_launch_order"] = dvr.get("fps_launch_order")
        out["dvr_fps_4k_two_frames_in_flight"] = dvr.get("fps_two_frames_in_flight")
    if dvr_head:
        out["dvr_head_1080p"] = dvr_head

    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

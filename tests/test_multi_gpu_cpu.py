"""The N > 1 path rehearsed with two gloo ranks on the CPU. The structures are the CPU checker's (a real
LBVH of a triangle soup, a real Depth-Box VDB) and the stripes are rendered by the checker's marcher, so
what crosses the rank boundary is what the GPU ranks exchange: the tree's one allocation, the VDB's one
allocation, compact stripe blocks gathered two frames deep."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fill_tree(bvh, ko, syn, n_tri, seed):
    """An LBVH object on host tensors holding the checker's tree of a soup of n_tri triangles."""
    from raytracer_amd import _lib, multi_gpu
    pos, tri = syn.triangle_soup(n_tri, seed=seed)
    ref = ko.lbvh_build(pos, tri)
    N, V, M = n_tri, pos.shape[0], int(ref["M"])
    bvh._clear()
    bvh.alloc_blob(N, V)
    bvh.d_positions.copy_(torch.from_numpy(pos))
    bvh.d_face_pos_idx.copy_(torch.from_numpy(tri.view(np.int32)))
    bvh.d_faces.copy_(torch.from_numpy(ref["faces"].view(np.int32)))
    bvh.d_compacted[:M].copy_(torch.from_numpy(ref["compactedFaces"].view(np.int32)))
    bvh.d_nodes[:(M - 1) * 32].copy_(torch.from_numpy(ref["internalNodes"].view(np.uint8).reshape(-1)))
    dd = _lib.LbvhDeviceData()
    box = ref["rootAABB"].reshape(-1)
    for k in range(3):
        dd.root_aabb_min[k], dd.root_aabb_max[k] = float(box[k]), float(box[3 + k])
    dd.face_num, dd.compacted_face_num = N, M
    bvh.d_device_data.copy_(torch.from_numpy(np.frombuffer(bytes(dd), np.uint8).copy()))
    multi_gpu.bind_lbvh_blob(bvh)
    bvh._complete = True
    return ref


def _check_tree(bvh, ref, n_tri):
    dd = bvh.GetDeviceData()
    M = int(ref["M"])
    assert bvh.IsComplete() and dd["faceNum"] == n_tri and dd["compactedFaceNum"] == M
    assert dd["faces"] == bvh.d_faces.data_ptr() and dd["internalNodes"] == bvh.d_nodes.data_ptr()
    assert dd["compactedFaces"] == bvh.d_compacted.data_ptr()
    np.testing.assert_array_equal(dd["rootAABB"], ref["rootAABB"].reshape(-1))
    assert bvh.d_faces.numpy().view(np.uint32).tobytes() == ref["faces"].tobytes()
    assert bvh.d_compacted[:M].numpy().view(np.uint32).tobytes() == ref["compactedFaces"].tobytes()
    assert bvh.d_nodes[:(M - 1) * 32].numpy().tobytes() == ref["internalNodes"].tobytes()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import kkoracle as ko
    from raytracer_amd import _lib, multi_gpu, synthetic as syn
    from raytracer_amd.data import camera as cam
    from raytracer_amd.dpbx_vdb import DepthBoxVDB
    from raytracer_amd.lbvh import LBVH
    from vol_util import HEAD_TF, dvr_setup

    # ---- LBVH broadcast, twice into the same object: the second tree must replace the first whole -------
    bvh = LBVH("cpu")
    for n_tri, seed in ((3000, 11), (1700, 12)):
        if rank == 0:
            _fill_tree(bvh, ko, syn, n_tri, seed)
        else:  # leftovers of an earlier tree that must not survive a broadcast
            bvh.d_accel, bvh._accel_stale, bvh.d_workspace = "old repacking", False, "old workspace"
            bvh._host_dd = {"stale": True}
        multi_gpu.broadcast_lbvh(bvh, src=0)
        pos, tri = syn.triangle_soup(n_tri, seed=seed)
        _check_tree(bvh, ko.lbvh_build(pos, tri), n_tri)
        assert bvh.d_positions.numpy().tobytes() == pos.tobytes()
        if rank != 0:
            assert bvh.d_accel is None and bvh._accel_stale and bvh.d_workspace is None

    # ---- VDB broadcast ------------------------------------------------------------------------------------
    dim = (32, 64, 64)
    vol = syn.head_volume(dim)
    ref = ko.OracleVDB(vol, (3, 2, 2), 1, True, HEAD_TF.empty_scalar_ranges)
    vdb = DepthBoxVDB("cpu")
    if rank == 0:
        plan = _lib.VdbPlan()
        C.memmove(C.byref(plan.param), ref.param.tobytes(), 124)
        plan.leaf_num, plan.node_num, plan.child_num = ref.leaf_num, len(ref.nodes), len(ref.child_pools)
        plan.atlas_map_num, plan.atlas_voxels = len(ref.atlas_brick_to_node), ref.atlas.size
        vdb.alloc_blob(plan, 1)
        vdb.d_node_pools.copy_(torch.from_numpy(ref.nodes.view(np.uint8).reshape(-1)))
        vdb.d_child_pools.copy_(torch.from_numpy(ref.child_pools.view(np.int32)))
        vdb.d_atlas_brick_to_node.copy_(torch.from_numpy(ref.atlas_brick_to_node.view(np.int32)))
        vdb.d_atlas.copy_(torch.from_numpy(ref.atlas.reshape(-1)))
        vdb._complete = True
    multi_gpu.broadcast_vdb(vdb, src=0)
    assert vdb.IsComplete() and vdb.plan.leaf_num == ref.leaf_num and vdb.voxel_bytes == 1
    assert bytes(vdb.plan.param)[:124] == ref.param.tobytes()[:124]
    dd = vdb.GetDeviceData()
    assert dd.atlas == vdb.d_atlas.data_ptr() and dd.child_pools == vdb.d_child_pools.data_ptr()
    assert dd.node_pools == vdb.d_node_pools.data_ptr() and dd.leaf_num == ref.leaf_num
    assert vdb.d_atlas.numpy().tobytes() == ref.atlas.tobytes()
    assert vdb.d_child_pools.numpy().view(np.uint32).tobytes() == ref.child_pools.tobytes()
    assert vdb.d_node_pools.numpy().tobytes() == ref.nodes.tobytes()

    # ---- stripes: each rank renders its own (the checker's marcher), gathered two frames deep ---------------
    W, H = 40, 100   # 7 stripes, the last one partial
    g = multi_gpu.StripeGatherer(W, H, world, rank, "cpu")
    frame = torch.zeros((H, W, 4), dtype=torch.uint8)
    assembled, want = [], []
    for orbit in (0, 30, 60):
        w2s, c, proj = dvr_setup(dim, (2, 1, 1), W, H, 1.0, orbit)
        view = (w2s, cam.inverse_projective(proj), c.pos, c.rotation())
        buf = g.begin(frame)
        if g.frames >= 2 and rank == 0:
            assembled.append(frame.numpy().copy())  # the frame of two submits ago has just been put together
        for j, s in enumerate(multi_gpu.stripes_of(H, world, rank)):
            rows = (16 * s, min(H, 16 * s + 16))
            img = np.zeros((H, W, 4), np.uint8)
            ko.dvr_render(ref, HEAD_TF.dat, *view, W, H, rows=rows, rgba=img)
            buf[j, :rows[1] - rows[0]] = torch.from_numpy(img[rows[0]:rows[1]])
        g.submit()
        if rank == 0:
            want.append(ko.dvr_render(ref, HEAD_TF.dat, *view, W, H)[0])
    if rank == 0:
        g._wait(g.slot, frame)
        assembled.append(frame.numpy().copy())
        g._wait(g.slot ^ 1, frame)
        assembled.append(frame.numpy().copy())
        assert len(assembled) == 3
        for a, b in zip(assembled, want):
            assert a.tobytes() == b.tobytes()
        assert (want[0][..., 3] > 0).mean() > 0.05
    else:
        g.flush()
    # ---- path tracer: the scene's one allocation is broadcast, each rank path-traces its stripes with the
    # checker from the arrays it RECEIVED, two Render(Scene) calls are gathered two frames deep ------------
    from raytracer_amd.data.scene_setup import veach_arrays, veach_camera
    from raytracer_amd.ray_tracer import RayTracer
    veach = dict(np.load(os.path.join(ROOT, "tests", "golden", "veach_mis.npz")))
    tracer = RayTracer("cpu")
    if rank == 0:
        arr0 = veach_arrays(veach)
        tracer.SetMesh(veach["positions"], veach["normals"], veach["tex_coords"],
                       veach["grp_start_face_indices"], veach["face_position_indices"],
                       veach["face_normal_indices"], veach["face_tex_coord_indices"], arr0["lights"],
                       arr0["materials"])
    multi_gpu.broadcast_scene(tracer, src=0)
    arr = tracer.scene_host_arrays()
    want_arr = veach_arrays(veach)
    for name in ("triangles", "lights", "materials"):
        assert arr[name].tobytes() == want_arr[name].tobytes(), name
    assert arr["positions"].tobytes() == veach["positions"].astype(np.float32).tobytes()
    assert tracer._scene.light_num == 3 and tracer._scene.triangles == tracer._keep["triangles"].data_ptr()
    assert tracer._scene.materials == tracer._keep["materials"].data_ptr()
    tree = ko.lbvh_build(arr["positions"], arr["tri"])
    PW, PH = 48, 50  # 4 stripes, the last one partial
    c, proj = veach_camera(veach, PW, PH)
    view = (np.eye(4, dtype=np.float32), cam.inverse_projective(proj), c.pos, c.rotation())
    g = multi_gpu.StripeGatherer(PW, PH, world, rank, "cpu")
    frame = torch.zeros((PH, PW, 4), dtype=torch.uint8)
    acc = np.zeros((PH, PW, 3), np.float32)
    img = np.zeros((PH, PW, 4), np.uint8)
    got_frames = []
    for sample in (1, 2, 3):
        buf = g.begin(frame)
        if g.frames >= 2 and rank == 0:
            got_frames.append(frame.numpy().copy())
        for j, s in enumerate(multi_gpu.stripes_of(PH, world, rank)):
            rows = (16 * s, min(PH, 16 * s + 16))
            ko.pt_render_sample(tree, arr, *view, acc, img, float(sample), rows=rows)
            buf[j, :rows[1] - rows[0]] = torch.from_numpy(img[rows[0]:rows[1]])
        g.submit()
    if rank == 0:
        for k in (g.slot, g.slot ^ 1):
            g._wait(k, frame)
            got_frames.append(frame.numpy().copy())
        wacc = np.zeros((PH, PW, 3), np.float32)
        wimg = np.zeros((PH, PW, 4), np.uint8)
        for sample in (1, 2, 3):  # the frame one rank renders, call by call
            ko.pt_render_sample(ko.lbvh_build(veach["positions"], veach["face_position_indices"]), want_arr,
                                *view, wacc, wimg, float(sample))
            assert got_frames[sample - 1].tobytes() == wimg.tobytes(), f"Render call {sample}"
        assert (wimg[..., :3].max(axis=2) > 0).mean() > 0.2
    else:
        g.flush()
    # the blocking form over a frame rendered in place
    frame2 = torch.zeros((H, W, 4), dtype=torch.uint8)
    for s in multi_gpu.stripes_of(H, world, rank):
        frame2[16 * s:16 * s + 16] = 10 * (rank + 1) + s
    multi_gpu.gather_stripes(frame2, world, rank, 0)
    if rank == 0:
        for s in range(7):
            assert (frame2[16 * s:min(H, 16 * s + 16)] == 10 * (s % world + 1) + s).all()
    assert multi_gpu.split_range(10, 3, 0) == (0, 4) and multi_gpu.split_range(10, 3, 2) == (7, 10)
    assert multi_gpu.stripes_per_rank(H, world) == _lib.lib().kk_stripes_per_rank(H, world) == 4
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_and_gather(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)

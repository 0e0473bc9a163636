"""The N > 1 path rehearsed with two gloo ranks on the CPU: structure broadcast and stripe gather."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd import _lib, multi_gpu
    from raytracer_amd.lbvh import LBVH
    from raytracer_amd.dpbx_vdb import DepthBoxVDB

    # ---- LBVH broadcast: arrays made on rank 0 arrive bit for bit, the header is rebuilt locally ----
    bvh = LBVH("cpu")
    N, V, M = 1000, 3000, 990
    if rank == 0:
        g = torch.Generator().manual_seed(1)
        bvh.face_num = N
        bvh.d_positions = torch.rand((V, 3), generator=g)
        bvh.d_face_pos_idx = torch.arange(3 * N, dtype=torch.int32).reshape(N, 3)
        bvh.d_faces = torch.randperm(N, generator=g).to(torch.int32)
        bvh.d_compacted = torch.arange(N, dtype=torch.int32)
        bvh.d_nodes = torch.randint(0, 255, (N * 32,), generator=g, dtype=torch.uint8)
        multi_gpu.write_lbvh_device_data(bvh, np.array([0, 0, 0, 1, 1, 1], np.float32), N, M)
        bvh._complete = True
    multi_gpu.broadcast_lbvh(bvh, src=0)
    dd = bvh.GetDeviceData()
    assert bvh.IsComplete() and dd["faceNum"] == N and dd["compactedFaceNum"] == M
    assert dd["faces"] == bvh.d_faces.data_ptr() and dd["internalNodes"] == bvh.d_nodes.data_ptr()
    torch.save({"faces": bvh.d_faces, "nodes": bvh.d_nodes[:(M - 1) * 32], "pos": bvh.d_positions},
               os.path.join(out_dir, f"bvh{rank}.pt"))

    # ---- VDB broadcast -----------------------------------------------------------------------------------
    vdb = DepthBoxVDB("cpu")
    if rank == 0:
        plan = _lib.VdbPlan()
        plan.leaf_num, plan.node_num, plan.child_num, plan.atlas_map_num = 3, 4, 64, 4
        plan.param.vox_per_atlas_brick, plan.param.root_lev = 10, 1
        plan.atlas_voxels = 3 * 1000
        vdb.plan, vdb.voxel_bytes = plan, 1
        vdb.d_node_pools = torch.arange(4 * 32, dtype=torch.uint8)
        vdb.d_child_pools = torch.arange(64, dtype=torch.int32)
        vdb.d_atlas_brick_to_node = torch.arange(4, dtype=torch.int32)
        vdb.d_atlas = (torch.arange(3000) % 251).to(torch.uint8)
        vdb.device_data = _lib.VdbDeviceData()
        vdb.rebind()
    multi_gpu.broadcast_vdb(vdb, src=0)
    assert vdb.plan.leaf_num == 3 and vdb.plan.param.vox_per_atlas_brick == 10
    assert vdb.GetDeviceData().atlas == vdb.d_atlas.data_ptr()
    assert int(vdb.d_atlas.to(torch.int64).sum()) == int((torch.arange(3000) % 251).sum())

    # ---- stripe gather: each rank paints its own stripes, rank 0 ends up with the whole frame ----------
    H, W = 100, 24   # 7 stripes, the last one partial
    frame = torch.zeros((H, W, 4), dtype=torch.uint8)
    for s in multi_gpu.stripes_of(H, world, rank):
        frame[16 * s:16 * s + 16] = 10 * (rank + 1) + s
    scratch = multi_gpu.gather_stripes(frame, world, rank, 0)
    multi_gpu.gather_stripes(frame, world, rank, 0, scratch)  # buffers are reused on the next frame
    if rank == 0:
        for s in range(7):
            assert int(frame[16 * s, 0, 0]) == 10 * (s % world + 1) + s
            assert (frame[16 * s:min(H, 16 * s + 16)] == frame[16 * s, 0, 0]).all()
    assert multi_gpu.split_range(10, 3, 0) == (0, 4) and multi_gpu.split_range(10, 3, 2) == (7, 10)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_and_gather(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(os.path.join(tmp_path, "bvh0.pt"))
    b = torch.load(os.path.join(tmp_path, "bvh1.pt"))
    for k in a:
        assert torch.equal(a[k], b[k]), k

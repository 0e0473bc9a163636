"""One process per GPU: a structure built once on one rank is broadcast to the others over RCCL
(torch.distributed, backend "nccl"), the image or ray buffer is split between ranks, and per-rank
tiles are gathered to rank 0. Nothing here is on the per-ray data path: there is one broadcast at
set-up and one gather per frame.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .lbvh import LBVH


def _bcast_meta(values, src, device):
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.broadcast(t, src=src)
    return t.cpu().tolist()


def write_lbvh_device_data(bvh: LBVH, root_aabb, face_num: int, compacted_num: int) -> None:
    """Rebuilds the 64-byte DeviceData of a received tree: device pointers are local to each rank."""
    dd = _lib.LbvhDeviceData()
    for k in range(3):
        dd.root_aabb_min[k] = float(root_aabb[k])
        dd.root_aabb_max[k] = float(root_aabb[3 + k])
    dd.face_num, dd.compacted_face_num = face_num, compacted_num
    dd.faces, dd.compacted_faces = bvh.d_faces.data_ptr(), bvh.d_compacted.data_ptr()
    dd.internal_nodes = bvh.d_nodes.data_ptr()
    raw = np.frombuffer(bytes(dd), dtype=np.uint8).copy()
    bvh.d_device_data = torch.from_numpy(raw).to(bvh.device)
    bvh._host_dd = None


def broadcast_lbvh(bvh: LBVH, src: int = 0) -> None:
    """Broadcasts the built tree and the mesh it indexes: faces (4N B), compactedFaces (4M),
    internalNodes (32(M-1)), positions (12V), triangle indices (12N)."""
    rank = dist.get_rank()
    dev = bvh.device
    if rank == src:
        dd = bvh.GetDeviceData()
        meta = [dd["faceNum"], bvh.d_positions.shape[0], dd["compactedFaceNum"]] + \
            [float(x) for x in dd["rootAABB"]]
    else:
        meta = [0.0] * 9
    meta = _bcast_meta(meta, src, dev)
    N, V, M = int(meta[0]), int(meta[1]), int(meta[2])
    root = np.array(meta[3:9], dtype=np.float64).astype(np.float32)
    if rank != src:
        bvh.face_num = N
        bvh.d_positions = torch.empty((V, 3), dtype=torch.float32, device=dev)
        bvh.d_face_pos_idx = torch.empty((N, 3), dtype=torch.int32, device=dev)
        bvh.d_faces = torch.empty(N, dtype=torch.int32, device=dev)
        bvh.d_compacted = torch.empty(N, dtype=torch.int32, device=dev)
        bvh.d_nodes = torch.empty(N * 32, dtype=torch.uint8, device=dev)
    nodes_used = max(M - 1, 1) * 32
    for t in (bvh.d_positions, bvh.d_face_pos_idx, bvh.d_faces, bvh.d_compacted[:M],
              bvh.d_nodes[:nodes_used]):
        dist.broadcast(t, src=src)
    if rank != src:
        write_lbvh_device_data(bvh, root, N, M)
        bvh._complete = True


def split_range(total: int, world: int, rank: int):
    """Contiguous share [begin, end) of `total` items for `rank`."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


# ---- volume path ---------------------------------------------------------------------------------------

def broadcast_vdb(vdb, src: int = 0) -> None:
    """Broadcasts a built Depth-Box VDB: the 176-byte plan, node pools, child pools, atlas map and the
    brick atlas; every rank then points its device-data struct at its own copies."""
    from .dpbx_vdb import DepthBoxVDB  # noqa: F401
    rank = dist.get_rank()
    dev = vdb.device
    plan_bytes = torch.zeros(C.sizeof(_lib.VdbPlan) + 8, dtype=torch.uint8, device=dev)
    if rank == src:
        raw = bytes(vdb.plan) + int(vdb.voxel_bytes).to_bytes(8, "little")
        plan_bytes = torch.from_numpy(np.frombuffer(raw, np.uint8).copy()).to(dev)
    dist.broadcast(plan_bytes, src=src)
    if rank != src:
        raw = plan_bytes.cpu().numpy().tobytes()
        vdb.plan = _lib.VdbPlan.from_buffer_copy(raw[:C.sizeof(_lib.VdbPlan)])
        vdb.voxel_bytes = int.from_bytes(raw[C.sizeof(_lib.VdbPlan):], "little")
        p = vdb.plan
        vdb.d_node_pools = torch.empty(int(p.node_num) * 32, dtype=torch.uint8, device=dev)
        vdb.d_child_pools = torch.empty(int(p.child_num), dtype=torch.int32, device=dev)
        vdb.d_atlas_brick_to_node = torch.empty(int(p.atlas_map_num), dtype=torch.int32, device=dev)
        vdb.d_atlas = torch.empty(int(p.atlas_voxels) * vdb.voxel_bytes, dtype=torch.uint8, device=dev)
    for t in (vdb.d_node_pools, vdb.d_child_pools, vdb.d_atlas_brick_to_node, vdb.d_atlas):
        dist.broadcast(t, src=src)
    if rank != src:
        dd = _lib.VdbDeviceData()
        dd.leaf_num, dd.voxel_bytes, dd.vdb_param = vdb.plan.leaf_num, vdb.voxel_bytes, vdb.plan.param
        vdb.device_data = dd
        vdb.rebind()
        vdb._complete = True


STRIPE_ROWS = 16


def stripes_of(height: int, world: int, rank: int):
    """Indices of the 16-row stripes rank `rank` renders: round-robin, so that empty borders and the
    dense centre of the picture are spread over all ranks."""
    n = (height + STRIPE_ROWS - 1) // STRIPE_ROWS
    return list(range(rank, n, world))


def gather_stripes(rgba: torch.Tensor, world: int, rank: int, dst: int = 0, scratch=None):
    """Gathers every rank's stripes of an [H, W, 4] uint8 frame into rank `dst`'s frame. One collective
    per frame; ranks with one stripe fewer send a padded block."""
    H, W = rgba.shape[0], rgba.shape[1]
    n = (H + STRIPE_ROWS - 1) // STRIPE_ROWS
    per = (n + world - 1) // world
    pad_h = n * STRIPE_ROWS
    if scratch is None:
        scratch = {}
    key = (H, W, world, rgba.device)
    if scratch.get("key") != key:
        scratch["key"] = key
        scratch["send"] = torch.zeros((per, STRIPE_ROWS, W, 4), dtype=torch.uint8, device=rgba.device)
        scratch["recv"] = ([torch.zeros((per, STRIPE_ROWS, W, 4), dtype=torch.uint8, device=rgba.device)
                            for _ in range(world)] if rank == dst else None)
        scratch["padded"] = torch.zeros((pad_h, W, 4), dtype=torch.uint8, device=rgba.device)
    send = scratch["send"]
    mine = stripes_of(H, world, rank)
    padded = scratch["padded"]
    padded[:H] = rgba
    view = padded.view(n, STRIPE_ROWS, W, 4)
    send[:len(mine)] = view[rank::world]
    if send.is_cuda and dist.get_backend() == "gloo":
        # gloo gathers host tensors only (rehearsals of the multi-rank path on one GPU)
        host = [torch.empty(send.shape, dtype=send.dtype) for _ in range(world)] if rank == dst else None
        dist.gather(send.cpu(), host, dst=dst)
        if rank == dst:
            for r in range(world):
                scratch["recv"][r].copy_(host[r])
    else:
        dist.gather(send, scratch["recv"], dst=dst)
    if rank == dst:
        for r in range(world):
            cnt = len(range(r, n, world))
            view[r::world] = scratch["recv"][r][:cnt]
        rgba.copy_(padded[:H])
    return scratch

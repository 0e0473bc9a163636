"""One process per GPU: a structure built once on one rank is broadcast to the others over RCCL, the
image or ray buffer is split between ranks, and per-rank stripes are gathered to rank 0. Nothing here
is on the per-ray data path: there are two collectives at set-up (sizes, then the structure's one
allocation) and one gather per frame, which runs beside the next frame's rendering.

Two transports carry the same calls: torch.distributed (backend "nccl" = RCCL; "gloo" for rehearsals on
the CPU or on one GPU) and the library's own RCCL communicator (`Communicator`, the kk_comm_* entry
points a C++ host uses)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .lbvh import LBVH

STRIPE_ROWS = 16


class Communicator:
    """The library's RCCL communicator (kk_comm_*). The 128-byte id made on rank 0 reaches the other
    ranks through torch.distributed's store here; a C++ host hands it over by its own means."""

    def __init__(self, rank: int, world: int, device):
        self.rank, self.world, self.device = rank, world, torch.device(device)
        l = _lib.lib()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _lib.check(l.kk_comm_unique_id(ident), "kk_comm_unique_id")
        if world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(l.kk_comm_init(world, rank, ident, C.byref(self.handle)), "kk_comm_init")

    def broadcast(self, t: torch.Tensor, src: int = 0) -> None:
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().kk_comm_broadcast(self.handle, t.data_ptr(), t.numel() * t.element_size(), src,
                                                    _lib.current_stream_ptr()), "kk_comm_broadcast")

    def broadcast_host(self, raw: bytes, src: int = 0) -> bytes:
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().kk_comm_broadcast_host(self.handle, buf, len(raw), src,
                                                         _lib.current_stream_ptr()), "kk_comm_broadcast_host")
        return bytes(buf)

    def gather(self, send: torch.Tensor, recv, dst: int = 0, stream_ptr=None) -> None:
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().kk_comm_gather(
                self.handle, send.data_ptr(), recv.data_ptr() if recv is not None else None,
                send.numel() * send.element_size(), dst,
                stream_ptr if stream_ptr is not None else _lib.current_stream_ptr()), "kk_comm_gather")

    def close(self) -> None:
        if self.handle:
            _lib.lib().kk_comm_destroy(self.handle)
            self.handle = C.c_void_p()


def _bcast_bytes(raw: bytes, src: int, device, comm: Communicator | None) -> bytes:
    """A few host bytes (sizes, a plan) from src to every rank."""
    if comm is not None:
        return comm.broadcast_host(raw, src)
    t = torch.from_numpy(np.frombuffer(raw, np.uint8).copy()).to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy().tobytes()


def _bcast_tensor(t: torch.Tensor, src: int, comm: Communicator | None) -> None:
    if comm is not None:
        comm.broadcast(t, src)
    else:
        dist.broadcast(t, src=src)


def bind_lbvh_blob(bvh: LBVH) -> None:
    """Points the DeviceData at the head of the tree's allocation at this rank's own arrays (device
    pointers do not travel)."""
    N, V = bvh.face_num, bvh.d_positions.shape[0]
    if bvh.d_blob.is_cuda:
        with torch.cuda.device(bvh.device):
            _lib.check(_lib.lib().kk_lbvh_blob_bind(bvh.d_blob.data_ptr(), N, V, _lib.current_stream_ptr()),
                       "kk_lbvh_blob_bind")
    else:  # host tensors (rehearsals of the multi-rank logic without a GPU)
        dd = _lib.LbvhDeviceData.from_buffer_copy(bvh.d_device_data.numpy().tobytes())
        dd.faces, dd.compacted_faces = bvh.d_faces.data_ptr(), bvh.d_compacted.data_ptr()
        dd.internal_nodes = bvh.d_nodes.data_ptr()
        bvh.d_device_data.copy_(torch.from_numpy(np.frombuffer(bytes(dd), np.uint8).copy()))
    bvh._host_dd = None


def broadcast_lbvh(bvh: LBVH, src: int = 0, comm: Communicator | None = None) -> None:
    """Broadcasts the built tree and the mesh it indexes: the sizes (16 bytes), then the tree's one
    allocation - DeviceData, positions (12V), triangle indices (12N), faces (4N), compactedFaces (4N),
    internalNodes (32N) - in a single collective. A receiving rank drops everything it had derived from
    an earlier tree (build workspace, trace-side repacking, host copy of the header)."""
    rank = comm.rank if comm is not None else dist.get_rank()
    raw = np.zeros(2, np.int64)
    if rank == src:
        raw[:] = (bvh.face_num, bvh.d_positions.shape[0])
    raw = np.frombuffer(_bcast_bytes(raw.tobytes(), src, bvh.device, comm), np.int64)
    N, V = int(raw[0]), int(raw[1])
    if rank != src:
        bvh._complete = False
        bvh._clear()
        bvh.alloc_blob(N, V)
    _bcast_tensor(bvh.d_blob, src, comm)
    bind_lbvh_blob(bvh)
    if rank != src:
        bvh._complete = True


def broadcast_scene(tracer, src: int = 0, comm: Communicator | None = None) -> None:
    """Broadcasts what RayTracer.SetMesh uploaded - triangles, triangle-to-position indices, positions,
    normals, texture coordinates, lights, materials - beside the tree: the sizes (32 bytes), then the
    scene's one allocation in a single collective; every rank's scene struct points into its own copy."""
    rank = comm.rank if comm is not None else dist.get_rank()
    raw = bytes(C.sizeof(_lib.PtSceneSizes))
    if rank == src:
        raw = bytes(tracer.scene_sizes)
    raw = _bcast_bytes(raw, src, tracer.device, comm)
    if rank != src:
        tracer.alloc_scene_blob(_lib.PtSceneSizes.from_buffer_copy(raw))
        tracer.host_arrays = None
    _bcast_tensor(tracer.d_scene_blob, src, comm)
    tracer._modified = True


def split_range(total: int, world: int, rank: int):
    """Contiguous share [begin, end) of `total` items for `rank`."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


# ---- volume path ---------------------------------------------------------------------------------------

def broadcast_vdb(vdb, src: int = 0, comm: Communicator | None = None) -> None:
    """Broadcasts a built Depth-Box VDB: the 176-byte plan (+ voxel size), then node pools, child pools,
    atlas map and brick atlas as the one allocation they live in; every rank's device-data struct points
    into its own copy."""
    rank = comm.rank if comm is not None else dist.get_rank()
    n = C.sizeof(_lib.VdbPlan) + 8
    raw = bytes(n)
    if rank == src:
        raw = bytes(vdb.plan) + int(vdb.voxel_bytes).to_bytes(8, "little")
    raw = _bcast_bytes(raw, src, vdb.device, comm)
    if rank != src:
        vdb._complete = False
        vdb._clear()
        plan = _lib.VdbPlan.from_buffer_copy(raw[:C.sizeof(_lib.VdbPlan)])
        vdb.alloc_blob(plan, int.from_bytes(raw[C.sizeof(_lib.VdbPlan):], "little"))
    _bcast_tensor(vdb.d_blob, src, comm)
    if rank != src:
        vdb._complete = True


def stripes_of(height: int, world: int, rank: int):
    """Indices of the 16-row stripes rank `rank` renders: round-robin, so that empty borders and the
    dense centre of the picture are spread over all ranks."""
    n = (height + STRIPE_ROWS - 1) // STRIPE_ROWS
    return list(range(rank, n, world))


def stripes_per_rank(height: int, world: int) -> int:
    n = (height + STRIPE_ROWS - 1) // STRIPE_ROWS
    return (n + world - 1) // world


def unstripe(blocks: torch.Tensor, world: int, rgba: torch.Tensor) -> None:
    """blocks uint8 [world, per, 16, W, 4] (rank r's j-th stripe is image stripe r + j * world) ->
    rgba uint8 [H, W, 4]."""
    H, W = rgba.shape[0], rgba.shape[1]
    if blocks.is_cuda:
        with torch.cuda.device(rgba.device):
            _lib.check(_lib.lib().kk_unstripe_rgba(blocks.data_ptr(), world, W, H, rgba.data_ptr(),
                                                   _lib.current_stream_ptr()), "kk_unstripe_rgba")
        return
    n = (H + STRIPE_ROWS - 1) // STRIPE_ROWS
    for s in range(n):
        rows = min(STRIPE_ROWS, H - STRIPE_ROWS * s)
        rgba[STRIPE_ROWS * s:STRIPE_ROWS * s + rows] = blocks[s % world, s // world, :rows]


class StripeGatherer:
    """Per-frame gather of the ranks' stripe blocks to rank `dst`, two frames deep: a rank renders frame
    i straight into send block i % 2 (compact stripes, no repacking), the gather of that block is issued
    without waiting for it, and frame i + 1 is rendered into the other block meanwhile. A block is waited
    for (on the stream, not on the host) only when its slot comes round again; on `dst` the received
    blocks are then put back into image order.

        g = StripeGatherer(W, H, world, rank, device)
        for each frame:  buf = g.begin(fb.rgba); render this rank's stripes into buf; g.submit()
        g.flush(fb.rgba)     # the last frame, assembled on dst
    """

    def __init__(self, W: int, H: int, world: int, rank: int, device, dst: int = 0,
                 comm: Communicator | None = None):
        self.W, self.H, self.world, self.rank, self.dst, self.comm = W, H, world, rank, dst, comm
        self.device = torch.device(device)
        per = stripes_per_rank(H, world)
        shape = (per, STRIPE_ROWS, W, 4)
        self.send = [torch.zeros(shape, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.blocks = ([torch.zeros((world,) + shape, dtype=torch.uint8, device=self.device) for _ in range(2)]
                       if rank == dst else [None, None])
        self.pending = [None, None]  # what to wait for before slot k is reused
        self.slot = 0
        self.frames = 0
        self.side = None
        if comm is not None and self.device.type == "cuda":
            self.side = torch.cuda.Stream(device=self.device)  # the gathers' own stream

    def _wait(self, k: int, rgba) -> None:
        p = self.pending[k]
        if p is None:
            return
        self.pending[k] = None
        if p == "done":
            pass
        elif isinstance(p, torch.cuda.Event):
            torch.cuda.current_stream(self.device).wait_event(p)
        else:
            p.wait()  # torch.distributed work: the current stream waits for the collective
        if self.rank == self.dst and rgba is not None:
            unstripe(self.blocks[k], self.world, rgba)

    def begin(self, rgba=None) -> torch.Tensor:
        """The block to render the next frame's stripes into. Waits (stream-side) for the gather that
        last used it; on dst that frame is assembled into `rgba`."""
        self._wait(self.slot, rgba)
        return self.send[self.slot]

    def submit(self) -> None:
        k = self.slot
        send, blocks = self.send[k], self.blocks[k]
        if self.comm is not None:
            cur = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            self.side.wait_event(ready)
            self.comm.gather(send, blocks, self.dst, stream_ptr=int(self.side.cuda_stream))
            done = torch.cuda.Event()
            done.record(self.side)
            self.pending[k] = done
        elif send.is_cuda and dist.get_backend() == "gloo":
            # gloo moves host tensors only (rehearsals of the multi-rank path on one GPU)
            host = [torch.empty(send.shape, dtype=send.dtype) for _ in range(self.world)] \
                if self.rank == self.dst else None
            dist.gather(send.cpu(), host, dst=self.dst)
            if self.rank == self.dst:
                for r in range(self.world):
                    blocks[r].copy_(host[r])
            self.pending[k] = "done"
        else:
            lst = [blocks[r] for r in range(self.world)] if self.rank == self.dst else None
            work = dist.gather(send, lst, dst=self.dst, async_op=True)
            self.pending[k] = work if work is not None else "done"
        self.slot ^= 1
        self.frames += 1

    def flush(self, rgba=None) -> None:
        """Waits for the gathers still out, oldest first; the newest frame ends up in `rgba` on dst."""
        self._wait(self.slot, rgba)
        self._wait(self.slot ^ 1, rgba)


def gather_stripes(rgba: torch.Tensor, world: int, rank: int, dst: int = 0, scratch=None):
    """One blocking gather of a frame whose stripes were rendered in place (each rank's stripes at their
    own rows of its `rgba`): packs this rank's stripes, gathers, assembles on dst. The frame loop uses
    StripeGatherer instead, which needs no packing and does not block."""
    H, W = rgba.shape[0], rgba.shape[1]
    if scratch is None:
        scratch = {}
    key = (H, W, world, rgba.device)
    if scratch.get("key") != key:
        scratch["key"] = key
        scratch["g"] = StripeGatherer(W, H, world, rank, rgba.device, dst)
    g = scratch["g"]
    buf = g.begin()
    for j, s in enumerate(stripes_of(H, world, rank)):
        rows = min(STRIPE_ROWS, H - STRIPE_ROWS * s)
        buf[j, :rows] = rgba[STRIPE_ROWS * s:STRIPE_ROWS * s + rows]
    g.submit()
    g.flush(rgba)
    return scratch

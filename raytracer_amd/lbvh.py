"""Host-side mirror of kouek::RayTracer::LBVH (include/ray_tracer/lbvh.h:20-78) for Python callers.

Method names follow the reference (BuildFrom, IsComplete, GetDeviceData, GetDeivceDataDevicePtr - the
misspelling is the reference's). Device memory is held in torch tensors; the build itself is
kk_lbvh_build in the C-ABI library.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np
import torch

from . import _lib

NODE_DTYPE = np.dtype([("lft", "<u4"), ("rht", "<u4"), ("aabb", "<f4", (6,))])
LEAF_MASK = 0x80000000


def _to_device(a, dtype, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype).contiguous()


class LBVH:
    StatTag = "[LBVH Status]"
    ErrTag = "[LBVH Error]"

    def __init__(self, device="cuda:0", verbose: bool = False):
        self.device = torch.device(device)
        self.verbose = verbose
        self._complete = False
        self._clear()

    def _clear(self):
        self.d_positions = self.d_face_pos_idx = None
        self.d_faces = self.d_compacted = self.d_nodes = None
        self.d_device_data = self.d_workspace = None
        self.d_blob = self.blob_layout = None
        self.d_accel = None
        self._accel_stale = True
        self._accel_checked = None
        self._host_dd = None
        self.ws_layout = None
        self.face_num = 0

    # -- reference API ---------------------------------------------------------------------------
    def BuildFrom(self, positions, face_position_indices, keep_stages: bool = False,
                  sync: bool = True) -> None:
        """positions float32 [V,3], face_position_indices uint32 [N,3] (numpy or torch). On failure a
        message is printed and IsComplete() stays False, as in the reference."""
        self._complete = False
        self._clear()
        try:
            self._build(positions, face_position_indices, keep_stages, sync)
        except _lib.KoukeError as e:
            print(f"{self.ErrTag} {e}", file=sys.stderr)
            return
        self._complete = True
        if self.verbose and sync:
            dd = self.GetDeviceData()
            print(f"{self.StatTag} Builded LBVH.\n\tcompact num: {dd['compactedFaceNum']}, "
                  f"face num: {dd['faceNum']}")

    def IsComplete(self) -> bool:
        return self._complete

    def GetDeivceDataDevicePtr(self) -> int:
        return 0 if self.d_device_data is None else self.d_device_data.data_ptr()

    def GetDeviceData(self) -> dict:
        """Host copy of the 64-byte DeviceData (synchronises)."""
        if self._host_dd is None:
            raw = self.d_device_data.cpu().numpy().tobytes()
            # the read-back has synchronised: the sort's failure word of the last (re)build is final
            if self.d_workspace is not None and self.ws_layout is not None:
                if int(self.stage("error_flag", 1, np.uint32)[0]) != 0:
                    self._complete = False
                    raise _lib.KoukeError("radix sort look-back gave up (device error flag set): "
                                          "the tree of the last build is not valid")
            dd = _lib.LbvhDeviceData.from_buffer_copy(raw)
            self._host_dd = {
                "rootAABB": np.array(list(dd.root_aabb_min) + list(dd.root_aabb_max), np.float32),
                "faceNum": int(dd.face_num), "compactedFaceNum": int(dd.compacted_face_num),
                "faces": int(dd.faces), "compactedFaces": int(dd.compacted_faces),
                "internalNodes": int(dd.internal_nodes),
            }
        return self._host_dd

    # -- implementation --------------------------------------------------------------------------
    def alloc_blob(self, N: int, V: int) -> None:
        """The tree and the mesh it indexes live in one allocation (so that they travel to other GPUs in
        one collective, kk_lbvh_blob_layout_of); the arrays are views into it."""
        lay = _lib.LbvhBlobLayout()
        _lib.check(_lib.lib().kk_lbvh_blob_layout_of(N, V, C.byref(lay)), "kk_lbvh_blob_layout_of")
        b = torch.empty(int(lay.total), dtype=torch.uint8, device=self.device)
        self.d_blob, self.blob_layout, self.face_num = b, lay, N
        piece = lambda off, nbytes: b[int(off):int(off) + nbytes]
        self.d_device_data = piece(lay.device_data, 64)
        self.d_device_data.zero_()
        self.d_positions = piece(lay.positions, 12 * V).view(torch.float32).reshape(V, 3)
        self.d_face_pos_idx = piece(lay.face_pos_idx, 12 * N).view(torch.int32).reshape(N, 3)
        self.d_faces = piece(lay.faces, 4 * N).view(torch.int32)
        self.d_compacted = piece(lay.compacted, 4 * N).view(torch.int32)
        self.d_nodes = piece(lay.nodes, 32 * N)

    def _build(self, positions, tri, keep_stages, sync):
        l = _lib.lib()
        dev = self.device
        pos_t = positions if isinstance(positions, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(positions, dtype=np.float32))
        tri_t = tri if isinstance(tri, torch.Tensor) else torch.from_numpy(
            np.ascontiguousarray(tri).astype(np.uint32, copy=False).view(np.int32))
        pos_t, tri_t = pos_t.reshape(-1, 3), tri_t.reshape(-1, 3)
        V, N = pos_t.shape[0], tri_t.shape[0]
        if N == 0 or V == 0:
            raise _lib.KoukeError("empty mesh: the build needs at least one face")
        # the kernels follow these indices into the positions: one past the array is a GPU memory fault
        # (as unsigned numbers, so that a negative int32 view of a large index counts as large)
        if tri_t.is_cuda:
            word = torch.empty(1, dtype=torch.int32, device=tri_t.device)
            with torch.cuda.device(tri_t.device):
                _lib.check(l.kk_check_indices(tri_t.contiguous().data_ptr(), 3 * N, V, word.data_ptr(),
                                              _lib.current_stream_ptr()), "kk_check_indices")
        else:
            top = int(tri_t.view(torch.int32).to(torch.int64).bitwise_and(0xFFFFFFFF).max())
            if top >= V:
                raise _lib.KoukeError(f"a face refers to position {top}, the mesh has {V}")
        self.alloc_blob(N, V)
        self.d_positions.copy_(pos_t.to(torch.float32))
        self.d_face_pos_idx.copy_(tri_t.view(torch.int32) if tri_t.dtype != torch.int32 else tri_t)
        flags = _lib.KK_LBVH_KEEP_STAGES if keep_stages else 0
        lay = _lib.LbvhWsLayout()
        _lib.check(l.kk_lbvh_workspace_layout(N, V, flags, C.byref(lay)), "kk_lbvh_workspace_layout")
        self.ws_layout = lay
        self.d_workspace = torch.empty(lay.total_bytes, dtype=torch.uint8, device=dev)
        self.rebuild(flags)
        if sync:
            torch.cuda.synchronize(dev)
            err = self.stage("error_flag", 1, np.uint32)
            if int(err[0]) != 0:
                raise _lib.KoukeError("radix sort look-back gave up (device error flag set)")

    def rebuild(self, flags: int = 0) -> None:
        """Enqueue the build again on the current stream over the same buffers (no allocation, no
        synchronisation)."""
        l = _lib.lib()
        self._host_dd = None
        self._accel_stale = True  # the trace-side repacking describes the previous tree
        self._accel_checked = None  # ... and its depth check too: depth follows the geometry, not (N, M)
        with torch.cuda.device(self.device):
            _lib.check(l.kk_lbvh_build(self.d_positions.data_ptr(), self.d_positions.shape[0],
                                       self.d_face_pos_idx.data_ptr(), self.face_num,
                                       self.d_faces.data_ptr(), self.d_compacted.data_ptr(),
                                       self.d_nodes.data_ptr(), self.d_device_data.data_ptr(),
                                       self.d_workspace.data_ptr(), self.d_workspace.numel(), flags,
                                       _lib.current_stream_ptr()), "kk_lbvh_build")

    def trace_accel(self, rebuild: bool = False):
        """The trace-side layout of this tree (64-byte records of four quantised boxes + 48-byte triangle records),
        built on first use on the current stream; `rebuild` repacks it over the same buffer."""
        l = _lib.lib()
        N, M = self.face_num, self.GetDeviceData()["compactedFaceNum"]
        if getattr(self, "d_accel", None) is None or self._accel_sizes != (N, M):
            self.d_accel = torch.empty(l.kk_trace_accel_bytes(N, M), dtype=torch.uint8, device=self.device)
            self._accel_sizes = (N, M)
            rebuild = True
        if rebuild or getattr(self, "_accel_stale", False):
            self._accel_stale = False
            with torch.cuda.device(self.device):
                # a tree built here still has its leaf boxes in the build workspace; a received one has not
                ws = getattr(self, "d_workspace", None)
                leaf_boxes = ws.data_ptr() + self.ws_layout.leaf_aabbs if ws is not None else None
                _lib.check(l.kk_trace_accel_build_with_leaf_boxes(
                    self.GetDeivceDataDevicePtr(), self.d_positions.data_ptr(), self.d_face_pos_idx.data_ptr(),
                    N, M, leaf_boxes, self.d_accel.data_ptr(), self.d_accel.numel(),
                    _lib.current_stream_ptr()), "kk_trace_accel_build_with_leaf_boxes")
                if not getattr(self, "_accel_checked", None) == (N, M):
                    # once per tree shape: the build's own status (depth the walks' stacks can hold)
                    _lib.check(l.kk_trace_accel_check(self.d_accel.data_ptr(), N, M,
                                                      _lib.current_stream_ptr()), "kk_trace_accel_check")
                    self._accel_checked = (N, M)
        return self.d_accel

    # -- read-back helpers (parity tests, logs) --------------------------------------------------------
    def stage(self, name: str, count: int, dtype) -> np.ndarray:
        off = getattr(self.ws_layout, name)
        nbytes = int(count) * np.dtype(dtype).itemsize
        return self.d_workspace[off:off + nbytes].cpu().numpy().view(dtype).copy()

    def to_host(self, keep_stages: bool = False) -> dict:
        dd = self.GetDeviceData()
        N, M = dd["faceNum"], dd["compactedFaceNum"]
        out = {
            "N": N, "M": M, "rootAABB": dd["rootAABB"],
            "faces": self.d_faces.cpu().numpy().view(np.uint32).copy(),
            "compactedFaces": self.d_compacted[:M].cpu().numpy().view(np.uint32).copy(),
            "internalNodes": self.d_nodes[:max(M - 1, 1) * 32].cpu().numpy().view(NODE_DTYPE).copy(),
            "sortedMortons": self.stage("sorted_mortons", N, np.uint32),
            "uniqueMortons": self.stage("unique_mortons", M, np.uint32),
            "leafAABBs": self.stage("leaf_aabbs", 6 * M, np.float32).reshape(M, 6),
        }
        if keep_stages:  # stages only a build with keep_stages=True writes
            out["mortons"] = self.stage("unsorted_mortons", N, np.uint32)
            out["nodeParents"] = self.stage("node_parents", max(M - 1, 0), np.uint32)
            out["leafParents"] = self.stage("leaf_parents", M, np.uint32)
        return out

"""Host-side mirror of kouek::RayCaster::DepthBoxVDB (include/ray_caster/dpbx_vdb.h:27-143).

BuildFrom keeps the reference's configuration (apronWid, log2Dims, useDepthBox, volume,
emptyScalarRngs); the CUDA texture argument becomes a `Volume` (linear device buffer + dimensions).
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np
import torch

from . import _lib

NODE_DTYPE = np.dtype([("pos", "<i4", (3,)), ("brickPosInAtlas", "<i4", (3,)),
                       ("childPoolOffsRelative", "<u8")])
UNDEF_RLT_IDX = 0xFFFFFFFF


class Volume:
    """Replaces the point-filtered CUDA::Texture of the volume: voxels T[dz][dy][dx], x fastest."""

    def __init__(self, data, device="cuda:0"):
        if isinstance(data, torch.Tensor):
            t = data
            assert t.dtype in (torch.uint8, torch.int16, torch.uint16)
            self.itemsize = 1 if t.dtype == torch.uint8 else 2
            self.tensor = t.to(device).contiguous()
        else:
            a = np.ascontiguousarray(data)
            assert a.dtype in (np.uint8, np.uint16) and a.ndim == 3
            self.itemsize = a.dtype.itemsize
            self.tensor = torch.from_numpy(a.view(np.uint8 if self.itemsize == 1 else np.int16)).to(device)
        self.device = torch.device(device)
        self.dz, self.dy, self.dx = (int(s) for s in self.tensor.shape)

    @property
    def np_dtype(self):
        return np.uint8 if self.itemsize == 1 else np.uint16


_PINNED = {}


def _pinned(name: str, nbytes: int) -> "torch.Tensor":
    """A page-locked host buffer of at least nbytes, kept per purpose and device and grown when needed
    (builds on one device are taken to follow one another, as the class is used)."""
    t = _PINNED.get(name)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True)
        _PINNED[name] = t
    return t[:nbytes]


class DepthBoxVDB:
    MaxLevelNum = 3
    MaxLog2Dim = 10
    ErrTag = "[DepthBoxVDB Error]"

    def __init__(self, device="cuda:0", verbose: bool = False):
        self.device = torch.device(device)
        self.verbose = verbose
        self._complete = False
        self._clear()

    def _clear(self):
        self.plan = None
        self.d_node_pools = self.d_child_pools = self.d_atlas_brick_to_node = self.d_atlas = None
        self.device_data = None
        self.d_blob = self.blob_layout = None
        self.voxel_bytes = 0
        self.timings = {}

    def BuildFrom(self, volume: Volume, empty_scalar_rngs, apron_wid: int = 1, log2_dims=(4, 4, 3),
                  use_depth_box: bool = True) -> None:
        self._complete = False
        self._clear()
        try:
            self._build(volume, empty_scalar_rngs, apron_wid, log2_dims, use_depth_box)
        except _lib.KoukeError as e:
            print(f"{self.ErrTag} {e}", file=sys.stderr)
            return
        self._complete = True
        if self.verbose:
            print(self.statistics_text())

    def memory_statistics(self) -> dict:
        """What the reference prints after a build (src/ray_caster/dpbx_vdb.cu:586-607): element counts and
        sizes of the pools, the atlas extent in voxels as the reference's 3-D array holds it, and the bytes
        in use (here the atlas stores the valid bricks only, one after the other)."""
        pm = self.plan.param
        e = int(pm.vox_per_atlas_brick)
        atlas_bytes = int(self.plan.atlas_voxels) * self.voxel_bytes
        return {"node_pools": (self.GetNodePoolsSize(), 32), "child_pools": (self.GetChildPoolsSize(), 4),
                "atlas_brick_to_node_indices": (self.GetAtlasBrickToNodeIndicesSize(), 4),
                "atlas_size": (int(pm.atlas_brick_per_vdb[0]) * e, int(pm.atlas_brick_per_vdb[1]) * e,
                               int(pm.atlas_brick_per_vdb[2]) * e, self.voxel_bytes),
                "atlas_bricks_stored": int(self.plan.leaf_num),
                "total_bytes": (self.GetNodePoolsSize() * 32 + self.GetChildPoolsSize() * 4 +
                                self.GetAtlasBrickToNodeIndicesSize() * 4 + atlas_bytes)}

    def statistics_text(self) -> str:
        m = self.memory_statistics()
        a = m["atlas_size"]
        return ("Build VDB with\n"
                f"\tNode pools of size {m['node_pools'][0]}x{m['node_pools'][1]}\n"
                f"\tChild pools of size {m['child_pools'][0]}x{m['child_pools'][1]}\n"
                f"\tMaps from atlas brick to node index of size {m['atlas_brick_to_node_indices'][0]}x4\n"
                f"\tAtlas of size {a[0]}x{a[1]}x{a[2]}x{a[3]} ({m['atlas_bricks_stored']} bricks of "
                f"{int(self.plan.param.vox_per_atlas_brick)}^3 stored)\n"
                f"\tTotal use of memory {m['total_bytes']}")

    def IsComplete(self) -> bool:
        return self._complete

    def GetDeviceData(self):
        """Host copy of the 176-byte device-data struct (its pointers are device pointers)."""
        return self.device_data

    def GetNodePoolsSize(self) -> int:
        return 0 if self.plan is None else int(self.plan.node_num)

    def GetChildPoolsSize(self) -> int:
        return 0 if self.plan is None else int(self.plan.child_num)

    def GetAtlasBrickToNodeIndicesSize(self) -> int:
        return 0 if self.plan is None else int(self.plan.atlas_map_num)

    # -- implementation ------------------------------------------------------------------------------
    def _build(self, vol: Volume, rngs, apron_wid, log2_dims, use_depth_box):
        l = _lib.lib()
        dev = self.device
        sfx = "u8" if vol.itemsize == 1 else "u16"
        rng = np.ascontiguousarray(np.asarray(rngs, vol.np_dtype).reshape(-1, 2))
        rng_p = rng.ctypes.data_as(C.c_void_p)
        l2 = (C.c_uint8 * 3)(*[int(x) for x in log2_dims])
        for x in log2_dims:
            if not (0 < int(x) <= self.MaxLog2Dim):
                raise _lib.KoukeError("log2Dim > 0 && log2Dim <= MaxLog2Dim should be true.")
        B = 1 << int(log2_dims[0])
        slots = [(d + B - 1) // B for d in (vol.dx, vol.dy, vol.dz)]
        n_slots = slots[0] * slots[1] * slots[2]
        stream = _lib.current_stream_ptr()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        import time
        tw = [time.perf_counter()]
        with torch.cuda.device(dev):
            d_valid = torch.empty(n_slots, dtype=torch.int32, device=dev)
            ev[0].record()
            _lib.check(getattr(l, f"kk_vdb_classify_{sfx}")(
                vol.tensor.data_ptr(), vol.dx, vol.dy, vol.dz, int(log2_dims[0]), rng_p, rng.shape[0],
                d_valid.data_ptr(), stream), "kk_vdb_classify")
            ev[1].record()
            # read back through page-locked memory kept from build to build (a pageable copy of these
            # 100 KB costs twice the time)
            pinned = _pinned(f"valid@{dev}", n_slots * 4)
            pinned.copy_(d_valid.view(torch.uint8), non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            h_valid = pinned.numpy().view(np.uint32)
            tw.append(time.perf_counter())
            plan = _lib.VdbPlan()
            _lib.check(l.kk_vdb_plan_build(vol.dx, vol.dy, vol.dz, l2, int(apron_wid), int(use_depth_box),
                                           h_valid.ctypes.data_as(C.c_void_p), C.byref(plan)),
                       "kk_vdb_plan_build")
            tw.append(time.perf_counter())
            # The atlas fill starts at once, with the leaves' positions made on the GPU from the flags
            # (kk_vdb_leaf_nodes), and runs while the host wires the pools. The three small arrays are
            # wired where they lie in the blob (they come before the atlas), so one copy carries them over.
            self.alloc_blob(plan, vol.itemsize)
            lay = self.blob_layout
            leaf_nodes = torch.empty(int(plan.leaf_num) * 32, dtype=torch.uint8, device=dev)
            _lib.check(l.kk_vdb_leaf_nodes(d_valid.data_ptr(), C.byref(plan), leaf_nodes.data_ptr(), stream),
                       "kk_vdb_leaf_nodes")
            ev[2].record()
            _lib.check(getattr(l, f"kk_vdb_fill_atlas_{sfx}")(
                vol.tensor.data_ptr(), vol.dx, vol.dy, vol.dz, C.byref(plan), leaf_nodes.data_ptr(),
                rng_p, rng.shape[0], self.d_atlas.data_ptr(), stream), "kk_vdb_fill_atlas")
            ev[3].record()
            tw.append(time.perf_counter())
            head_t = _pinned(f"head@{dev}", int(lay.atlas))
            head = head_t.numpy()
            head[:] = 0
            nodes = head[int(lay.node_pools):int(lay.node_pools) + int(plan.node_num) * 32].view(NODE_DTYPE)
            children = head[int(lay.child_pools):int(lay.child_pools) + int(plan.child_num) * 4].view(np.uint32)
            amap = head[int(lay.atlas_map):int(lay.atlas_map) + int(plan.atlas_map_num) * 4].view(np.uint32)
            _lib.check(l.kk_vdb_wire(C.byref(plan), h_valid.ctypes.data_as(C.c_void_p),
                                     nodes.ctypes.data_as(C.c_void_p), children.ctypes.data_as(C.c_void_p),
                                     amap.ctypes.data_as(C.c_void_p)), "kk_vdb_wire")
            tw.append(time.perf_counter())
            self.d_blob[:int(lay.atlas)].copy_(head_t, non_blocking=True)
            torch.cuda.synchronize(dev)
        # (the host arrays kept below must not alias the page-locked buffers of the next build)
        nodes, children, amap = nodes.copy(), children.copy(), amap.copy()
        h_valid = None
        tw.append(time.perf_counter())
        self.timings = {"classify_ms": ev[0].elapsed_time(ev[1]), "fill_atlas_ms": ev[2].elapsed_time(ev[3])}
        for k, name in enumerate(("classify_and_readback", "plan", "alloc_and_launch_fill", "wire",
                                  "upload_and_sync")):
            self.timings["host_" + name + "_ms"] = (tw[k + 1] - tw[k]) * 1e3
        self.host_nodes, self.host_children, self.host_atlas_map = nodes, children, amap

    def alloc_blob(self, plan, voxel_bytes: int) -> None:
        """Node pools, child pools, atlas map and atlas in one allocation (one collective carries them to
        another GPU, kk_vdb_blob_layout_of); the arrays are views, the device-data struct points into it."""
        lay = _lib.VdbBlobLayout()
        _lib.check(_lib.lib().kk_vdb_blob_layout_of(C.byref(plan), int(voxel_bytes), C.byref(lay)),
                   "kk_vdb_blob_layout_of")
        b = torch.empty(int(lay.total), dtype=torch.uint8, device=self.device)
        self.d_blob, self.blob_layout = b, lay
        self.plan, self.voxel_bytes = plan, int(voxel_bytes)
        piece = lambda off, nbytes: b[int(off):int(off) + nbytes]
        self.d_node_pools = piece(lay.node_pools, int(plan.node_num) * 32)
        self.d_child_pools = piece(lay.child_pools, int(plan.child_num) * 4).view(torch.int32)
        self.d_atlas_brick_to_node = piece(lay.atlas_map, int(plan.atlas_map_num) * 4).view(torch.int32)
        self.d_atlas = piece(lay.atlas, int(plan.atlas_voxels) * int(voxel_bytes))
        dd = _lib.VdbDeviceData()
        _lib.check(_lib.lib().kk_vdb_blob_device_data(b.data_ptr(), C.byref(plan), int(voxel_bytes), C.byref(dd)),
                   "kk_vdb_blob_device_data")
        self.device_data = dd

    def atlas_to_host(self) -> np.ndarray:
        A = int(self.plan.param.vox_per_atlas_brick)
        a = self.d_atlas.cpu().numpy().view(np.uint8 if self.voxel_bytes == 1 else np.uint16)
        return a.reshape(int(self.plan.leaf_num), A, A, A)

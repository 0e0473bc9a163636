"""ctypes binding of lib/libkouek_hip.so. There is no fallback: if the library is missing or a call
fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# (KK_LIB_PATH: another build of the library, for comparing two builds on one machine)
LIB_PATH = os.environ.get("KK_LIB_PATH") or os.path.join(_HERE, "lib", "libkouek_hip.so")


class KoukeError(RuntimeError):
    pass


class LbvhWsLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("total_bytes", "sorted_mortons", "unsorted_mortons",
                                          "unique_mortons", "leaf_aabbs", "node_parents",
                                          "leaf_parents", "error_flag")]


class LbvhDeviceData(C.Structure):
    _fields_ = [("root_aabb_min", C.c_float * 3), ("root_aabb_max", C.c_float * 3),
                ("face_num", C.c_uint32), ("compacted_face_num", C.c_uint32),
                ("faces", C.c_uint64), ("compacted_faces", C.c_uint64),
                ("internal_nodes", C.c_uint64), ("_pad", C.c_uint64)]


assert C.sizeof(LbvhDeviceData) == 64


class VdbParam(C.Structure):
    _fields_ = [("root_lev", C.c_uint8), ("apron_wid", C.c_uint8), ("apron_dep_wid", C.c_uint8),
                ("log2_dims", C.c_uint8 * 3), ("dims", C.c_int32 * 3), ("vox_per_atlas_brick", C.c_int32),
                ("min_dep_pos_val_in_brick", C.c_int32), ("max_dep_pos_val_in_brick", C.c_int32),
                ("atlas_brick_per_vdb", C.c_int32 * 3), ("node_starts", C.c_uint64 * 3),
                ("child_starts", C.c_uint64 * 3), ("voxs_per_child", C.c_float * 3),
                ("vox_per_vdb", C.c_float), ("vox_per_vol", C.c_float * 3), ("_pad", C.c_uint32)]


class VdbDeviceData(C.Structure):
    _fields_ = [("node_pools", C.c_void_p), ("child_pools", C.c_void_p),
                ("atlas_brick_to_node_indices", C.c_void_p), ("atlas", C.c_void_p),
                ("leaf_num", C.c_uint64), ("voxel_bytes", C.c_uint64), ("vdb_param", VdbParam)]


class VdbPlan(C.Structure):
    _fields_ = [("param", VdbParam), ("leaf_slots", C.c_uint32 * 3), ("leaf_num", C.c_uint32),
                ("node_num", C.c_uint64), ("child_num", C.c_uint64), ("atlas_map_num", C.c_uint64),
                ("atlas_voxels", C.c_uint64)]


class DvrParams(C.Structure):
    _fields_ = [("w2s", C.c_float * 16), ("inv_proj", C.c_float * 16), ("max_alpha", C.c_float),
                ("sampling_rate", C.c_float), ("use_shading", C.c_int32), ("ka", C.c_float),
                ("kd", C.c_float), ("ks", C.c_float), ("shiness", C.c_float),
                ("light_pos", C.c_float * 3), ("light_col", C.c_float * 3)]


class DvrFrame(C.Structure):
    _fields_ = [("eye_pos", C.c_float * 3), ("eye_rot", C.c_float * 9)]


class LbvhBlobLayout(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("device_data", "positions", "face_pos_idx", "faces", "compacted",
                                          "nodes", "total")]


class PtSceneSizes(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("face_num", "vert_num", "normal_num", "tex_coord_num", "light_num",
                                          "material_num")] + [("light_tot_area", C.c_float), ("_pad", C.c_uint32)]


class PtSceneBlobLayout(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("triangles", "tri_to_position_indices", "positions", "normals",
                                          "tex_coords", "lights", "materials", "total")]


class VdbBlobLayout(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("node_pools", "child_pools", "atlas_map", "atlas", "total")]


assert C.sizeof(VdbParam) == 128 and C.sizeof(VdbDeviceData) == 176 and C.sizeof(VdbPlan) == 176

KK_LBVH_KEEP_STAGES = 1
KK_TRAVERSE_ORDERED = 0
KK_TRACE_UNSORTED = 1
KK_TRACE_REUSE_ORDER = 2
KK_TRACE_WATERTIGHT = 4
KK_TRACE_ENTRY_BOUND = 8

_lib = None


def build_library(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j8"])
    return LIB_PATH


def _declare(l):
    vp, u32, sz, i32 = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int
    l.kk_last_error.restype = C.c_char_p
    l.kk_build_info.restype = C.c_char_p
    l.kk_sort_workspace_bytes.restype = sz
    l.kk_sort_workspace_bytes.argtypes = [u32]
    l.kk_sort_pairs_u32.argtypes = [vp, vp, vp, vp, i32, u32, vp, sz, vp]
    l.kk_sort_workspace_error_offset.restype = sz
    l.kk_sort_workspace_error_offset.argtypes = [u32]
    l.kk_sort_pairs_u32_static.argtypes = [vp, vp, vp, vp, i32, u32, vp]
    l.kk_sort_workspace_status.argtypes = [vp, u32, vp]
    l.kk_sort_static_status.argtypes = [vp]
    l.kk_sort_static_init.argtypes = []
    l.kk_sort_set_ticketed.argtypes = [i32]
    l.kk_debug_sort_force_error.argtypes = [i32]
    l.kk_closest_hit_accel_status.argtypes = [vp, u32, vp]
    l.kk_dvr_tile_order_status.argtypes = [vp, u32, vp]
    l.kk_check_indices.argtypes = [vp, sz, u32, vp, vp]
    l.kk_lbvh_workspace_layout.argtypes = [u32, u32, u32, C.POINTER(LbvhWsLayout)]
    l.kk_lbvh_build.argtypes = [vp, u32, vp, u32, vp, vp, vp, vp, vp, sz, u32, vp]
    l.kk_closest_hit_workspace_bytes.restype = sz
    l.kk_closest_hit.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp, i32, vp, vp, sz, vp]
    l.kk_trace_accel_bytes.restype = sz
    l.kk_trace_accel_bytes.argtypes = [u32, u32]
    l.kk_trace_accel_layout.argtypes = [u32, u32, vp]
    l.kk_trace_accel_build.argtypes = [vp, vp, vp, u32, u32, vp, sz, vp]
    l.kk_trace_accel_build_with_leaf_boxes.argtypes = [vp, vp, vp, u32, u32, vp, vp, sz, vp]
    l.kk_trace_accel_check.argtypes = [vp, u32, u32, vp]
    l.kk_pt_set_work_counters.argtypes = [vp]
    l.kk_closest_hit_accel_workspace_bytes.restype = sz
    l.kk_closest_hit_accel_workspace_bytes.argtypes = [u32]
    l.kk_closest_hit_accel.argtypes = [vp, u32, u32, vp, u32, vp, vp, vp, u32, vp, vp, sz, vp]
    l.kk_pt_render_sample.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, u32, vp]
    l.kk_pt_render_samples.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, i32, u32, vp]
    l.kk_pt_render_stripes_compact.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, i32, u32, vp]
    l.kk_pt_scene_blob_layout_of.argtypes = [vp, C.POINTER(PtSceneBlobLayout)]
    l.kk_pt_scene_blob_bind.argtypes = [vp, vp, vp]
    l.kk_pt_scene_broadcast.argtypes = [vp, vp, vp, i32, vp]
    l.kk_rt_render_target.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, i32, vp]
    l.kk_rng_uniforms.argtypes = [u32, u32, u32, u32, vp, vp]
    l.kk_stream_create.argtypes = [C.POINTER(vp)]
    l.kk_stream_destroy.argtypes = [vp]
    for sfx in ("u8", "u16"):
        getattr(l, f"kk_vdb_classify_{sfx}").argtypes = [vp, u32, u32, u32, u32, vp, u32, vp, vp]
        getattr(l, f"kk_vdb_fill_atlas_{sfx}").argtypes = [vp, u32, u32, u32, vp, vp, vp, u32, vp, vp]
        getattr(l, f"kk_dvr_render_{sfx}").argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
        getattr(l, f"kk_raw_render_{sfx}").argtypes = [vp, u32, u32, u32, vp, vp, vp, i32, vp, i32, i32, i32, i32, vp, vp]
        getattr(l, f"kk_dvr_render_stripes_{sfx}").argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
        getattr(l, f"kk_dvr_render_target_{sfx}").argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, i32, i32, i32, i32, vp]
    for sfx in ("u8", "u16"):
        getattr(l, f"kk_dvr_render_stripes_compact_{sfx}").argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
        getattr(l, f"kk_dvr_render_ordered_{sfx}").argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
        getattr(l, f"kk_dvr_render_stripes_compact_ordered_{sfx}").argtypes = [
            vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    l.kk_dvr_stripe_tile_count.restype = u32
    l.kk_dvr_stripe_tile_count.argtypes = [i32, i32, i32, i32]
    l.kk_dvr_tile_count.restype = u32
    l.kk_dvr_tile_count.argtypes = [i32, i32]
    l.kk_dvr_tile_order_workspace_bytes.restype = sz
    l.kk_dvr_tile_order_workspace_bytes.argtypes = [u32]
    l.kk_dvr_tile_order.argtypes = [vp, u32, vp, vp, sz, vp]
    l.kk_comm_unique_id.argtypes = [vp]
    l.kk_comm_init.argtypes = [i32, i32, vp, C.POINTER(vp)]
    l.kk_comm_destroy.argtypes = [vp]
    l.kk_comm_rank.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    l.kk_comm_broadcast.argtypes = [vp, vp, sz, i32, vp]
    l.kk_comm_broadcast_host.argtypes = [vp, vp, sz, i32, vp]
    l.kk_comm_gather.argtypes = [vp, vp, vp, sz, i32, vp]
    l.kk_lbvh_blob_layout_of.argtypes = [u32, u32, C.POINTER(LbvhBlobLayout)]
    l.kk_lbvh_blob_bind.argtypes = [vp, u32, u32, vp]
    l.kk_lbvh_broadcast.argtypes = [vp, vp, u32, u32, i32, vp]
    l.kk_vdb_blob_layout_of.argtypes = [vp, u32, C.POINTER(VdbBlobLayout)]
    l.kk_vdb_blob_device_data.argtypes = [vp, vp, u32, vp]
    l.kk_vdb_broadcast.argtypes = [vp, vp, vp, u32, i32, vp]
    l.kk_stripes_per_rank.argtypes = [i32, i32]
    l.kk_gather_stripes.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    l.kk_unstripe_rgba.argtypes = [vp, i32, i32, i32, vp, vp]
    l.kk_vdb_plan_build.argtypes = [u32, u32, u32, vp, C.c_uint8, i32, vp, vp]
    l.kk_vdb_wire.argtypes = [vp, vp, vp, vp, vp]
    l.kk_vdb_leaf_nodes.argtypes = [vp, vp, vp, vp]
    for name in ("kk_malloc", "kk_free", "kk_memcpy_h2d", "kk_memcpy_d2h", "kk_memset",
                 "kk_stream_synchronize", "kk_set_device"):
        getattr(l, name).restype = i32


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KoukeError(
                f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                "(or __graft_entry__.build()). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().kk_last_error().decode(errors="replace")
        raise KoukeError(f"{what} failed ({rc}): {msg}")


def current_stream_ptr() -> int:
    import torch
    return int(torch.cuda.current_stream().cuda_stream)

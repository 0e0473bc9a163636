"""TEST INFRASTRUCTURE ONLY: ctypes access to the CPU checker (oracle/_build/libkkoracle.so) and,
when present, to the pieces of the reference compiled as they are (oracle/_ref/libkkref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
shipped package (raytracer_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libkkoracle.so")
_REF = os.path.join(_HERE, "_ref", "libkkref.so")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")

INODE_DTYPE = np.dtype([("lft", "<u4"), ("rht", "<u4"), ("aabb", "<f4", (6,))])
assert INODE_DTYPE.itemsize == 32
LEAF_MASK = np.uint32(0x80000000)
FLT_MAX = np.float32(np.finfo(np.float32).max)


def build(force: bool = False) -> None:
    """Compile the checker (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB) or not os.path.exists(_REF):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
    return _lib


def ref():
    """The compiled reference pieces, or None when they were never built on this tree."""
    global _ref
    if _ref is None and os.path.exists(_REF):
        _ref = C.CDLL(_REF)
    return _ref


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def lbvh_build(positions: np.ndarray, tri: np.ndarray, refit_mode: int = 1) -> dict:
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    tri = np.ascontiguousarray(tri, dtype=np.uint32).reshape(-1, 3)
    V, N = pos.shape[0], tri.shape[0]
    out = {
        "rootAABB": np.zeros(6, np.float32),
        "mortons": np.zeros(N, np.uint32),
        "sortedMortons": np.zeros(N, np.uint32),
        "faces": np.zeros(N, np.uint32),
        "compactedFaces": np.zeros(N, np.uint32),
        "uniqueMortons": np.zeros(N, np.uint32),
        "leafAABBs": np.zeros((N, 6), np.float32),
        "internalNodes": np.zeros(max(N, 1), INODE_DTYPE),
    }
    m = C.c_uint32(0)
    rc = lib().ko_lbvh_build(_vp(pos), C.c_uint32(V), _vp(tri), C.c_uint32(N), _vp(out["rootAABB"]),
                             _vp(out["mortons"]), _vp(out["sortedMortons"]), _vp(out["faces"]),
                             _vp(out["compactedFaces"]), _vp(out["uniqueMortons"]),
                             _vp(out["leafAABBs"]), _vp(out["internalNodes"]), C.c_int(refit_mode),
                             C.byref(m))
    if rc != 0:
        raise ValueError("oracle lbvh_build rejected the input (N == 0?)")
    M = int(m.value)
    out["M"] = M
    out["N"] = N
    out["compactedFaces"] = out["compactedFaces"][:M].copy()
    out["uniqueMortons"] = out["uniqueMortons"][:M].copy()
    out["leafAABBs"] = out["leafAABBs"][:M].copy()
    out["internalNodes"] = out["internalNodes"][:max(M - 1, 0)].copy()
    return out


def closest_hit(bvh: dict, positions: np.ndarray, tri: np.ndarray, rays: np.ndarray):
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    tri = np.ascontiguousarray(tri, dtype=np.uint32).reshape(-1, 3)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    R = rays.shape[0]
    t = np.zeros(R, np.float32)
    bary = np.zeros((R, 2), np.float32)
    prim = np.zeros(R, np.uint32)
    stats = np.zeros(2, np.uint64)
    lib().ko_closest_hit(_vp(bvh["rootAABB"]), C.c_uint32(bvh["N"]), C.c_uint32(bvh["M"]),
                         _vp(bvh["faces"]), _vp(bvh["compactedFaces"]), _vp(bvh["internalNodes"]),
                         _vp(pos), _vp(tri), _vp(rays), C.c_uint32(R), _vp(t), _vp(bary), _vp(prim),
                         _vp(stats))
    return t, bary, prim, {"node_fetches": int(stats[0]), "tri_tests": int(stats[1])}


def check_tree_invariants(bvh: dict) -> int:
    """The reference's own structural self-check (include/ray_tracer/test_options.h:23-96): walk from
    node 0, every internal node, leaf and face must be reached exactly once. Returns the tree height
    counted the way that check counts it."""
    M, N = bvh["M"], bvh["N"]
    nodes = bvh["internalNodes"]
    seen_i = np.zeros(max(M - 1, 0), np.int32)
    seen_l = np.zeros(M, np.int32)
    seen_f = np.zeros(N, np.int32)
    height = 0
    if M > 1:
        level = [0]
        while level:
            height += 1
            nxt = []
            for n in level:
                seen_i[n] += 1
                for ch in (int(nodes["lft"][n]), int(nodes["rht"][n])):
                    if ch & 0x80000000:
                        c = ch & 0x7FFFFFFF
                        seen_l[c] += 1
                        beg = int(bvh["compactedFaces"][c])
                        end = N if c == M - 1 else int(bvh["compactedFaces"][c + 1])
                        for k in range(beg, end):
                            seen_f[bvh["faces"][k]] += 1
                    else:
                        nxt.append(ch)
            level = nxt
        assert (seen_i == 1).all(), "an internal node is not reached exactly once"
        assert (seen_l == 1).all(), "a leaf is not reached exactly once"
        assert (seen_f == 1).all(), "a face is not reached exactly once"
    return height


# ---- single primitives, for the glm cross-check -------------------------------------------------

def _call(fn, *arrays_and_out):
    fn(*[_vp(a) for a in arrays_and_out])
    return arrays_and_out[-1]


def hit_triangle(ray6, tri9, which="oracle"):
    out = np.zeros(3, np.float32)
    l = lib() if which == "oracle" else ref()
    f = l.ko_hit_triangle if which == "oracle" else l.kkref_hit_triangle
    return _call(f, np.ascontiguousarray(ray6, np.float32), np.ascontiguousarray(tri9, np.float32), out)


def hit_aabb(ray6, aabb6, which="oracle"):
    out = np.zeros(2, np.float32)
    l = lib() if which == "oracle" else ref()
    f = l.ko_hit_aabb if which == "oracle" else l.kkref_hit_aabb
    return _call(f, np.ascontiguousarray(ray6, np.float32), np.ascontiguousarray(aabb6, np.float32), out)


# ---- path tracer ---------------------------------------------------------------------------------------

class _KoPtScene(C.Structure):
    _fields_ = [("inodes", C.c_void_p), ("faces", C.c_void_p), ("compactedFaces", C.c_void_p),
                ("faceNum", C.c_uint32), ("compactedFaceNum", C.c_uint32), ("positions", C.c_void_p),
                ("facePosIdx", C.c_void_p), ("normals", C.c_void_p), ("triangles", C.c_void_p),
                ("materials", C.c_void_p), ("lights", C.c_void_p), ("lightNum", C.c_uint32),
                ("lightTotArea", C.c_float), ("maxPathDepth", C.c_uint32), ("w2s", C.c_float * 16),
                ("invProj", C.c_float * 16), ("eyePos", C.c_float * 3), ("eyeRot", C.c_float * 9)]


def pt_render_sample(bvh: dict, arrays: dict, w2s, inv_proj, eye_pos, eye_rot, accum: np.ndarray,
                     rgba: np.ndarray, sample_idx: float, seed: int = 1234, max_path_depth: int = 4,
                     rows=None) -> None:
    """One Render(Scene) call on the CPU. arrays: triangles / tri / positions / normals / lights /
    materials as numpy arrays in the reference's struct layouts. accum [H,W,3] f32 and rgba [H,W,4]
    u8 are updated in place."""
    H, W = accum.shape[:2]
    sc = _KoPtScene()
    keep = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    sc.inodes, sc.faces = _vp(bvh["internalNodes"]).value, _vp(bvh["faces"]).value
    sc.compactedFaces = _vp(bvh["compactedFaces"]).value
    sc.faceNum, sc.compactedFaceNum = bvh["N"], bvh["M"]
    sc.positions, sc.facePosIdx = _vp(keep["positions"]).value, _vp(keep["tri"]).value
    sc.normals, sc.triangles = _vp(keep["normals"]).value, _vp(keep["triangles"]).value
    sc.materials = _vp(keep["materials"]).value
    sc.lights = _vp(keep["lights"]).value if keep["lights"].size else None
    sc.lightNum = int(keep["lights"].size)
    tot = np.float32(0)
    for a in keep["lights"]["area"]:
        tot = np.float32(tot + a)
    sc.lightTotArea = float(tot)
    sc.maxPathDepth = max_path_depth
    for name, val in (("w2s", w2s), ("invProj", inv_proj), ("eyePos", eye_pos), ("eyeRot", eye_rot)):
        arr = getattr(sc, name)
        for i, x in enumerate(np.asarray(val, np.float32).reshape(-1)):
            arr[i] = float(x)
    y0, y1 = rows if rows is not None else (0, H)
    assert accum.flags.c_contiguous and rgba.flags.c_contiguous
    lib().ko_pt_render_sample(C.byref(sc), _vp(accum), _vp(rgba), C.c_int(W), C.c_int(H), C.c_int(y0),
                              C.c_int(y1), C.c_float(sample_idx), C.c_uint32(seed))


RT_TARGETS = ("Scene", "AABBs", "Triangles", "Lights", "Positions", "Normals", "TextureCoords")


def rt_render_target(bvh: dict, arrays: dict, w2s, inv_proj, eye_pos, eye_rot, W: int, H: int,
                     target: str, display_tree_height: int = 0, rows=None) -> np.ndarray:
    """One Render call of an inspection target on the CPU -> rgba [H,W,4] u8."""
    sc = _KoPtScene()
    keep = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    sc.inodes, sc.faces = _vp(bvh["internalNodes"]).value, _vp(bvh["faces"]).value
    sc.compactedFaces = _vp(bvh["compactedFaces"]).value
    sc.faceNum, sc.compactedFaceNum = bvh["N"], bvh["M"]
    sc.positions, sc.facePosIdx = _vp(keep["positions"]).value, _vp(keep["tri"]).value
    sc.normals, sc.triangles = _vp(keep["normals"]).value, _vp(keep["triangles"]).value
    sc.materials = _vp(keep["materials"]).value
    sc.lights = _vp(keep["lights"]).value if keep["lights"].size else None
    sc.lightNum = int(keep["lights"].size)
    for name, val in (("w2s", w2s), ("invProj", inv_proj), ("eyePos", eye_pos), ("eyeRot", eye_rot)):
        arr = getattr(sc, name)
        for i, x in enumerate(np.asarray(val, np.float32).reshape(-1)):
            arr[i] = float(x)
    root = np.ascontiguousarray(bvh["rootAABB"], np.float32)
    rgba = np.zeros((H, W, 4), np.uint8)
    y0, y1 = rows if rows is not None else (0, H)
    lib().ko_rt_render_target(C.byref(sc), _vp(root), C.c_int(display_tree_height),
                              C.c_int(RT_TARGETS.index(target)), _vp(rgba), C.c_int(W), C.c_int(H),
                              C.c_int(y0), C.c_int(y1))
    return rgba


def rng_uniforms(seed: int, pixel: int, sample: int, count: int) -> np.ndarray:
    out = np.zeros(count, np.float32)
    lib().ko_rng_uniforms(C.c_uint32(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(count),
                          _vp(out))
    return out


# ---- Depth-Box VDB and the DVR marcher ----------------------------------------------------------------

VDB_PARAM_DTYPE = np.dtype({
    "names": ["rootLev", "apronWid", "apronDepWid", "log2Dims", "dims", "voxPerAtlasBrick",
              "minDepPosValInBrick", "maxDepPosValInBrick", "atlasBrickPerVDB", "nodeStarts",
              "childStarts", "voxsPerChild", "voxPerVDB", "voxPerVol"],
    "formats": ["u1", "u1", "u1", ("u1", (3,)), ("<i4", (3,)), "<i4", "<i4", "<i4", ("<i4", (3,)),
                ("<u8", (3,)), ("<u8", (3,)), ("<f4", (3,)), "<f4", ("<f4", (3,))],
    "offsets": [0, 1, 2, 3, 8, 20, 24, 28, 32, 48, 72, 96, 108, 112], "itemsize": 128})
VDB_NODE_DTYPE = np.dtype([("pos", "<i4", (3,)), ("brickPosInAtlas", "<i4", (3,)),
                           ("childPoolOffsRelative", "<u8")])
assert VDB_NODE_DTYPE.itemsize == 32


class OracleVDB:
    """Handle of a VDB built by the CPU checker; arrays are copied out on construction."""

    def __init__(self, volume: np.ndarray, log2_dims=(5, 4, 3), apron_wid=1, use_depth_box=True,
                 empty_ranges=()):
        vol = np.ascontiguousarray(volume)
        assert vol.dtype in (np.uint8, np.uint16) and vol.ndim == 3
        self.is16 = int(vol.dtype == np.uint16)
        dz, dy, dx = vol.shape
        rng = np.ascontiguousarray(np.asarray(empty_ranges, vol.dtype).reshape(-1, 2))
        l2 = np.asarray(log2_dims, np.uint8)
        fn = lib().ko_vdb_build_u16 if self.is16 else lib().ko_vdb_build_u8
        fn.restype = C.c_void_p
        self.handle = fn(_vp(vol), C.c_int(dx), C.c_int(dy), C.c_int(dz), _vp(l2),
                         C.c_uint8(apron_wid), C.c_int(int(use_depth_box)), _vp(rng),
                         C.c_uint32(rng.shape[0]))
        if not self.handle:
            raise ValueError("the reference would reject this volume / configuration")
        sizes = np.zeros(5, np.uint64)
        lib().ko_vdb_sizes(C.c_void_p(self.handle), C.c_int(self.is16), _vp(sizes))
        self.param = np.zeros((), VDB_PARAM_DTYPE)
        self.nodes = np.zeros(int(sizes[0]), VDB_NODE_DTYPE)
        self.child_pools = np.zeros(int(sizes[1]), np.uint32)
        self.atlas_brick_to_node = np.zeros(int(sizes[2]), np.uint32)
        self.atlas = np.zeros(int(sizes[3]), vol.dtype)
        self.leaf_num = int(sizes[4])
        for what, arr in enumerate((self.param, self.nodes, self.child_pools,
                                    self.atlas_brick_to_node, self.atlas)):
            lib().ko_vdb_get(C.c_void_p(self.handle), C.c_int(self.is16), C.c_int(what), _vp(arr))
        A = int(self.param["voxPerAtlasBrick"])
        self.atlas = self.atlas.reshape(self.leaf_num, A, A, A)

    def self_test_rays(self, max_leaves: int = 256):
        """The reference's test_rayCastVDB (include/ray_caster/test_options.h:88-242) on this VDB: per
        ray (6 axis-parallel, 4 diagonal, through the centre of the VDB's cube) the leaves met, the
        count of wrong atlas reverse mappings, the leaves' node indices, and the rays (pos, dir)."""
        counts = np.zeros(10, np.int32)
        wrong = np.zeros(10, np.int32)
        idx = np.full((10, max_leaves), 0xFFFFFFFF, np.uint32)
        rays = np.zeros((10, 6), np.float32)
        lib().ko_vdb_self_test_rays(C.c_void_p(self.handle), C.c_int(self.is16), _vp(counts), _vp(wrong),
                                    _vp(idx), C.c_int(max_leaves), _vp(rays))
        return counts, wrong, idx, rays

    def __del__(self):
        if getattr(self, "handle", None):
            lib().ko_vdb_free(C.c_void_p(self.handle), C.c_int(self.is16))
            self.handle = None


class _KoDvrParams(C.Structure):
    _fields_ = [("w2s", C.c_float * 16), ("invProj", C.c_float * 16), ("eyePos", C.c_float * 3),
                ("eyeRot", C.c_float * 9), ("maxAlpha", C.c_float), ("samplingRate", C.c_float),
                ("useShading", C.c_int32), ("ka", C.c_float), ("kd", C.c_float), ("ks", C.c_float),
                ("shiness", C.c_float), ("lightPos", C.c_float * 3), ("lightCol", C.c_float * 3)]


def _dvr_params(w2s, inv_proj, eye_pos, eye_rot, max_alpha, sampling_rate, shading):
    pr = _KoDvrParams()
    for name, val in (("w2s", w2s), ("invProj", inv_proj), ("eyePos", eye_pos), ("eyeRot", eye_rot)):
        arr = getattr(pr, name)
        for i, x in enumerate(np.asarray(val, np.float32).reshape(-1)):
            arr[i] = float(x)
    pr.maxAlpha, pr.samplingRate = max_alpha, sampling_rate
    pr.useShading = 0
    if shading is not None:
        pr.useShading = 1
        pr.ka, pr.kd, pr.ks, pr.shiness = (shading[k] for k in ("ka", "kd", "ks", "shiness"))
        for i in range(3):
            pr.lightPos[i] = float(shading["lightPosInVolumeSpace"][i])
            pr.lightCol[i] = float(shading["lightCol"][i])
    return pr


def raw_render(volume: np.ndarray, tf: np.ndarray, w2s, inv_proj, eye_pos, eye_rot, W: int, H: int,
               max_alpha=0.9, sampling_rate=3.0, shading=None, rows=None, target: str = "Scene"):
    """One RenderRAWVolume frame on the CPU (target "Scene" or "AABB") -> (rgba [H,W,4] u8, samples)."""
    vol = np.ascontiguousarray(volume)
    assert vol.dtype in (np.uint8, np.uint16) and vol.ndim == 3
    dz, dy, dx = vol.shape
    pr = _dvr_params(w2s, inv_proj, eye_pos, eye_rot, max_alpha, sampling_rate, shading)
    tf = np.ascontiguousarray(tf, np.float32).reshape(256, 4)
    rgba = np.zeros((H, W, 4), np.uint8)
    y0, y1 = rows if rows is not None else (0, H)
    samples = C.c_uint64(0)
    lib().ko_raw_render(_vp(vol), C.c_int(int(vol.dtype == np.uint16)), C.c_int(dx), C.c_int(dy),
                        C.c_int(dz), _vp(tf), C.byref(pr), C.c_int(("Scene", "AABB").index(target)),
                        _vp(rgba), C.c_int(W), C.c_int(H), C.c_int(y0), C.c_int(y1), C.byref(samples))
    return rgba, int(samples.value)


def dvr_render_target(vdb: OracleVDB, w2s, inv_proj, eye_pos, eye_rot, W: int, H: int, target: str,
                      display_lev: int = 0, display_leaf_layer: int = 0, sampling_rate=3.0, rows=None):
    """One RenderDepthBoxVDB frame of an inspection target ("AABBs", "Depths",
    "PositionsAfterDepthSkip") on the CPU -> rgba [H,W,4] u8."""
    pr = _dvr_params(w2s, inv_proj, eye_pos, eye_rot, 0.9, sampling_rate, None)
    rgba = np.zeros((H, W, 4), np.uint8)
    y0, y1 = rows if rows is not None else (0, H)
    t = ("Scene", "AABBs", "Depths", "PositionsAfterDepthSkip").index(target)
    assert t > 0
    lib().ko_dvr_render_target(C.c_void_p(vdb.handle), C.c_int(vdb.is16), C.byref(pr), C.c_int(t),
                               C.c_int(display_lev), C.c_int(display_leaf_layer), _vp(rgba), C.c_int(W),
                               C.c_int(H), C.c_int(y0), C.c_int(y1))
    return rgba


def dvr_render(vdb: OracleVDB, tf: np.ndarray, w2s, inv_proj, eye_pos, eye_rot, W: int, H: int,
               max_alpha=0.9, sampling_rate=3.0, shading=None, rows=None, rgba=None):
    """One RenderDepthBoxVDB(Scene) frame on the CPU -> (rgba [H,W,4] u8, number of samples taken)."""
    pr = _KoDvrParams()
    for name, val in (("w2s", w2s), ("invProj", inv_proj), ("eyePos", eye_pos), ("eyeRot", eye_rot)):
        arr = getattr(pr, name)
        for i, x in enumerate(np.asarray(val, np.float32).reshape(-1)):
            arr[i] = float(x)
    pr.maxAlpha, pr.samplingRate = max_alpha, sampling_rate
    pr.useShading = 0
    if shading is not None:
        pr.useShading = 1
        pr.ka, pr.kd, pr.ks, pr.shiness = (shading[k] for k in ("ka", "kd", "ks", "shiness"))
        for i in range(3):
            pr.lightPos[i] = float(shading["lightPosInVolumeSpace"][i])
            pr.lightCol[i] = float(shading["lightCol"][i])
    tf = np.ascontiguousarray(tf, np.float32).reshape(256, 4)
    if rgba is None:
        rgba = np.zeros((H, W, 4), np.uint8)
    y0, y1 = rows if rows is not None else (0, H)
    samples = C.c_uint64(0)
    lib().ko_dvr_render(C.c_void_p(vdb.handle), C.c_int(vdb.is16), _vp(tf), C.byref(pr), _vp(rgba),
                        C.c_int(W), C.c_int(H), C.c_int(y0), C.c_int(y1), C.byref(samples))
    return rgba, int(samples.value)

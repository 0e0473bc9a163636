"""Closest-hit queries and (later) the path tracer: host-side mirror of kouek::RayTracer::RayTracer
(include/ray_tracer/ray_tracer.h:23-136)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .lbvh import LBVH


def closest_hit_status(workspace, ray_num: int) -> None:
    """Waits for the current stream and raises KoukeError when the last closest_hit(layout="trace") call
    with this workspace (the fourth entry of its `out`) found its ray sort had given up: the hits are
    right all the same (the kernel then traces in input order), the call ran at the unsorted rate."""
    with torch.cuda.device(workspace.device):
        _lib.check(_lib.lib().kk_closest_hit_accel_status(workspace.data_ptr(), int(ray_num),
                                                          _lib.current_stream_ptr()),
                   "kk_closest_hit_accel_status")


def closest_hit(lbvh: LBVH, rays, count: bool = False, out=None, layout: str = "trace", sort: bool = True,
                reuse_order: bool = False, watertight: bool = False, check: bool = False,
                entry_bound: bool = False):
    """Traces rays [R,6] (pos, dir) through a built LBVH with rayCastLBVH<false>'s result semantics
    (src/ray_tracer/ray_tracer_render.cu:192-270). Returns torch tensors (t [R], bary [R,2],
    prim [R] int32 with -1 = miss) and, when `count`, a dict of work counters.
    layout "trace" walks the repacked two-box records (rays in Morton order unless sort=False);
    layout "build" walks the build's own 32-byte nodes. Both give the same hits. reuse_order keeps the
    ray order found by the previous call with the same `out` workspace (same rays). watertight swaps the
    triangle test for the watertight one (trace layout only; hits may differ on grazing rays). check
    waits for the call and raises if its ray sort gave up (closest_hit_status; a loop that must not
    synchronise per call passes its own `out` and asks once at the end). entry_bound: groups of pending
    slots carry a bound of their entry distance and are dropped unvisited behind the best hit (same hits)."""
    l = _lib.lib()
    dev = lbvh.device
    d_rays = rays if isinstance(rays, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(rays, dtype=np.float32))
    d_rays = d_rays.to(device=dev, dtype=torch.float32).contiguous().reshape(-1, 6)
    R = d_rays.shape[0]
    assert layout in ("trace", "build")
    ws_bytes = (l.kk_closest_hit_accel_workspace_bytes(R) if layout == "trace"
                else l.kk_closest_hit_workspace_bytes())
    if out is None:
        t = torch.empty(R, dtype=torch.float32, device=dev)
        bary = torch.empty((R, 2), dtype=torch.float32, device=dev)
        prim = torch.empty(R, dtype=torch.int32, device=dev)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    else:
        t, bary, prim, ws = out
    counters = torch.zeros(16, dtype=torch.int64, device=dev) if count else None
    with torch.cuda.device(dev):
        if layout == "trace":
            accel = lbvh.trace_accel()
            N, M = lbvh._accel_sizes
            _lib.check(l.kk_closest_hit_accel(accel.data_ptr(), N, M, d_rays.data_ptr(), R, t.data_ptr(),
                                              bary.data_ptr(), prim.data_ptr(),
                                              (_lib.KK_TRACE_REUSE_ORDER if reuse_order else
                                               (0 if sort else _lib.KK_TRACE_UNSORTED)) |
                                              (_lib.KK_TRACE_WATERTIGHT if watertight else 0) |
                                              (_lib.KK_TRACE_ENTRY_BOUND if entry_bound else 0),
                                              counters.data_ptr() if count else None, ws.data_ptr(),
                                              ws.numel(), _lib.current_stream_ptr()),
                       "kk_closest_hit_accel")
        else:
            _lib.check(l.kk_closest_hit(lbvh.GetDeivceDataDevicePtr(), lbvh.d_positions.data_ptr(),
                                        lbvh.d_face_pos_idx.data_ptr(), d_rays.data_ptr(), R,
                                        t.data_ptr(), bary.data_ptr(), prim.data_ptr(),
                                        _lib.KK_TRAVERSE_ORDERED,
                                        counters.data_ptr() if count else None, ws.data_ptr(), ws.numel(),
                                        _lib.current_stream_ptr()), "kk_closest_hit")
    if check and layout == "trace":
        closest_hit_status(ws, R)
    if count:
        c = counters.cpu().numpy()
        return t, bary, prim, {"node_fetches": int(c[0]), "tri_tests": int(c[1]), "rays": int(c[2]),
                               "wave_passes": int(c[3]), "visit_lanes": int(c[4]), "tri_phases": int(c[5]),
                               "tri_lanes": int(c[6]), "empty_visits": int(c[7]),
                               "empty_visits_by_limit": int(c[8]), "used_slots_visited": int(c[9]),
                               "leaf_only_visits": int(c[10]), "slots_dropped_unvisited": int(c[11]),
                               # wall clock (100 MHz ticks -> us): the launch from its first wave's start to
                               # its last wave's exit, how long before that exit the first wave found every
                               # queue used up, and the mean life of a wave
                               "launch_us": (int(c[14]) - ((1 << 62) - int(c[12]))) / 100.0,
                               "tail_us": (int(c[14]) - ((1 << 62) - int(c[13]))) / 100.0 if c[13] else 0.0,
                               "wave_life_sum_us": int(c[15]) / 100.0}
    return t, bary, prim


# ---- materials / lights as the reference lays them out ------------------------------------------------

TRIANGLE_DTYPE = np.dtype([("normIdx", "<u4", (3,)), ("texCoordIdx", "<u4", (3,)), ("grpIdx", "<u4"),
                           ("_pad", "<u4")])
MATERIAL_DTYPE = np.dtype([("emissive", "u1"), ("_pad", "u1", (3,)), ("emission", "<f4", (3,)),
                           ("kd", "<f4", (3,)), ("ks", "<f4", (3,)), ("tr", "<f4", (3,)),
                           ("ni", "<f4"), ("ns", "<f4"), ("kdOverKdAddKs", "<f4")])
# union at offset 4: quad {o,u,v,n,invArea} / sphere {o,r,rSqr}
LIGHT_DTYPE = np.dtype({"names": ["type", "quad_o", "quad_u", "quad_v", "quad_n", "quad_invArea",
                                  "sphere_o", "sphere_r", "sphere_rSqr", "area", "radiance"],
                        "formats": ["u1", ("<f4", (3,)), ("<f4", (3,)), ("<f4", (3,)), ("<f4", (3,)),
                                    "<f4", ("<f4", (3,)), "<f4", "<f4", "<f4", ("<f4", (3,))],
                        "offsets": [0, 4, 16, 28, 40, 52, 4, 16, 20, 56, 60], "itemsize": 80})
assert TRIANGLE_DTYPE.itemsize == 32 and MATERIAL_DTYPE.itemsize == 64 and LIGHT_DTYPE.itemsize == 80

f32 = np.float32


def create_material(kd, ks, tr, ni, ns, emission=(0, 0, 0)) -> np.ndarray:
    """Material::Create (include/ray_tracer/material.h:23-39)."""
    m = np.zeros((), MATERIAL_DTYPE)
    kd, ks, emission = np.asarray(kd, f32), np.asarray(ks, f32), np.asarray(emission, f32)
    m["emissive"] = bool((emission != 0).any())
    m["emission"], m["kd"], m["ks"], m["tr"] = emission, kd, ks, np.asarray(tr, f32)
    m["ns"], m["ni"] = f32(ns), f32(ni)
    kd_avg = f32(f32(f32(kd[0] + kd[1]) + kd[2]) / f32(3))
    ks_avg = f32(f32(f32(ks[0] + ks[1]) + ks[2]) / f32(3))
    with np.errstate(invalid="ignore", divide="ignore"):
        m["kdOverKdAddKs"] = f32(kd_avg / f32(kd_avg + ks_avg))
    return m


def create_sphere_light(r, o, radiance) -> np.ndarray:
    """Light::CreateSphere (material.h:82-93): area = 2 * two_pi * r^2."""
    l = np.zeros((), LIGHT_DTYPE)
    r = f32(r)
    l["type"] = 1
    l["sphere_o"], l["sphere_r"], l["sphere_rSqr"] = np.asarray(o, f32), r, f32(r * r)
    l["area"] = f32(f32(f32(2) * f32(6.28318530717958647692528676655900576)) * f32(r * r))
    l["radiance"] = np.asarray(radiance, f32)
    return l


def create_quad_light(o, u, v, radiance) -> np.ndarray:
    """Light::CreateQuad (material.h:68-81)."""
    from .data.camera import _cross, _normalize, _dot, _c
    l = np.zeros((), LIGHT_DTYPE)
    o, u, v = np.asarray(o, f32), np.asarray(u, f32), np.asarray(v, f32)
    l["type"] = 0
    l["quad_o"], l["quad_u"], l["quad_v"] = o, u, v
    l["quad_n"] = _normalize(_cross(u, v))
    area = f32(_c("sqrtf", _dot(u, u)) * _c("sqrtf", _dot(v, v)))
    l["area"] = area
    l["quad_invArea"] = f32(f32(1) / area)
    l["radiance"] = np.asarray(radiance, f32)
    return l


def stack_structs(items, dtype) -> np.ndarray:
    """Array of structs from a list of 0-d struct values, keeping `dtype`'s explicit offsets
    (np.array() of such a list would repack the fields)."""
    out = np.zeros(len(items), dtype)
    for i, it in enumerate(items):
        out[i] = it
    return out


class _PtScene(C.Structure):
    _fields_ = [("display_tree_height", C.c_uint8), ("max_path_depth", C.c_uint8),
                ("_pad", C.c_uint8 * 2), ("light_num", C.c_uint32), ("w2s", C.c_float * 16),
                ("inv_proj", C.c_float * 16), ("triangles", C.c_void_p),
                ("tri_to_position_indices", C.c_void_p), ("positions", C.c_void_p),
                ("normals", C.c_void_p), ("tex_coords", C.c_void_p), ("lights", C.c_void_p),
                ("materials", C.c_void_p), ("light_tot_area", C.c_float), ("_pad2", C.c_uint32),
                ("trace_accel", C.c_void_p), ("accel_face_num", C.c_uint32),
                ("accel_compacted_face_num", C.c_uint32)]


class _PtFrame(C.Structure):
    _fields_ = [("eye_pos", C.c_float * 3), ("eye_rot", C.c_float * 9)]


class FrameBuffer:
    """Headless stand-in for the GL surface the reference renders into: RGBA8, row 0 = bottom row."""

    def __init__(self, width: int, height: int, device="cuda:0"):
        self.width, self.height = int(width), int(height)
        self.rgba = torch.zeros((self.height, self.width, 4), dtype=torch.uint8, device=device)

    def ReadBack(self) -> np.ndarray:
        return self.rgba.cpu().numpy()


class RayTracer:
    """kouek::RayTracer::RayTracer (include/ray_tracer/ray_tracer.h:23-136): setters with the
    reference's restart-on-change behaviour, Render(framebuffer) adds one sample per call."""
    ErrTag = "[RayTracer Error]"
    SEED = 1234  # curand_init(1234, pixel, 0), ray_tracer_render.cu:572

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        self.lbvh = None
        self.max_sample_per_pixel = 1024.0
        self._prev_max_spp = 1024.0
        self.curr_accu_spp = 0.0
        self._prev_size = (0, 0)
        self._prev_fb = None
        self._modified = True
        self._scene = _PtScene()
        self._scene.max_path_depth = 4
        self._frame = _PtFrame()
        self.d_accum = None
        self._keep = {}

    # -- setters (ray_tracer.cu:15-107); any change restarts accumulation at the next Render ----------
    def _set(self, obj, name, value):
        cur = getattr(obj, name)
        if hasattr(cur, "__len__"):
            new = [float(x) for x in np.asarray(value, np.float32).reshape(-1)]
            if list(cur) != new:
                for i, x in enumerate(new):
                    cur[i] = x
                self._modified = True
        elif cur != value:
            setattr(obj, name, value)
            self._modified = True

    def SetMesh(self, positions, normals, tex_coords, group_start_face_indices, face_position_indices,
                face_normal_indices, face_tex_coord_indices, lights, materials):
        dev = self.device
        fpi = np.ascontiguousarray(face_position_indices, np.uint32).reshape(-1, 3)
        n = fpi.shape[0]
        tris = np.zeros(n, TRIANGLE_DTYPE)
        tris["normIdx"] = np.asarray(face_normal_indices, np.uint32).reshape(-1, 3)
        tris["texCoordIdx"] = np.asarray(face_tex_coord_indices, np.uint32).reshape(-1, 3)
        # group of a face: last group whose start is <= the face (ray_tracer.cu:43-57)
        gs = np.asarray(group_start_face_indices, np.uint32)
        tris["grpIdx"] = (np.searchsorted(gs[1:], np.arange(n, dtype=np.uint32), side="right")
                          if gs.size > 1 else 0)
        lights = lights if isinstance(lights, np.ndarray) else stack_structs(lights, LIGHT_DTYPE)
        materials = (materials if isinstance(materials, np.ndarray)
                     else stack_structs(materials, MATERIAL_DTYPE))
        assert lights.dtype.itemsize == 80 and materials.dtype.itemsize == 64, "struct layout lost"
        # the kernels follow these indices into the arrays below: one past an array is a GPU memory fault
        if n:
            for what, idx, have in (("position", fpi, np.asarray(positions).reshape(-1, 3).shape[0]),
                                    ("normal", tris["normIdx"], np.asarray(normals).reshape(-1, 3).shape[0]),
                                    ("material", tris["grpIdx"], int(materials.size))):
                if int(idx.max()) >= have:
                    raise ValueError(f"a face refers to {what} {int(idx.max())}, the mesh has {have}")
            n_tex = np.asarray(tex_coords).reshape(-1, 2).shape[0] if len(tex_coords) else 0
            if n_tex and int(tris["texCoordIdx"].max()) >= n_tex:
                raise ValueError(f"a face refers to texture coordinate {int(tris['texCoordIdx'].max())}, "
                                 f"the mesh has {n_tex}")

        sizes = _lib.PtSceneSizes()
        sizes.face_num, sizes.vert_num = n, int(np.asarray(positions).reshape(-1, 3).shape[0])
        sizes.normal_num = int(np.asarray(normals).reshape(-1, 3).shape[0])
        sizes.tex_coord_num = int(np.asarray(tex_coords).reshape(-1, 2).shape[0]) if len(tex_coords) else 0
        sizes.light_num, sizes.material_num = int(lights.size), int(materials.size)
        tot = np.float32(0)
        for a in lights["area"]:
            tot = np.float32(tot + a)
        sizes.light_tot_area = float(tot)
        self.alloc_scene_blob(sizes)
        k = self._keep

        def up(name, a):
            a = np.ascontiguousarray(a)
            if a.size:
                k[name].copy_(torch.from_numpy(a.view(np.uint8).reshape(-1).copy()))

        up("triangles", tris), up("tri", fpi)
        up("positions", np.asarray(positions, np.float32))
        up("normals", np.asarray(normals, np.float32))
        up("tex_coords", np.asarray(tex_coords, np.float32))
        up("lights", lights), up("materials", materials)
        self.host_arrays = {"triangles": tris, "tri": fpi, "positions": np.asarray(positions, np.float32),
                            "normals": np.asarray(normals, np.float32), "lights": lights,
                            "materials": materials}
        self._modified = True

    def alloc_scene_blob(self, sizes: "_lib.PtSceneSizes") -> None:
        """The arrays of SetMesh live in one allocation (so that the scene travels to other GPUs in one
        collective, kk_pt_scene_blob_layout_of); the scene struct's pointers point into it."""
        lay = _lib.PtSceneBlobLayout()
        _lib.check(_lib.lib().kk_pt_scene_blob_layout_of(C.byref(sizes), C.byref(lay)),
                   "kk_pt_scene_blob_layout_of")
        b = torch.zeros(int(lay.total), dtype=torch.uint8, device=self.device)
        self.d_scene_blob, self.scene_sizes, self.scene_blob_layout = b, sizes, lay
        piece = lambda off, nbytes: b[int(off):int(off) + nbytes]
        self._keep = {"triangles": piece(lay.triangles, 32 * sizes.face_num),
                      "tri": piece(lay.tri_to_position_indices, 12 * sizes.face_num),
                      "positions": piece(lay.positions, 12 * sizes.vert_num),
                      "normals": piece(lay.normals, 12 * sizes.normal_num),
                      "tex_coords": piece(lay.tex_coords, 8 * sizes.tex_coord_num),
                      "lights": piece(lay.lights, 80 * sizes.light_num),
                      "materials": piece(lay.materials, 64 * sizes.material_num)}
        _lib.check(_lib.lib().kk_pt_scene_blob_bind(b.data_ptr(), C.byref(sizes), C.byref(self._scene)),
                   "kk_pt_scene_blob_bind")
        self._modified = True

    def scene_host_arrays(self) -> dict:
        """The scene's arrays read back from its allocation, in the reference's struct layouts (what
        SetMesh uploaded, or what a broadcast delivered)."""
        # (np.frombuffer over the bytes: copying a struct array with explicit offsets field by field would
        # not keep its padding bytes)
        k = {n: t.cpu().numpy().tobytes() for n, t in self._keep.items()}
        return {"triangles": np.frombuffer(k["triangles"], TRIANGLE_DTYPE),
                "tri": np.frombuffer(k["tri"], np.uint32).reshape(-1, 3),
                "positions": np.frombuffer(k["positions"], np.float32).reshape(-1, 3),
                "normals": np.frombuffer(k["normals"], np.float32).reshape(-1, 3),
                "tex_coords": np.frombuffer(k["tex_coords"], np.float32).reshape(-1, 2),
                "lights": np.frombuffer(k["lights"], LIGHT_DTYPE),
                "materials": np.frombuffer(k["materials"], MATERIAL_DTYPE)}

    use_trace_tree = True  # walk the 8-wide trace-side tree (same image); False: the binary tree as built

    def SetLBVH(self, lbvh: LBVH):
        self.lbvh = lbvh

    def RenderGathered(self, fb: FrameBuffer, gatherer, samples: int = 1) -> None:
        """Several GPUs (not part of the reference): one Render(Scene) call of a frame shared between the
        ranks of `gatherer` (multi_gpu.StripeGatherer). Every rank adds the sample(s) to its round-robin
        16-row stripes, straight into the block it sends; the gather is issued without waiting for it and
        the frame of two calls ago is put together in fb on the gathering rank (gatherer.flush(fb.rgba)
        brings the last ones in). The random stream is keyed by the image pixel, so the gathered frame is
        the frame one GPU renders, byte for byte."""
        buf = gatherer.begin(fb.rgba)
        self.Render(fb, samples=samples, stripes=(gatherer.rank, gatherer.world), compact_out=buf)
        gatherer.submit()

    def SetWorldToScene(self, w2s):
        self._set(self._scene, "w2s", w2s)

    def SetCameraToWorld(self, eye_pos, eye_rot):
        self._set(self._frame, "eye_pos", eye_pos)
        self._set(self._frame, "eye_rot", eye_rot)

    def SetProjection(self, proj):
        from .data.camera import inverse_projective
        self._set(self._scene, "inv_proj", inverse_projective(proj))

    def SetDisplayTreeHeight(self, h: int):
        self._set(self._scene, "display_tree_height", int(h))

    def SetMaxPathDepth(self, d: int):
        self._set(self._scene, "max_path_depth", int(d))

    def SetMaxSamplePerPixel(self, v: float):
        self.max_sample_per_pixel = float(v)

    def GetMaxSamplePerPixel(self) -> float:
        return self.max_sample_per_pixel

    def GetCurrentAccumulatedSamplePerPixel(self) -> float:
        return self.curr_accu_spp

    RENDER_TARGETS = ("Scene", "AABBs", "Triangles", "Lights", "Positions", "Normals", "TextureCoords",
                      "None")

    def Render(self, fb: FrameBuffer, rows=None, target: str = "Scene", samples: int = 1, stripes=None,
               compact_out=None) -> None:
        """ray_tracer_render.cu:433-647. `target` is one of RENDER_TARGETS (RayTracer::RenderTarget);
        "Scene" path-traces and accumulates one sample, the others draw one inspection frame. `rows` =
        (y0, y1) restricts the call to a stripe of the image (multi-GPU tiling); the sample counter
        advances all the same. `samples` > 1 does the work of that many consecutive calls in one launch
        (same image, bit for bit; never past maxSamplePerPixel). `stripes` = (first, stride) with
        `compact_out` (uint8 tensor [per, 16, W, 4]) renders the 16-row stripes first, first + stride, ...
        (one GPU's share of a frame dealt out to several) one after the other into compact_out - the block
        a rank hands to the gathering rank - and accumulates them in a compact buffer of the same rows;
        fb then only names the frame's size."""
        import sys
        if self.lbvh is None or not self.lbvh.IsComplete():
            print(f"{self.ErrTag} lbvh is not set.", file=sys.stderr)
            return
        if target not in self.RENDER_TARGETS:
            raise ValueError(f"unknown render target {target!r}")
        if target == "None":
            return
        if target != "Scene":
            y0, y1 = rows if rows is not None else (0, fb.height)
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().kk_rt_render_target(
                    self.lbvh.GetDeivceDataDevicePtr(), C.byref(self._scene), C.byref(self._frame),
                    self.RENDER_TARGETS.index(target), fb.rgba.data_ptr(), fb.width, fb.height, int(y0),
                    int(y1), _lib.current_stream_ptr()), "kk_rt_render_target")
            # like the reference, a switch of target restarts the accumulation of the next Scene call
            self._modified = True
            return
        size = (fb.width, fb.height) + (tuple(int(v) for v in stripes) if stripes is not None else ())
        need_redraw = self._modified
        self._modified = False
        if self._prev_size != size or self.d_accum is None:
            acc_rows = fb.height
            if stripes is not None:
                acc_rows = 16 * int(_lib.lib().kk_stripes_per_rank(fb.height, int(stripes[1])))
            self.d_accum = torch.zeros((acc_rows, fb.width, 3), dtype=torch.float32, device=self.device)
        need_redraw |= (self._prev_max_spp != self.max_sample_per_pixel or self._prev_fb is not fb or
                        self._prev_size != size)
        if need_redraw:
            self.curr_accu_spp = 1.0
            self.d_accum.zero_()
        self._prev_fb, self._prev_size = fb, size
        if self.curr_accu_spp > self.max_sample_per_pixel:
            return
        y0, y1 = rows if rows is not None else (0, fb.height)
        with torch.cuda.device(self.device):
            if self.use_trace_tree:
                self._scene.trace_accel = self.lbvh.trace_accel().data_ptr()
                self._scene.accel_face_num, self._scene.accel_compacted_face_num = self.lbvh._accel_sizes
            else:
                self._scene.trace_accel = None
            n = int(min(max(int(samples), 1), self.max_sample_per_pixel - self.curr_accu_spp + 1))
            if stripes is not None:
                if compact_out is None:
                    raise ValueError("stripes need compact_out")
                need = self.d_accum.shape[0] * fb.width * 4
                if compact_out.numel() * compact_out.element_size() < need:
                    raise _lib.KoukeError("compact_out is smaller than this rank's stripes")
                _lib.check(_lib.lib().kk_pt_render_stripes_compact(
                    self.lbvh.GetDeivceDataDevicePtr(), C.byref(self._scene), C.byref(self._frame),
                    self.d_accum.data_ptr(), compact_out.data_ptr(), fb.width, fb.height, int(stripes[0]),
                    int(stripes[1]), C.c_float(self.curr_accu_spp), n, self.SEED, _lib.current_stream_ptr()),
                    "kk_pt_render_stripes_compact")
            else:
                _lib.check(_lib.lib().kk_pt_render_samples(
                    self.lbvh.GetDeivceDataDevicePtr(), C.byref(self._scene), C.byref(self._frame),
                    self.d_accum.data_ptr(), fb.rgba.data_ptr(), fb.width, fb.height, int(y0), int(y1),
                    C.c_float(self.curr_accu_spp), n, self.SEED, _lib.current_stream_ptr()),
                    "kk_pt_render_samples")
        self._prev_max_spp = self.max_sample_per_pixel
        self.curr_accu_spp += float(n)

"""OBJ + MTL loader with the reference's semantics (include/data/mesh.h:89-234, 322-461).

Strict: only the tags g, f, v, vn, vt, mtllib, usemtl (OBJ) and Kd, Ks, Tr, Ni, Ns, newmtl (MTL) are
accepted; any other tag makes the mesh incomplete, as in the reference (mesh.h:119-122, 410-413).
Faces must be `f a/b/c d/e/f g/h/i` (mesh.h:129-143). Numbers are parsed with the C library's strtof
so that they carry the same float32 bits as the reference's sscanf("%f").
"""
from __future__ import annotations

import ctypes
import ctypes.util
import os
import re
from dataclasses import dataclass, field

import numpy as np

_libc = ctypes.CDLL(ctypes.util.find_library("c") or "libc.so.6")
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p)]


def _strtof(tok: str) -> np.float32:
    return np.float32(_libc.strtof(tok.encode(), None))


_FLOAT_RE = re.compile(r"\s*([+-]?(?:\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|inf|nan))", re.I)


def _scan_floats(s: str, n: int):
    """sscanf("%f" * n): returns the floats parsed before the first failure."""
    out, pos = [], 0
    for _ in range(n):
        m = _FLOAT_RE.match(s, pos)
        if not m:
            break
        out.append(_strtof(m.group(1)))
        pos = m.end()
    return out


_OBJ_TAGS = ("g", "f", "v", "vn", "vt", "mtllib", "usemtl")
_MTL_TAGS = ("Kd", "Ks", "Tr", "Ni", "Ns", "newmtl")
_FACE_RE = re.compile(r"\s*(\d+)/(\d+)/(\d+)\s+(\d+)/(\d+)/(\d+)\s+(\d+)/(\d+)/(\d+)")

QUAD, SPHERE = 0, 1


@dataclass
class ObjLight:
    type: int
    radiance: np.ndarray
    o: np.ndarray = None
    u: np.ndarray = None
    v: np.ndarray = None
    r: np.float32 = np.float32(0)


@dataclass
class ObjMaterial:
    kd: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    ks: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    tr: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    ni: np.float32 = np.float32(0)
    ns: np.float32 = np.float32(0)


class OBJMesh:
    """`input_lights` maps a material name to (type, radiance) as the scene XML gives it
    (app/rt_lbvh.cpp:116-142)."""

    def __init__(self, path: str | None = None, input_lights: dict | None = None):
        self.is_complete = False
        self.positions: list = []
        self.normals: list = []
        self.tex_coords: list = []
        self.grp_start_face_indices: list = []
        self.face_position_indices: list = []
        self.face_normal_indices: list = []
        self.face_tex_coord_indices: list = []
        self.grp2mtl_names: dict = {}
        self.name2mtls: dict = {}
        self.lights: list = []
        self.error = ""
        if path is not None:
            self._load(path, input_lights)
            self._finalize()

    # -- accessors in array form ---------------------------------------------------------------
    def _finalize(self):
        self.positions = np.asarray(self.positions, np.float32).reshape(-1, 3)
        self.normals = np.asarray(self.normals, np.float32).reshape(-1, 3)
        self.tex_coords = np.asarray(self.tex_coords, np.float32).reshape(-1, 2)
        self.grp_start_face_indices = np.asarray(self.grp_start_face_indices, np.uint32)
        self.face_position_indices = np.asarray(self.face_position_indices, np.uint32).reshape(-1, 3)
        self.face_normal_indices = np.asarray(self.face_normal_indices, np.uint32).reshape(-1, 3)
        self.face_tex_coord_indices = np.asarray(self.face_tex_coord_indices, np.uint32).reshape(-1, 3)

    @staticmethod
    def _parse_tag(line: str, tags) -> tuple:
        offs = line.find(" ")
        if offs < 0:
            offs = len(line)
        head = line[:offs]
        for i, t in enumerate(tags):
            if head == t:
                return i, offs
        return len(tags), offs

    def _fail(self, msg: str):
        self.error = msg
        return False

    def _load(self, path: str, input_lights):
        try:
            fh = open(path, "r", errors="replace")
        except OSError:
            return self._fail(f'Cannot open file at "{path}".')
        prev_is_light, prev_light = False, None
        with fh:
            for ln, raw in enumerate(fh, 1):
                line = raw.rstrip("\n").rstrip("\r")
                if not line:
                    continue
                tag, offs = self._parse_tag(line, _OBJ_TAGS)
                if tag == len(_OBJ_TAGS):
                    return self._fail(f"Failed parsing line {ln}.")
                rest = line[offs + 1:]
                name = _OBJ_TAGS[tag]
                if name == "g":
                    self.grp_start_face_indices.append(len(self.face_position_indices))
                elif name == "f":
                    m = _FACE_RE.match(rest)
                    if not m:
                        return self._fail(f"Failed parsing line {ln}.")
                    v = [int(x) - 1 for x in m.groups()]
                    # indices count from one and fit 32 bits (the reference lets a zero wrap to 2^32 - 1)
                    if min(v) < 0 or max(v) >= 0xFFFFFFFF:
                        return self._fail(f"Failed parsing line {ln}.")
                    self.face_position_indices.append((v[0], v[3], v[6]))
                    self.face_tex_coord_indices.append((v[1], v[4], v[7]))
                    self.face_normal_indices.append((v[2], v[5], v[8]))
                elif name in ("v", "vn"):
                    f = _scan_floats(rest, 3)
                    if len(f) != 3:
                        return self._fail(f"Failed parsing line {ln}.")
                    (self.positions if name == "v" else self.normals).append(f)
                elif name == "vt":
                    f = _scan_floats(rest, 2)
                    if len(f) != 2:
                        return self._fail(f"Failed parsing line {ln}.")
                    self.tex_coords.append(f)
                elif name == "mtllib":
                    if not self._load_mtl(os.path.join(os.path.dirname(path), rest)):
                        return False
                elif name == "usemtl":
                    if not self.grp_start_face_indices:  # a material belongs to a group
                        return self._fail(f"Failed parsing line {ln}.")
                    if prev_is_light and not self._append_light(prev_light, len(self.grp_start_face_indices) - 2):
                        return False
                    gi = len(self.grp_start_face_indices) - 1
                    if gi in self.grp2mtl_names:
                        return self._fail(f"Failed parsing line {ln}.")
                    self.grp2mtl_names[gi] = rest
                    prev_is_light = bool(input_lights) and rest in input_lights
                    if prev_is_light:
                        prev_light = input_lights[rest]
        if prev_is_light and not self._append_light(prev_light, len(self.grp_start_face_indices) - 1):
            return False
        # Every index must name an element that exists: the arrays go to the GPU as they are, and a kernel
        # that follows an index past its array faults. (The reference does not check.)
        for idx, have in ((self.face_position_indices, len(self.positions)),
                          (self.face_normal_indices, len(self.normals)),
                          (self.face_tex_coord_indices, len(self.tex_coords))):
            if idx and int(np.asarray(idx, np.int64).max()) >= have:
                fi = int(np.argmax(np.asarray(idx, np.int64).max(axis=1) >= have))
                return self._fail(f"Face {fi} refers to a position, normal or texture coordinate that the "
                                  "file does not define.")
        self.is_complete = True
        return True

    def _append_light(self, input_light, gi: int) -> bool:
        """mesh.h:322-381. Note the last group ends at faceNormalIndices.size() - 1 (one face short)."""
        ltype, radiance = input_light
        pos = np.asarray(self.positions, np.float32).reshape(-1, 3)
        fpi = self.face_position_indices
        gs = self.grp_start_face_indices
        if gi < 0 or gi >= len(gs) or not self.face_normal_indices:
            return self._fail("A light's material is used before a group with faces exists.")
        fi_end = len(self.face_normal_indices) - 1 if gi == len(gs) - 1 else gs[gi + 1]
        # the group's faces are read here, while the file is still being parsed: they and their vertices
        # must have come before
        if gs[gi] > fi_end or fi_end > len(fpi) or (ltype != QUAD and fi_end == gs[gi]):
            return self._fail(f"Group {gi} of a light holds no faces.")
        for fi in range(gs[gi], fi_end):
            if max(fpi[fi]) >= pos.shape[0]:
                return self._fail(f"Face {fi} of a light refers to a position that is not defined.")
        lht = ObjLight(type=ltype, radiance=np.asarray(radiance, np.float32))
        if ltype == QUAD:
            if fi_end - gs[gi] != 2:
                return self._fail(f"Triangle number of group {gi} is not 2.")
            idx = fpi[gs[gi]]
            vecs = [pos[idx[1]] - pos[idx[0]], pos[idx[2]] - pos[idx[1]], pos[idx[0]] - pos[idx[2]]]

            def dot(a, b):
                t = a * b
                return np.float32(np.float32(t[0] + t[1]) + t[2])

            dots = [dot(vecs[0], vecs[1]), dot(vecs[1], vecs[2]), dot(vecs[2], vecs[0])]
            for i in range(3):
                if abs(dots[i]) < np.finfo(np.float32).eps:
                    lht.o = pos[idx[(i + 1) % 3]].copy()
                    lht.u = pos[idx[(i + 2) % 3]] - lht.o
                    lht.v = pos[idx[i]] - lht.o
                    break
            else:
                return self._fail(f"Triangle at {gs[gi]} is not right.")
        else:
            fi_beg = gs[gi]
            div = np.float32(3.0) * np.float32(fi_end - fi_beg)
            o = np.zeros(3, np.float32)
            for fi in range(fi_beg, fi_end):
                for i in range(3):
                    o = (o + pos[fpi[fi][i]] / div).astype(np.float32)
            r = np.float32(0)
            for fi in range(fi_beg, fi_end):
                for i in range(3):
                    d = (pos[fpi[fi][i]] - o).astype(np.float32)
                    t = d * d
                    dist = np.sqrt(np.float32(np.float32(t[0] + t[1]) + t[2]), dtype=np.float32)
                    r = np.float32(r + np.float32(dist / div))
            lht.o, lht.r = o, r
        self.lights.append(lht)
        return True

    def _load_mtl(self, path: str) -> bool:
        try:
            fh = open(path, "r", errors="replace")
        except OSError:
            return self._fail(f'Cannot open file at "{path}".')
        self.name2mtls = {}
        cur = None
        with fh:
            for ln, raw in enumerate(fh, 1):
                line = raw.rstrip("\n").rstrip("\r")
                if not line:
                    continue
                tag, offs = self._parse_tag(line, _MTL_TAGS)
                if tag == len(_MTL_TAGS):
                    return self._fail(f"Failed parsing line {ln}.")
                rest = line[offs + 1:]
                name = _MTL_TAGS[tag]
                if name == "newmtl":
                    if rest in self.name2mtls:
                        return self._fail(f"Failed parsing line {ln}.")
                    cur = self.name2mtls[rest] = ObjMaterial()
                elif cur is None:  # a property before the first newmtl
                    return self._fail(f"Failed parsing line {ln}.")
                elif name in ("Ni", "Ns"):
                    f = _scan_floats(rest, 1)
                    if len(f) != 1:
                        return self._fail(f"Failed parsing line {ln}.")
                    setattr(cur, name.lower(), f[0])
                else:
                    f = _scan_floats(rest, 3)
                    if len(f) != 3:
                        return self._fail(f"Failed parsing line {ln}.")
                    setattr(cur, name.lower(), np.asarray(f, np.float32))
        return True


def parse_scene_xml(path: str):
    """The scene description read by app/rt_lbvh.cpp:60-146: camera (size, eye, lookat, up) and the
    `<light mtlname= type= radiance=>` elements. Returns None if any required piece is missing."""
    import xml.etree.ElementTree as ET

    try:
        text = open(path, "r", errors="replace").read()
        # the files hold several top-level elements; wrap them so a strict parser accepts them
        body = re.sub(r"<\?xml[^>]*\?>", "", text)
        root = ET.fromstring("<scene>" + body + "</scene>")
    except (OSError, ET.ParseError, ValueError):  # unreadable or not XML: no scene description
        return None
    cam = root.find("camera")
    if cam is None or cam.get("type") != "perspective":
        return None
    try:
        size = (int(cam.get("width")), int(cam.get("height")))

        def v3(tag):
            e = cam.find(tag)
            return np.array([_strtof(e.get(k)) for k in ("x", "y", "z")], np.float32)

        eye, lookat, up = v3("eye"), v3("lookat"), v3("up")
    except (TypeError, AttributeError, ValueError):
        return None
    lights = {}
    for l in root.findall("light"):
        t = l.get("type")
        if t not in ("Quad", "Sphere"):
            return None
        rad = l.get("radiance")
        name = l.get("mtlname")
        if rad is None or name is None:
            return None
        parts = rad.split(",")
        if len(parts) != 3:
            return None
        lights[name] = (QUAD if t == "Quad" else SPHERE, np.array([_strtof(p) for p in parts], np.float32))
    return {"size": size, "eye": eye, "lookat": lookat, "up": up, "lights": lights}

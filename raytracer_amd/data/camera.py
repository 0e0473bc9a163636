"""Camera and projection set-up of the two demo applications, in float32 with the C library's
single-precision functions so the matrices carry the same bits as the reference's glm code:
  FPSCamera            include/cg/FPS_camera.h:25-31, 70-141
  perspective_fov      glm::perspectiveFov (3rd/glm/ext/matrix_clip_space.inl:472-482), RH, [-1,1] depth
  inverse_projective   kouek::Math::InverseProjective (include/cg/math.h:14-52)
  world_to_scene       app/dvr_dpbx_vdb.cpp:272-273
All matrices are column-major 4x4 / 3x3 numpy arrays indexed m[col][row] (glm's convention).
"""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("cosf", "sinf", "acosf", "asinf", "sqrtf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]

f32 = np.float32


def _c(name, x):
    return f32(getattr(_libm, name)(ctypes.c_float(float(x))))


def radians(deg) -> np.float32:
    return f32(f32(deg) * f32(0.01745329251994329576923690768489))


def _dot(a, b):
    t = (a * b).astype(f32)
    return f32(f32(t[0] + t[1]) + t[2])


def _cross(x, y):
    return np.array([f32(x[1] * y[2]) - f32(y[1] * x[2]),
                     f32(x[2] * y[0]) - f32(y[2] * x[0]),
                     f32(x[0] * y[1]) - f32(y[0] * x[1])], f32)


def _normalize(v):
    v = np.asarray(v, f32)
    return (v * f32(f32(1) / _c("sqrtf", _dot(v, v)))).astype(f32)


def perspective_fov(fov_rad, width, height, z_near, z_far) -> np.ndarray:
    fov_rad, width, height = f32(fov_rad), f32(width), f32(height)
    z_near, z_far = f32(z_near), f32(z_far)
    h = f32(_c("cosf", f32(0.5) * fov_rad) / _c("sinf", f32(0.5) * fov_rad))
    w = f32(f32(h * height) / width)
    m = np.zeros((4, 4), f32)
    m[0][0] = w
    m[1][1] = h
    m[2][2] = f32(-f32(z_far + z_near) / f32(z_far - z_near))
    m[2][3] = f32(-1)
    m[3][2] = f32(-f32(f32(2) * z_far * z_near) / f32(z_far - z_near))
    return m


def _mat2_inverse(m):
    inv_det = f32(f32(1) / f32(f32(m[0][0] * m[1][1]) - f32(m[1][0] * m[0][1])))
    return np.array([[f32(m[1][1] * inv_det), f32(-m[0][1] * inv_det)],
                     [f32(-m[1][0] * inv_det), f32(m[0][0] * inv_det)]], f32)


def _mat2_mul(a, b):
    r = np.zeros((2, 2), f32)
    for c in range(2):
        for row in range(2):
            r[c][row] = f32(f32(a[0][row] * b[c][0]) + f32(a[1][row] * b[c][1]))
    return r


def inverse_projective(p) -> np.ndarray:
    p = np.asarray(p, f32)
    a = np.array([[p[0][0], p[0][1]], [p[1][0], p[1][1]]], f32)
    b = np.array([[p[2][0], p[2][1]], [p[3][0], p[3][1]]], f32)
    c = np.array([[p[0][2], p[0][3]], [p[1][2], p[1][3]]], f32)
    d = np.array([[p[2][2], p[2][3]], [p[3][2], p[3][3]]], f32)
    a = _mat2_inverse(a)
    ab = _mat2_mul(a, b)
    ca = _mat2_mul(c, a)
    cab = _mat2_mul(ca, b)
    dcab = (d - cab).astype(f32)
    det = f32(f32(dcab[1][1] * dcab[0][0]) - f32(dcab[1][0] * dcab[0][1]))
    if abs(det) <= np.finfo(f32).eps:
        return np.eye(4, dtype=f32)
    d1 = _mat2_inverse(dcab)
    d2 = (-d1).astype(f32)
    c1 = _mat2_mul(d2, ca)
    b1 = _mat2_mul(ab, d2)
    a1 = (a - _mat2_mul(ab, c1)).astype(f32)
    return np.array([[a1[0][0], a1[0][1], c1[0][0], c1[0][1]],
                     [a1[1][0], a1[1][1], c1[1][0], c1[1][1]],
                     [b1[0][0], b1[0][1], d1[0][0], d1[0][1]],
                     [b1[1][0], b1[1][1], d1[1][0], d1[1][1]]], f32)


def world_to_scene(scales) -> np.ndarray:
    s = np.asarray(scales, f32)
    inv = (f32(1) / s).astype(f32)
    t = (f32(0.5) * s).astype(f32)
    m = np.zeros((4, 4), f32)
    for k in range(3):
        m[k][k] = inv[k]
        m[3][k] = f32(inv[k] * t[k])
    m[3][3] = f32(1)
    return m


def volume_scales(spaces, dim) -> np.ndarray:
    """app/dvr_dpbx_vdb.cpp:202-210: spaces / min(dim) * dim."""
    spaces, dimf = np.asarray(spaces, f32), np.asarray(dim, f32)
    return ((spaces / f32(min(dim))).astype(f32) * dimf).astype(f32)


class FPSCamera:
    def __init__(self, eye=(1, 1, 1), center=(0, 0, 0), up=(0, 1, 0)):
        self.LookAt(eye, center, up)

    def LookAt(self, eye, center, up=(0, 1, 0)):
        self.pos = np.asarray(eye, f32).copy()
        self.forward = _normalize(np.asarray(center, f32) - self.pos)
        self.world_up = np.asarray(up, f32).copy()
        self.right = _normalize(_cross(self.forward, self.world_up))
        self.up = _normalize(_cross(self.right, self.forward))
        self.pitch = _c("asinf", self.forward[1])
        cp = _c("cosf", self.pitch)
        if abs(self.forward[0]) > abs(self.forward[2]):
            self.yaw = _c("acosf", f32(self.forward[0] / cp))
        else:
            self.yaw = _c("asinf", f32(-self.forward[2] / cp))

    def Revolve(self, fwd_dist, yaw_dif_deg, pitch_dif_deg):
        pitch_dif = radians(f32(-f32(pitch_dif_deg)))
        yaw_dif = radians(f32(-f32(yaw_dif_deg)))
        limit = radians(60.0)
        self.pitch = f32(self.pitch + pitch_dif)
        if self.pitch > limit:
            self.pitch = limit
        elif self.pitch < -limit:
            self.pitch = f32(-limit)
        self.yaw = f32(self.yaw + yaw_dif)
        pi = f32(3.14159265358979323846264338327950288)
        if self.yaw > pi:
            self.yaw = f32(self.yaw - f32(f32(2) * pi))
        elif self.yaw < -pi:
            self.yaw = f32(self.yaw + f32(f32(2) * pi))
        fwd_dist = f32(fwd_dist)
        center = (self.pos + fwd_dist * self.forward).astype(f32)
        cy, sy = _c("cosf", self.yaw), _c("sinf", self.yaw)
        cp, sp = _c("cosf", self.pitch), _c("sinf", self.pitch)
        self.forward = _normalize(np.array([f32(cp * cy), sp, f32(f32(-cp) * sy)], f32))
        self.right = _normalize(_cross(self.forward, self.world_up))
        self.up = _normalize(_cross(self.right, self.forward))
        self.pos = (center - fwd_dist * self.forward).astype(f32)

    def GetPos(self):
        return self.pos

    def GetRFUP(self):
        return self.right, self.forward, self.up, self.pos

    def rotation(self) -> np.ndarray:
        """glm::mat3(R, U, -F) as the applications pass it to SetCameraToWorld."""
        return np.array([self.right, self.up, -self.forward], f32)

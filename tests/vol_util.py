"""Synthetic volumes and DVR set-ups shared by the volume-path tests."""
import numpy as np

from raytracer_amd import synthetic as syn
from raytracer_amd.data import camera as cam
from raytracer_amd.data.transfer_function import (README_HEAD_TF, README_KINGSNAKE_TF,
                                                   TransferFunction)


def blob_volume(shape, seed=1, dtype=np.uint8, fill=0.35):
    """Smooth blobs over an empty background; edges need not be multiples of the brick size."""
    dz, dy, dx = shape
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(np.arange(dz), np.arange(dy), np.arange(dx), indexing="ij")
    v = np.zeros(shape, np.float64)
    for _ in range(6):
        c = rng.uniform(0.15, 0.85, 3) * np.array([dz, dy, dx])
        r = rng.uniform(0.08, 0.25) * min(shape)
        d = np.sqrt((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2)
        v = np.maximum(v, np.clip(1.0 - d / r, 0, 1))
    noise = rng.uniform(0, 1, shape)
    top = np.iinfo(dtype).max
    out = np.where(v > 0, (0.25 + 0.75 * v) * top, noise * 0.15 * top)
    return out.astype(dtype)


def dvr_setup(dim_xyz, spaces=(1, 1, 1), W=160, H=120, eye_scale=1.0, orbit_steps=0):
    scales = cam.volume_scales(spaces, dim_xyz)
    w2s = cam.world_to_scene(scales)
    c = cam.FPSCamera((0, 0, float(max(scales)) * eye_scale), (0, 0, 0))
    for _ in range(orbit_steps):
        c.Revolve(cam._c("sqrtf", cam._dot(c.pos, c.pos)), 0.5, 0.0)
    proj = cam.perspective_fov(cam.radians(90.0), W, H, 1.0, 3.0)
    return w2s, c, proj


HEAD_TF = TransferFunction(README_HEAD_TF)
SNAKE_TF = TransferFunction(README_KINGSNAKE_TF)

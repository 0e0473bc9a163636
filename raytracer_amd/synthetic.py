"""Deterministic synthetic inputs of the shapes the benchmark configurations name.

Everything is driven by a splitmix64 counter generator so that any slice of a data set can be produced
independently (a rank can generate only its own rays, a test only the first 100k triangles) and is the
same on every machine. A float in [0, 1) is made from the top 24 bits: (x >> 40) * 2**-24.
"""
from __future__ import annotations

import numpy as np

SEED_SOUP = 0x5EED0001
SEED_RAYS = 0x5EED0002
SEED_HEAD = 0x5EED0003
SEED_SNAKE = 0x5EED0004

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, idx: np.ndarray) -> np.ndarray:
    """Value number `idx` (uint64 array) of the splitmix64 stream started at `seed`."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, idx: np.ndarray) -> np.ndarray:
    """float32 in [0, 1) from the top 24 bits."""
    return ((splitmix64(seed, idx) >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def triangle_soup(n_tri: int, seed: int = SEED_SOUP, first: int = 0, half_extent: float = 0.004,
                  chunk: int = 1 << 20):
    """Non-indexed triangle soup: triangle k has a centre c ~ U[0,1)^3 and vertices
    c + half_extent * U[-1,1)^3. Returns (positions float32 [3n,3], indices uint32 [n,3]).
    Twelve draws per triangle: counter = 12 * k + j."""
    pos = np.empty((n_tri, 3, 3), dtype=np.float32)
    he = np.float32(half_extent)
    for s in range(0, n_tri, chunk):
        e = min(n_tri, s + chunk)
        k = np.arange(first + s, first + e, dtype=np.uint64)
        base = k[:, None] * np.uint64(12) + np.arange(12, dtype=np.uint64)[None, :]
        u = uniform01(seed, base)  # [m, 12]
        c = u[:, 0:3]
        off = (np.float32(2.0) * u[:, 3:12] - np.float32(1.0)).reshape(-1, 3, 3)
        pos[s:e] = c[:, None, :] + he * off
    idx = np.arange(3 * n_tri, dtype=np.uint32).reshape(n_tri, 3)
    return pos.reshape(-1, 3), idx


def incoherent_rays(n_rays: int, seed: int = SEED_RAYS, first: int = 0) -> np.ndarray:
    """Rays with origin ~ U[0,1)^3 and direction uniform on the sphere (z = 1 - 2u, phi = 2 pi v).
    Returns float32 [n, 6] = (pos, dir). Five draws per ray: counter = 5 * k + j."""
    k = np.arange(first, first + n_rays, dtype=np.uint64)
    base = k[:, None] * np.uint64(5) + np.arange(5, dtype=np.uint64)[None, :]
    u = uniform01(seed, base).astype(np.float64)
    z = 1.0 - 2.0 * u[:, 3]
    phi = 2.0 * np.pi * u[:, 4]
    s = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    rays = np.empty((n_rays, 6), dtype=np.float32)
    rays[:, 0:3] = u[:, 0:3]
    rays[:, 3] = s * np.cos(phi)
    rays[:, 4] = s * np.sin(phi)
    rays[:, 5] = z
    return rays


def grid_test_mesh(dim, sz=(1.0, 1.0, 1.0)):
    """One triangle per grid cell, as the reference's OBJMesh::CreateGridTestMesh builds it
    (include/data/mesh.h:255-284): the cell origin advances by repeated float addition."""
    dim = [int(d) for d in dim]
    d_sz = (np.array(sz, dtype=np.float32) / np.array(dim, dtype=np.float32)).astype(np.float32)
    n = dim[0] * dim[1] * dim[2]
    pos = np.empty((n, 3, 3), dtype=np.float32)
    g = np.zeros(3, dtype=np.float32)
    k = 0
    for _z in range(dim[2]):
        g[1] = 0
        for _y in range(dim[1]):
            g[0] = 0
            for _x in range(dim[0]):
                pos[k, 0] = g
                pos[k, 1] = g + np.array([d_sz[0], 0, 0], dtype=np.float32)
                pos[k, 2] = g + np.array([0, d_sz[1], d_sz[2]], dtype=np.float32)
                k += 1
                g[0] = np.float32(g[0] + d_sz[0])
            g[1] = np.float32(g[1] + d_sz[1])
        g[2] = np.float32(g[2] + d_sz[2])
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return pos.reshape(-1, 3), idx


def _hash_noise(seed: int, x: np.ndarray, y: np.ndarray, z: np.ndarray) -> np.ndarray:
    idx = (z.astype(np.uint64) << np.uint64(40)) | (y.astype(np.uint64) << np.uint64(20)) | x.astype(np.uint64)
    return uniform01(seed, idx)


def head_volume(dim=(128, 256, 256), seed: int = SEED_HEAD) -> np.ndarray:
    """Head-shaped uint8 volume, array index order [z, y, x]: nested ellipsoid shells.
    r = |p / (0.80, 0.90, 0.90)| with p in [-1,1]^3:
      r > 0.85       : 20 + 10*noise (empty under the Head transfer function, whose range [0,49] has zero opacity)
      0.78 < r <= .85: 65 (skin)
      0.55 < r <= .78: 128 + 60*smooth (bone)
      r <= 0.55      : 90 (interior), all with a little hash noise so bricks are not constant."""
    dx, dy, dz = dim
    out = np.empty((dz, dy, dx), dtype=np.uint8)
    xs = np.arange(dx)
    ys = np.arange(dy)
    px = ((xs + 0.5) / dx * 2.0 - 1.0) / 0.80
    py = ((ys + 0.5) / dy * 2.0 - 1.0) / 0.90
    X, Y = np.meshgrid(xs, ys, indexing="xy")  # [dy, dx]
    PX, PY = np.meshgrid(px, py, indexing="xy")
    for z in range(dz):
        pz = ((z + 0.5) / dz * 2.0 - 1.0) / 0.90
        r = np.sqrt(PX * PX + PY * PY + pz * pz)
        nz = _hash_noise(seed, X, Y, np.full_like(X, z)).astype(np.float64)
        v = np.where(r > 0.85, 20.0 + 10.0 * nz, 0.0)
        v = np.where((r > 0.78) & (r <= 0.85), 65.0 + 4.0 * nz, v)
        t = np.clip((0.78 - r) / (0.78 - 0.55), 0.0, 1.0)
        smooth = t * t * (3.0 - 2.0 * t)
        v = np.where((r > 0.55) & (r <= 0.78), 128.0 + 60.0 * smooth + 6.0 * nz, v)
        v = np.where(r <= 0.55, 90.0 + 6.0 * nz, v)
        out[z] = np.clip(v, 0, 255).astype(np.uint8)
    return out


def kingsnake_volume(dim=(1024, 1024, 795), seed: int = SEED_SNAKE, turns: float = 6.0,
                     coil_radius: float = 0.30, tube_radius: float = 0.045) -> np.ndarray:
    """Kingsnake-shaped sparse uint8 volume [z, y, x]: a helical tube (value 130..255, radial
    profile) over a background of 30 + 40*noise (< 95, empty under the Kingsnake transfer function)."""
    dx, dy, dz = dim
    out = np.empty((dz, dy, dx), dtype=np.uint8)
    xs = np.arange(dx)
    ys = np.arange(dy)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    PX = (X + 0.5) / dx - 0.5
    PY = (Y + 0.5) / dy - 0.5
    for z in range(dz):
        pz = (z + 0.5) / dz
        ang = 2.0 * np.pi * turns * pz
        cx, cy = coil_radius * np.cos(ang), coil_radius * np.sin(ang)
        # distance to the helix centre line, measured in the slice (the pitch is shallow)
        d = np.sqrt((PX - cx) ** 2 + (PY - cy) ** 2)
        nz = _hash_noise(seed, X, Y, np.full_like(X, z)).astype(np.float64)
        bg = 30.0 + 40.0 * nz
        prof = np.clip(1.0 - d / tube_radius, 0.0, 1.0)
        tube = 130.0 + 125.0 * prof * prof * (3.0 - 2.0 * prof)
        out[z] = np.where(d < tube_radius, tube, bg).astype(np.uint8)
    return out


def kingsnake_volume_torch(dim=(1024, 1024, 795), device="cuda:0", seed: int = SEED_SNAKE,
                           turns: float = 6.0, coil_radius: float = 0.30, tube_radius: float = 0.045,
                           slab: int = 32):
    """The Kingsnake-shaped volume generated on the GPU, slab by slab (833 MB at full size is not a
    fixture): same helical tube as kingsnake_volume; the background noise comes from a 32-bit integer
    hash of the voxel index instead of splitmix64. Returns a uint8 torch tensor [dz, dy, dx]."""
    import torch
    dx, dy, dz = dim
    out = torch.empty((dz, dy, dx), dtype=torch.uint8, device=device)
    xs = torch.arange(dx, device=device, dtype=torch.float32)
    ys = torch.arange(dy, device=device, dtype=torch.float32)
    PX = ((xs + 0.5) / dx - 0.5)[None, None, :]
    PY = ((ys + 0.5) / dy - 0.5)[None, :, None]
    ix = torch.arange(dx, device=device, dtype=torch.int64)[None, None, :]
    iy = torch.arange(dy, device=device, dtype=torch.int64)[None, :, None]
    for z0 in range(0, dz, slab):
        z1 = min(dz, z0 + slab)
        zz = torch.arange(z0, z1, device=device, dtype=torch.float32)
        ang = (2.0 * np.pi * turns) * ((zz + 0.5) / dz)
        cx = (coil_radius * torch.cos(ang))[:, None, None]
        cy = (coil_radius * torch.sin(ang))[:, None, None]
        d = torch.sqrt((PX - cx) ** 2 + (PY - cy) ** 2)
        iz = torch.arange(z0, z1, device=device, dtype=torch.int64)[:, None, None]
        h = (ix * 73856093 + iy * 19349663 + iz * 83492791 + seed) & 0xFFFFFFFF
        h = ((h ^ (h >> 15)) * 0x85EBCA77) & 0xFFFFFFFF
        h = ((h ^ (h >> 13)) * 0xC2B2AE3D) & 0xFFFFFFFF
        h = h ^ (h >> 16)
        noise = (h >> 8).to(torch.float32) * (1.0 / 16777216.0)
        bg = 30.0 + 40.0 * noise
        prof = torch.clamp(1.0 - d / tube_radius, 0.0, 1.0)
        tube = 130.0 + 125.0 * prof * prof * (3.0 - 2.0 * prof)
        out[z0:z1] = torch.where(d < tube_radius, tube, bg).to(torch.uint8)
    return out

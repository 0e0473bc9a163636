"""RAW volume file loader with the reference's axis transform (include/data/raw_volume.h:26-103).

tr_axis = (a, b, c), a permutation of +-1, +-2, +-3: output axis X takes input axis |a| (reversed when
a < 0), and so on; the output dimension is dim[|a|-1], dim[|b|-1], dim[|c|-1]. Voxels are x fastest."""
from __future__ import annotations

import os
import sys

import numpy as np

ErrTag = "[RAWVolume Error]"
VOXEL_TYPES = {"uint8": np.uint8, "uint16": np.uint16}  # fromNameToVoxelType, app/dvr_dpbx_vdb.cpp


class RAWVolume:
    def __init__(self, path: str, dim, tr_axis=(1, 2, 3), dtype=np.uint8):
        self.is_complete = False
        self.dat = None  # numpy [dz, dy, dx]
        self.dim = None  # (dx, dy, dz) after the transform
        dtype = np.dtype(dtype)
        dim = tuple(int(d) for d in dim)
        tr = tuple(int(t) for t in tr_axis)
        if sorted(abs(t) for t in tr) != [1, 2, 3]:
            print(f"{ErrTag} Parameter trAxis is invalid.", file=sys.stderr)
            return
        vox_num = dim[0] * dim[1] * dim[2]
        try:
            size = os.path.getsize(path)
        except OSError:
            print(f'{ErrTag} Cannot open file at "{path}".', file=sys.stderr)
            return
        if size < vox_num * dtype.itemsize:
            print(f"{ErrTag} File {path} is too small to contain volume of size "
                  f"({dim[0]},{dim[1]},{dim[2]})", file=sys.stderr)
            return
        a = np.fromfile(path, dtype=dtype, count=vox_num).reshape(dim[2], dim[1], dim[0])
        self.dat, self.dim = transform_axes(a, tr)
        self.is_complete = True

    def IsComplete(self) -> bool:
        return self.is_complete

    def GetDimension(self):
        return self.dim


def transform_axes(a: np.ndarray, tr_axis):
    """a: [dz, dy, dx] -> (transformed array [dz', dy', dx'], (dx', dy', dz'))."""
    m = [abs(t) - 1 for t in tr_axis]  # output coordinate k reads input coordinate m[k]
    # numpy axis of coordinate c (0 = x) is 2 - c
    out = a.transpose(2 - m[2], 2 - m[1], 2 - m[0])
    for k in range(3):
        if tr_axis[k] < 0:
            out = np.flip(out, axis=2 - k)
    out = np.ascontiguousarray(out)
    return out, (out.shape[2], out.shape[1], out.shape[0])

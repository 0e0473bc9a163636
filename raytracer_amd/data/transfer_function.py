"""Transfer function: control points -> 256-entry RGBA table + empty scalar ranges, with the semantics
of kouek::Data::TransferFunction<T, Min, Max, 1> (include/data/transfer_function.h:34-127), including
its quirks: the interpolation cursor advances by at most one control point per scalar (:108-109), the
last table entry is the cursor's point (:126), and an empty range is only emitted once a non-empty
segment follows it (:113-121). Ranges are stored truncated to the voxel type.
"""
from __future__ import annotations

import numpy as np

from .mesh import _scan_floats

f32 = np.float32

# The two tables printed in the reference's README.md:69-87
README_HEAD_TF = """0.000000 0.000000 0.000000 0.000000 0.000000
49.840766 0.000000 0.000000 0.000000 0.000000
60.016020 5.041094 3.602943 52.500004 0.579299
127.955641 255.000000 255.000000 255.000000 5.000000
183.914396 8.000000 255.000000 0.000000 8.000000
255.000000 255.000000 0.000000 0.000000 10.000000
"""
README_KINGSNAKE_TF = """0.000000 0.000000 0.000000 0.000000 0.000000
95.370346 0.000000 0.000000 0.000000 0.000000
125.946533 255.000000 255.000000 255.000000 20.000001
159.491972 255.000000 157.499998 0.000000 20.000001
232.292712 30.000000 0.000000 255.000000 100.000002
255.000000 255.000000 0.000000 0.000000 255.000000
"""


class TransferFunction:
    def __init__(self, text: str | None = None, path: str | None = None, voxel_dtype=np.uint8):
        self.is_complete = False
        self.error = ""
        self.voxel_dtype = np.dtype(voxel_dtype)
        self.min_scalar = 0
        self.max_scalar = int(np.iinfo(self.voxel_dtype).max)
        self.points = {}
        self.dat = None
        self.empty_scalar_ranges = None
        if path is not None:
            try:
                text = open(path, "r", errors="replace").read()
            except OSError:
                self.error = f'Cannot open file at "{path}".'
                return
        lo, hi = f32(self.max_scalar), f32(self.min_scalar)
        for ln, line in enumerate(text.splitlines()):
            v = _scan_floats(line, 5)
            if len(v) != 5:
                self.error = f"Line {ln} is of invalid format."
                return
            scalar = v[0]
            lo, hi = min(lo, scalar), max(hi, scalar)
            col = (np.array(v[1:], f32) / f32(255.0)).astype(f32)
            self.points.setdefault(float(scalar), (scalar, col))  # std::map::emplace keeps the first
        if lo != self.min_scalar or hi != self.max_scalar:
            self.error = "not enough TF points"
            return
        self._compute()
        self.is_complete = True

    def _compute(self):
        pts = [self.points[k] for k in sorted(self.points)]
        Min, Max = self.min_scalar, self.max_scalar
        dat, ranges = [], []
        pi = prev = 0
        rng = [f32(Max), f32(Min)]
        for scalar in range(Min, Max):
            if f32(scalar) > pts[pi][0]:
                prev = pi
                pi += 1
            s, c = pts[pi]
            ps, pc = pts[prev]
            if pc[3] == 0 and c[3] == 0:
                if rng[0] > ps:
                    rng[0] = ps
                if rng[1] < s:
                    rng[1] = s
            elif rng[0] != f32(Max) and rng[1] != f32(Min):
                ranges.append((rng[0], rng[1]))
                rng = [f32(Max), f32(Min)]
            a = f32(1) if s == ps else f32(f32(f32(scalar) - ps) / f32(s - ps))
            dat.append((f32(f32(1) - a) * pc + a * c).astype(f32))
        dat.append(pts[pi][1])
        self.dat = np.array(dat, f32)
        # glm::vec<2, T> from a vec2: truncation towards zero
        self.empty_scalar_ranges = np.array([[int(r[0]), int(r[1])] for r in ranges],
                                            self.voxel_dtype).reshape(-1, 2)

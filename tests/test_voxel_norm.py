"""The marcher turns a 16-bit voxel v into float(v) / 65535 (and the dense-volume caster an 8-bit one into
float(v) / 255) without dividing: q0 = v * r with r the rounded reciprocal, e = fma(-q0, max, v),
q = fma(e, r, q0). This must be the correctly rounded quotient - what the reference's division gives - for
every voxel value; checked here in exact rational arithmetic (CPU only)."""
from fractions import Fraction

import numpy as np
import pytest

f32 = np.float32


def _round(fr: Fraction) -> np.float32:
    """fr rounded to the nearest float32, ties to even."""
    if fr == 0:
        return f32(0)
    x = f32(float(fr))  # (double rounding is put right by looking at the neighbours)
    cands = [x, np.nextafter(x, f32(np.inf)), np.nextafter(x, f32(-np.inf))]
    return f32(min(cands, key=lambda c: (abs(Fraction(float(c)) - fr),
                                         int(np.frombuffer(f32(c).tobytes(), dtype=np.uint32)[0]) & 1)))


@pytest.mark.parametrize("top", [255, 65535])
def test_reciprocal_with_one_correction_is_the_rounded_quotient(top):
    r = _round(Fraction(1, top))
    assert r == f32(1) / f32(top)
    for v in range(top + 1):
        q0 = _round(Fraction(v) * Fraction(float(r)))
        e = _round(Fraction(v) - Fraction(float(q0)) * top)
        q = _round(Fraction(float(e)) * Fraction(float(r)) + Fraction(float(q0)))
        assert q == f32(v) / f32(top), v

"""Instruction-issue roofline of kernels that are not bound by HBM (the DVR marcher, the path tracer).

A wave64 VALU instruction holds its SIMD for four cycles, so a kernel that issues I wave instructions
cannot take less than 4 * I / (SIMDs * clock): 1,024 SIMDs (256 CUs x 4) at the 2.4 GHz peak clock of
MI355X. I is not measurable inside a plain run; it is read from the counter digest committed under
profiles/ (SQ_INSTS_VALU of a rocprofv3 --pmc pass over the same kernel and workload) and named as such."""
from __future__ import annotations

import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 1024
CLOCK_HZ = 2.4e9
CYCLES_PER_WAVE_INSTRUCTION = 4


def _digest(names):
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                return name, json.load(open(path))
            except (OSError, ValueError):
                pass
    return None, None


def valu_issue(digests, ms_measured: float, dispatches_per_unit: float = 1.0, kernel_contains: str | None = None,
               unit: str = "frame") -> dict | None:
    """roofline dict for one `unit` (frame) that took ms_measured and is made of dispatches_per_unit
    dispatches of the profiled kernel. digests: file names under profiles/, newest first."""
    name, d = _digest(digests)
    if d is None:
        return None
    if kernel_contains is not None and "valu_instructions" not in d:
        d = next((v for k, v in d.items() if kernel_contains in k), None)
        if d is None:
            return None
    insts = float(d["valu_instructions"]) * dispatches_per_unit
    floor_ms = CYCLES_PER_WAVE_INSTRUCTION * insts / (SIMDS * CLOCK_HZ) * 1e3
    out = {"bound": "valu_issue", "unit": "ms", "peak": round(floor_ms, 4), "achieved": round(ms_measured, 4),
           "frac": round(floor_ms / ms_measured, 4),
           f"wave_valu_instructions_per_{unit}": int(insts),
           "floor": f"{CYCLES_PER_WAVE_INSTRUCTION} cycles x wave instructions / ({SIMDS} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz)",
           "instructions_source": f"profiles/{name} (SQ_INSTS_VALU of a rocprofv3 --pmc pass over the same kernel "
                                  f"and workload: {d.get('workload', '')}), not measured in this run",
           "valu_lane_utilisation_from_profile": round(float(d.get("valu_lane_utilisation", 0.0)), 3)}
    for k in ("hbm_read_bytes", "hbm_write_bytes"):
        if k in d:
            out[k + "_from_profile"] = int(d[k])
    return out

"""Instruction-issue roofline of kernels that are not bound by HBM (the DVR marcher, the path tracer).

The floor: a kernel that issues I wave64 VALU instructions cannot take less than I / (SIMDs x rate), with
1,024 SIMDs (256 CUs x 4) and rate = 0.6 G wave instructions per second and SIMD (four cycles per
instruction at the 2.4 GHz peak clock). What a SIMD of an MI355X sustains was measured with streams of
independent instructions on every SIMD at once, 5 to 8 waves per SIMD (scripts/valu_rate.hip,
profiles/r03_valu_issue_rate.txt; rates by wall time, the shader clock follows the load):
v_cvt_f32_ubyte, v_max3_f32, v_pk_fma_f32 and v_mul_lo_u32 0.55-0.58 G/s, plain v_fma_f32 0.84-0.91, a mix
of two conversions, a max3 and an fma (the make-up of a wide-node visit) 0.71-0.73; one wave alone 0.42-0.52.
The kernels priced here are conversions, packed fma, min / max and integer work: 0.6 sits between what
those classes and the mix reach, and the fraction at the mix's rate is reported beside the headline one.
I is not measurable inside a plain run; it is read from the counter digest committed under profiles/
(SQ_INSTS_VALU of a rocprofv3 --pmc pass over the same kernel and workload) and named as such."""
from __future__ import annotations

import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 1024
CLOCK_HZ = 2.4e9
CYCLES_PER_WAVE_INSTRUCTION = 4
MIX_RATE_PER_SIMD = 0.72e9  # measured: profiles/r03_valu_issue_rate.txt, the mixed stream at 5-8 waves per SIMD


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
           "frac_at_measured_mix_rate": round(insts / (SIMDS * MIX_RATE_PER_SIMD) * 1e3 / ms_measured, 4),
           "issue_rate_source": "profiles/r03_valu_issue_rate.txt (scripts/valu_rate.hip): 0.55-0.58 G wave instructions/s "
                                "per SIMD for conversions, max3, packed fma and integer multiply, 0.71-0.73 for a mix with "
                                "plain fma, 0.84-0.91 for plain fma alone; the floor uses 0.6, the second fraction 0.72",
           f"wave_valu_instructions_per_{unit}": int(insts),
           "floor": f"{CYCLES_PER_WAVE_INSTRUCTION} cycles x wave instructions / ({SIMDS} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz)",
           "instructions_source": f"profiles/{name} (SQ_INSTS_VALU of a rocprofv3 --pmc pass over the same kernel "
                                  f"and workload: {d.get('workload', '')}), not measured in this run",
           "valu_lane_utilisation_from_profile": round(float(d.get("valu_lane_utilisation", 0.0)), 3)}
    for k in ("hbm_read_bytes", "hbm_write_bytes"):
        if k in d:
            out[k + "_from_profile"] = int(d[k])
    return out

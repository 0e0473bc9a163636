"""profiles/<tag>_pt_pmc.json from a PMC summary (scripts/pmc_pt.sh): what the path-tracing kernel's time
goes to (veach-mis 1920x1080, scripts/prof_small.py pt)."""
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/summary_pmc_pt.json"
d = json.load(open(src))


def digest(name, c):
    cycles = c["GRBM_GUI_ACTIVE"] / 8  # the counter sums the 8 XCDs
    simds, cus = 1024, 256
    out = {
        "kernel": name,
        "workload": "veach-mis 1920x1080 (scripts/prof_small.py pt: the <false, 3> variant renders one sample per "
                    "pixel and dispatch, the <false, 4> variant four), mean of "
                    f"{c['dispatches']} dispatches",
        "kernel_cycles": cycles,
        "waves": c["SQ_WAVES"],
        "valu_instructions": c["SQ_INSTS_VALU"],
        "salu_instructions": c["SQ_INSTS_SALU"],
        "vector_load_instructions": c["SQ_INSTS_VMEM_RD"],
        "lds_instructions": c["SQ_INSTS_LDS"],
        # thread-cycles of vector ALU work over 64 lanes x the cycles vector ALU instructions were executing
        "valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]),
        "valu_lane_utilisation_per_instruction_at_4_cycles": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 4 * 64),
        # a wave64 fp32 instruction holds its SIMD for 4 cycles
        "valu_busy_frac_at_4_cycles_per_instruction": c["SQ_INSTS_VALU"] * 4 / simds / cycles,
        "wave_cycles_waiting_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "ta_busy_frac": c["TA_TA_BUSY_sum"] / cus / cycles,
        "td_busy_frac": c["TD_TD_BUSY_sum"] / cus / cycles,
        "l1_line_accesses": c["TCP_TOTAL_CACHE_ACCESSES_sum"],
        "l1_to_l2_read_requests": c["TCP_TCC_READ_REQ_sum"],
        "l1_hit_rate": 1 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"],
    }
    return out


out = {name: digest(name, c) for name, c in d.items() if "path_trace_persistent_kernel" in name}
json.dump(out, open("profiles/%s_pt_pmc.json" % os.environ.get("KK_PROFILE_TAG", "r02"), "w"), indent=1)
print(json.dumps(out, indent=1))

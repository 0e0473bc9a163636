"""profiles/<tag>_dvr_pmc.json from a PMC summary (scripts/collect_profiles.sh plus the dtex and dsq2 passes
of scripts/pmc_pass.sh): what the DVR marcher's time goes to."""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/summary_pmc_all.json"
head = len(sys.argv) > 2 and sys.argv[2] == "head"  # the Head-shape volume at 1080p (prof_small.py dvr_head)
d = json.load(open(src))
name = next(k for k in d if "dvr_kernel<unsigned char, false, false" in k)
c = d[name]
cycles = c["GRBM_GUI_ACTIVE"] / 8  # the counter sums the 8 XCDs
simds, cus = 1024, 256
out = {
    "kernel": name,
    "workload": ("Head-shape 128x256x256 uint8 at 1920x1080 (scripts/prof_small.py dvr_head), mean of " if head else
                 "Kingsnake-shape 1024x1024x795 uint8 at 3840x2160 (scripts/prof_small.py dvr), mean of ") +
                f"{c['dispatches']} dispatches",
    "kernel_cycles": cycles,
    "waves": c["SQ_WAVES"],
    "valu_instructions": c["SQ_INSTS_VALU"],
    "salu_instructions": c["SQ_INSTS_SALU"],
    "branch_instructions": c["SQ_INSTS_BRANCH"],
    "vector_load_instructions": c["SQ_INSTS_VMEM_RD"],
    "lds_instructions": c["SQ_INSTS_LDS"],
    "valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64),
    # a wave64 fp32 instruction holds its SIMD for 4 cycles
    "valu_busy_frac_at_4_cycles_per_instruction": c["SQ_INSTS_VALU"] * 4 / simds / cycles,
    "wave_cycles_waiting_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "ta_busy_frac": c["TA_TA_BUSY_sum"] / cus / cycles,
    "td_busy_frac": c["TD_TD_BUSY_sum"] / cus / cycles,
    "l1_line_accesses": c["TCP_TOTAL_CACHE_ACCESSES_sum"],
    "l1_to_l2_read_requests": c["TCP_TCC_READ_REQ_sum"],
    "l1_hit_rate": 1 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"],
}
if "TCC_REQ_sum" in c:
    out.update({"l2_hit_rate": c["TCC_HIT_sum"] / c["TCC_REQ_sum"],
                "hbm_read_bytes": c["TCC_EA0_RDREQ_sum"] * 128,
                "hbm_write_bytes": c["WRITE_SIZE"] * 1024})
json.dump(out, open("profiles/%s_dvr%s_pmc.json" % (__import__("os").environ.get("KK_PROFILE_TAG", "r02"),
                                                      "_head" if head else ""), "w"), indent=1)
print(json.dumps(out, indent=1))

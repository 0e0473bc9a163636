"""profiles/<tag>_closest_hit_pmc.json from a PMC summary (scripts/collect_profiles.sh): HBM-side bytes of the
closest-hit kernel per launch, as bench.py reports them in roofline.traffic."""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/summary_pmc_all.json"
d = json.load(open(src))
name = next(k for k in d if "closest_hit_accel_kernel<false, false" in k)
c = d[name]
reads = c["TCC_EA0_RDREQ_sum"] * 128  # every read request of this kernel is a 128-byte line
writes = c["WRITE_SIZE"] * 1024
out = {
    "kernel": name,
    "workload": "2,097,152 incoherent rays through the 10 M-triangle soup (scripts/prof_small.py), mean of "
                f"{c['dispatches']} dispatches",
    "method": "separate rocprofv3 --pmc passes (scripts/collect_profiles.sh). Reads: TCC_EA0_RDREQ x 128 B "
              "(TCC_EA0_RDREQ_32B is 0 and a size-split pass showed 128-byte lines only; on gfx950 FETCH_SIZE "
              "tallies every request at 64 B, so FETCH_SIZE x 1024 is half of this figure). Writes: "
              "WRITE_SIZE x 1024 (scattered 4-byte result stores).",
    "TCC_EA0_RDREQ_128B": c["TCC_EA0_RDREQ_sum"],
    "FETCH_SIZE_KB": c["FETCH_SIZE"],
    "WRITE_SIZE_KB": c["WRITE_SIZE"],
    "TCC_HIT": c["TCC_HIT_sum"],
    "TCC_MISS": c["TCC_MISS_sum"],
    "read_bytes_per_launch": int(reads),
    "write_bytes_per_launch": int(writes),
    "hbm_bytes_per_launch": int(reads + writes),
    "valu_instructions_per_launch": c["SQ_INSTS_VALU"],
    "valu_lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64),
    "l2_hit_rate": c["TCC_HIT_sum"] / c["TCC_REQ_sum"],
}
if "GRBM_GUI_ACTIVE" in c:
    # vector-memory path: busy cycles summed over the 256 CUs' units against the kernel's cycles
    # (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    cycles = c["GRBM_GUI_ACTIVE"] / 8
    out["texture_path"] = {
        "kernel_cycles": cycles,
        "ta_busy_frac": c["TA_TA_BUSY_sum"] / 256 / cycles,
        "td_busy_frac": c["TD_TD_BUSY_sum"] / 256 / cycles,
        "l1_line_accesses_per_launch": c["TCP_TOTAL_CACHE_ACCESSES_sum"],
        "l1_to_l2_read_requests_per_launch": c["TCP_TCC_READ_REQ_sum"],
        "vector_load_wave_instructions_per_launch": c.get("TA_FLAT_READ_WAVEFRONTS_sum", c.get("SQ_INSTS_VMEM_RD")),
    }
json.dump(out, open("profiles/%s_closest_hit_pmc.json" % __import__("os").environ.get("KK_PROFILE_TAG", "r02"), "w"), indent=1)
print(json.dumps(out, indent=1))

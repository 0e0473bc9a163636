"""profiles/<tag>_build_pmc.json: per kernel of the LBVH build, the trace-side build and the VDB build,
the bytes moved to and from memory (separate rocprofv3 --pmc passes, scripts/collect_profiles.sh) beside
the kernel's average duration (kernel-trace stats of the same driver, scripts/prof_small.py)."""
import csv
import json
import sys

pmc = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/summary_pmc_all.json"))
stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2] if len(sys.argv) > 2 else
                                                   "gpurun_out/kernel_stats_all.csv"))}
groups = {"lbvh_build": ("kk::lbvh::", "kk::rsort::"), "trace_side_build": ("kk::accel::",),
          "vdb_build": ("kk::vdb::",)}
skip = ("closest_hit_accel_kernel", "ray_keys_kernel")
out = {"method": "reads = TCC_EA0_RDREQ x 128 B (every read request is a 128-byte line on gfx950), writes = "
                 "WRITE_SIZE x 1024 B; mean per dispatch; avg_us from rocprofv3 --kernel-trace --stats of the "
                 "same run (10 M-triangle soup; Kingsnake-shape 1024x1024x795 uint8 volume)"}
for g, prefixes in groups.items():
    rows = []
    for name, c in pmc.items():
        if not name.startswith(prefixes) and not any(p in name for p in prefixes):
            continue
        if any(s in name for s in skip) or "TCC_EA0_RDREQ_sum" not in c:
            continue
        st = next((v for k, v in stats.items() if k.startswith(name) or name in k), None)
        avg_us = float(st["AverageNs"]) / 1e3 if st else None
        rd, wr = c["TCC_EA0_RDREQ_sum"] * 128, c.get("WRITE_SIZE", 0) * 1024
        rows.append({"kernel": name, "launches_per_run": int(st["Calls"]) if st else None,
                     "avg_us": round(avg_us, 1) if avg_us else None,
                     "read_MB": round(rd / 1e6, 1), "write_MB": round(wr / 1e6, 1),
                     "GBps": round((rd + wr) / (avg_us * 1e-6) / 1e9, 1) if avg_us else None})
    out[g] = sorted(rows, key=lambda r: -(r["avg_us"] or 0))
json.dump(out, open("profiles/%s_build_pmc.json" % __import__("os").environ.get("KK_PROFILE_TAG", "r02"), "w"), indent=1)
for g in groups:
    print(g)
    for r in out[g]:
        print("  ", r)

"""Summarises rocprofv3 --pmc passes (counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv, glob, sys, json, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    if not k.startswith("kk::"):
        continue
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))

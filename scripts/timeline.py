"""Prints the kernels of the last LBVH build found in a rocprofv3 kernel-trace CSV."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'bounds_partial' in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(r['Kernel_Name'][:56].ljust(56), f"{d:9.1f} us   ends at {(int(r['End_Timestamp']) - t0) / 1e3:9.1f}")

"""Median duration per kernel (LBVH build and radix sort kernels) from a rocprofv3 --kernel-trace database:\n    python scripts/kernel_medians.py gpurun_out/<dir>/t_results.db"""
import sqlite3,sys,collections
c=sqlite3.connect(sys.argv[1])
rows=c.execute("select name, end-start from kernels").fetchall()
d=collections.defaultdict(list)
for n,t in rows: d[n.split('(')[0][-40:]].append(t/1e3)
for n,v in sorted(d.items(), key=lambda x:-sum(x[1])):
    if 'lbvh' in n or 'rsort' in n:
        v=sorted(v); print(f"{n:42s} n={len(v):3d} med={v[len(v)//2]:8.1f}")

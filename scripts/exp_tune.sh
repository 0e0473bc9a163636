#!/bin/bash
for t in 40,16,24 32,16,24 24,16,16 48,16,24 40,24,32 40,12,32 40,32,40 32,24,32; do
  echo "tune $t"; KK_TRACE_TUNE=$t timeout -k 10 100 python -u scripts/exp_trace.py 2>&1 | grep "sort=True" | cut -c1-70 || exit 1
done

#!/bin/bash
# Runs on the GPU box: the headline figures (closest hit, path tracer, marcher, LBVH build) for each build of
# the library named on the command line ("default" = the tree's own), twice in turn. usage:
#   gpurun -- 'bash scripts/ab_headlines.sh default variants/x.so'
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = default ]; then unset KK_LIB_PATH; else export KK_LIB_PATH=$PWD/$v; fi
    echo "== $v"
    timeout -k 10 200 python -u scripts/exp_ray_sort.py --one default 2>/dev/null | tail -1
    timeout -k 10 200 python -u scripts/pt_quick.py 2>/dev/null | tail -1
    timeout -k 10 200 python -u scripts/exp_dvr_order_pipelined.py "same stream" 2>/dev/null | tail -1
  done
done

#!/bin/bash
# Runs on the GPU box: the GPU test suite, the bench line, and rocprofv3 kernel stats of bench.py itself.
R=$GRAFT_REPO_ROOT
tag=${1:-r03}
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${tag}_gputest.log 2>&1 || { tail -30 gpurun_out/${tag}_gputest.log; exit 1; }
tail -2 gpurun_out/${tag}_gputest.log
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -30 gpurun_out/${tag}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_bench -- python3 $R/bench.py > $R/gpurun_out/${tag}_bench_prof.json 2> $R/gpurun_out/${tag}_bench_prof.err || { tail -30 $R/gpurun_out/${tag}_bench_prof.err; exit 1; }
cd $R
cp gpurun_out/stats_bench/*/*kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats.csv
head -c 600 gpurun_out/${tag}_bench.json

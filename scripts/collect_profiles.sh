#!/bin/bash
# Runs on the GPU box: kernel-trace stats and the PMC passes of the short profiling driver, each in a
# run of its own, then one JSON summary (mean counter value per dispatch and kernel).
set -e
what=${1:-all}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$what -- python3 $R/scripts/prof_small.py $what > $R/gpurun_out/stats_$what.log 2>&1
cd $R
bash scripts/pmc_pass.sh fetch $what FETCH_SIZE
bash scripts/pmc_pass.sh write $what WRITE_SIZE
bash scripts/pmc_pass.sh ea $what TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
bash scripts/pmc_pass.sh l2 $what TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
bash scripts/pmc_pass.sh sq $what SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
python3 scripts/pmc_summary.py gpurun_out > gpurun_out/summary_pmc_$what.json
cp gpurun_out/stats_$what/*/*kernel_stats.csv gpurun_out/kernel_stats_$what.csv

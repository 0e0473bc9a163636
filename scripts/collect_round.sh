#!/bin/bash
# Runs on the GPU box: every profile of a round. "all": kernel stats and counter passes of path 1 and the
# DVR marcher; "pt": those of the path-tracing kernel. Each rocprofv3 run is a pass of its own.
what=${1:-all}
if [ "$what" = all ]; then
  bash scripts/collect_profiles.sh all || exit 1
  bash scripts/pmc_pass.sh dtex all TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE || exit 1
  bash scripts/pmc_pass.sh dsq2 all SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM || exit 1
  python3 scripts/pmc_summary.py gpurun_out > gpurun_out/summary_pmc_all.json
elif [ "$what" = head ]; then
  # the marcher on the Head-shape volume at 1080p: instruction and memory-path counters only
  rm -rf gpurun_out/pmc_*
  bash scripts/pmc_pass.sh head_sq dvr_head SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU || exit 1
  bash scripts/pmc_pass.sh head_tex dvr_head TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE || exit 1
  bash scripts/pmc_pass.sh head_sq2 dvr_head SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM || exit 1
  python3 scripts/pmc_summary.py gpurun_out > gpurun_out/summary_pmc_head.json
else
  cd /tmp && export TMPDIR=/tmp
  R=$GRAFT_REPO_ROOT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_pt -- python3 $R/scripts/prof_small.py pt > $R/gpurun_out/stats_pt.log 2>&1 || exit 1
  cd $R
  cp gpurun_out/stats_pt/*/*kernel_stats.csv gpurun_out/kernel_stats_pt.csv
  bash scripts/pmc_pt.sh pt || exit 1
  python3 scripts/pmc_summary.py gpurun_out > gpurun_out/summary_pmc_pt.json
fi

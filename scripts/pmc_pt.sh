#!/bin/bash
# usage: pmc_pt.sh <tag>  -- SQ and memory-path counter passes of the path-tracing kernel (prof_small.py pt)
tag=$1
bash scripts/pmc_pass.sh ${tag}_sq pt SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU || exit 1
bash scripts/pmc_pass.sh ${tag}_mem pt TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM || exit 1

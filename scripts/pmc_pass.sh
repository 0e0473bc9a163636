#!/bin/bash
# usage: pmc_pass.sh <name> <what> <counters...>   -- one rocprofv3 --pmc pass of its own
name=$1; what=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$name -- python3 $GRAFT_REPO_ROOT/scripts/prof_small.py $what > $GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log 2>&1
echo "pass $name rc=$?"

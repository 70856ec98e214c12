#!/bin/bash
# the whole GPU suite, then the PMC passes behind roofline.traffic of the ReaxFF sweep (usage: tools/r05_full_check.sh <tag>)
T=${1:-r05_g}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_gputests.log 2>&1; rc=$?; tail -3 gpurun_out/${T}_gputests.log; [ $rc -eq 0 ] || { grep -E "^E |Error" gpurun_out/${T}_gputests.log | head -20; exit $rc; }
tools/pmc_reax.sh > gpurun_out/${T}_pmc_reax.log 2>&1 && tail -25 gpurun_out/${T}_pmc_reax.log

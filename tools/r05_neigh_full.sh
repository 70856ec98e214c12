#!/bin/bash
# whole GPU suite, then same-box kernel timing of library variants (usage: tools/r05_neigh_full.sh <tag> "name:ENV=.. ENV=.." ...)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=$1; shift
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${T}_gputests.log 2>&1; rc=$?; tail -3 gpurun_out/${T}_gputests.log; [ $rc -eq 0 ] || exit $rc
tools/neigh_whatif.sh $T "$@"

#!/bin/bash
# same-box kernel timing of library variants after one test file (usage: tools/r05_timing_only.sh <tag> <test file or ""> "name:ENV=.." ...)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=$1; F=$2; shift; shift
if [ -n "$F" ]; then timeout -k 10 600 python -m pytest $F -x -q -m gpu 2>&1 | tail -3; fi
tools/neigh_whatif.sh $T "$@"

#!/bin/bash
# quick GPU check of a k_neigh_build change: list-sensitive parity tests, wave clocks (timing build), same-box kernel timing against the round-4 library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py tests/test_gpu_properties.py -x -q -m gpu -s > gpurun_out/r05_quick_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r05_quick_tests.log; [ $rc -eq 0 ] || exit $rc
tools/r05_neigh_clocks.sh 576 2>&1 | grep -E "mode|per row|wave clocks"
tools/neigh_whatif.sh ${1:-r05_nc} "new:" "nokeep:SCEMA_MD_KEEP_LIST=0" "new2:"

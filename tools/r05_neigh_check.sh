#!/bin/bash
# GPU check of a k_neigh_build change: the whole GPU suite with the default build mode, the list-sensitive tests again with every
# build in FP32 (SCEMA_MD_NEIGH_EXACT=0: static evaluations then go through the FP32 loop too), then the same-box kernel timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s > gpurun_out/r05_gputests.log 2>&1; rc=$?; tail -3 gpurun_out/r05_gputests.log; [ $rc -eq 0 ] || exit $rc
SCEMA_MD_NEIGH_EXACT=0 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py tests/test_gpu_properties.py -x -q -m gpu -k "not npairs" > gpurun_out/r05_gputests_f32.log 2>&1; tail -3 gpurun_out/r05_gputests_f32.log
tools/neigh_whatif.sh ${1:-r05_nb} "new:" "old:SCEMA_MD_LIB=libscema_md_old.so" "exact:SCEMA_MD_NEIGH_EXACT=1" "new2:"
SCEMA_MD_TIMING=1 python bench.py --sims 72 --steps 1 --warmup 1 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache gpurun_out/equil_pe10k.npz 2>&1 | grep -E "row max|far skin" | tail -2

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
for L in "$@"; do
  echo "== $L"
  env SCEMA_MD_LIB=libscema_md_$L.so SCEMA_MD_TIMING=1 SCEMA_MD_SPLIT=0 python bench.py --sims 576 --steps 1 --warmup 1 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C 2>&1 | grep -E "k_neigh_build" | tail -2
done

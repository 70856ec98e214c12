#!/bin/bash
# wave clocks of k_neigh_build from the -DPAIR_TIMING build of the library (built in the container: make OBJ=_obj_t LIBNAME=libscema_md_t.so EXTRA=-DPAIR_TIMING)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
for mode in "" "SCEMA_MD_NEIGH_EXACT=1"; do
  echo "== mode [$mode]"
  env SCEMA_MD_LIB=libscema_md_t.so SCEMA_MD_TIMING=1 SCEMA_MD_SPLIT=0 $mode python bench.py --sims ${1:-576} --steps 1 --warmup 1 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C 2>&1 | grep -E "k_neigh_build|row max" | tail -3
done

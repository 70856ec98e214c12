#!/bin/bash
# usage (on the GPU box, from the repo root): tools/neigh_timing.sh  -- clock-counter build (-DPAIR_TIMING): where the waves of k_neigh_build and k_pair spend their cycles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
touch scema_amd/csrc/md_types.h
make -C scema_amd/csrc -j16 -s HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics -DPAIR_TIMING" 2>&1 | grep -E "error" | head
SCEMA_MD_TIMING=1 python bench.py --sims ${1:-72} --steps 2 --warmup 1 --no-cpu-baseline --monotonic-updates 0 > gpurun_out/neigh_timing.json.log 2> gpurun_out/neigh_timing.err
grep "clocks" gpurun_out/neigh_timing.err | tail -8

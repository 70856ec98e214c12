#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pair_count.sh  -- counting build of k_pair (-DPAIR_COUNT): where the lanes of the row loop are
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
touch scema_amd/csrc/md_types.h
make -C scema_amd/csrc -j16 -s HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics -DPAIR_COUNT" 2>&1 | grep -E "error" | head
SCEMA_MD_TIMING=1 python bench.py --sims 72 --steps 2 --warmup 1 --no-cpu-baseline --monotonic-updates 0 > gpurun_out/pair_count.json.log 2> gpurun_out/pair_count.err
grep "k_pair lanes\|far skin band\|row entries" gpurun_out/pair_count.err | tail -12

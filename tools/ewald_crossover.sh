#!/bin/bash
# usage (GPU box, repo root): tools/ewald_crossover.sh   -- share of the reciprocal (Ewald) kernels in a step for replicas of 10k, 34k and 83k atoms
# at the reference's accuracy (where a particle-mesh solver would start to pay)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # name cells sims
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/xo_$1 -- python bench.py --cells $2 $3 $4 --sims $5 --steps 1 --warmup 1 --nss 40 --equil-steps 200 --no-cpu-baseline > gpurun_out/xo_$1.log 2>&1
  python tools/kernel_table.py gpurun_out/xo_$1 > gpurun_out/r02_ewald_crossover_$1.txt
  rm -rf gpurun_out/xo_$1
  head -8 gpurun_out/r02_ewald_crossover_$1.txt
}
run 10k 6 9 16 72
run 34k 9 13 24 24
run 83k 12 18 32 8

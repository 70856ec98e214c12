#!/bin/bash
# kernel table + gaps of a single replica (BASELINE config 2) (usage on the GPU box: tools/r05_1_prof.sh <tag>)
T=${1:-r05_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof1 -- python bench.py --sims 1 --steps 10 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > gpurun_out/${T}_prof1_bench.json.log 2>&1
python tools/kernel_table.py gpurun_out/${T}_prof1 > gpurun_out/${T}_kernel_table_bench_1sim.txt
python tools/kernel_gaps.py gpurun_out/${T}_prof1 16 > gpurun_out/${T}_kernel_gaps_bench_1sim.txt
rm -rf gpurun_out/${T}_prof1
head -24 gpurun_out/${T}_kernel_table_bench_1sim.txt; head -14 gpurun_out/${T}_kernel_gaps_bench_1sim.txt

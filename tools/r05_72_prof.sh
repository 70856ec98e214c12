#!/bin/bash
# kernel table + gaps of one GPU's share of the 8-GPU run (72 replicas) (usage on the GPU box: tools/r05_72_prof.sh <tag>)
T=${1:-r05_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof72 -- python bench.py --sims 72 --steps 6 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > gpurun_out/${T}_prof72_bench.json.log 2>&1
python tools/kernel_table.py gpurun_out/${T}_prof72 > gpurun_out/${T}_kernel_table_bench_72sims.txt
python tools/kernel_gaps.py gpurun_out/${T}_prof72 16 > gpurun_out/${T}_kernel_gaps_bench_72sims.txt
rm -rf gpurun_out/${T}_prof72
head -24 gpurun_out/${T}_kernel_table_bench_72sims.txt; head -12 gpurun_out/${T}_kernel_gaps_bench_72sims.txt

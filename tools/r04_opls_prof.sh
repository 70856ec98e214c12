#!/bin/bash
# kernel table of the headline workload (usage on the GPU box: tools/r04_opls_prof.sh <tag>)
T=${1:-r04_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > gpurun_out/${T}_prof_bench.json.log 2>&1
cp gpurun_out/${T}_prof/*/*kernel_stats.csv gpurun_out/${T}_kernel_stats_bench_576sims.csv
python tools/kernel_table.py gpurun_out/${T}_prof > gpurun_out/${T}_kernel_table_bench_576sims.txt
rm -rf gpurun_out/${T}_prof
head -14 gpurun_out/${T}_kernel_table_bench_576sims.txt

#!/bin/bash
# usage (on the GPU box): tools/r03_one.sh <tag> -- kernel table and gaps of a single-replica run (BASELINE config 2), and the PPPM kernels of the 576 batch
T=${1:-r03_c}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 2 --warmup 1 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof1 -- python bench.py --sims 1 --steps 10 --warmup 2 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > gpurun_out/${T}_prof1_bench.json.log 2>&1
python tools/kernel_table.py gpurun_out/${T}_prof1 > gpurun_out/${T}_kernel_table_bench_1sim.txt
python tools/kernel_gaps.py gpurun_out/${T}_prof1 16 > gpurun_out/${T}_kernel_gaps_bench_1sim.txt
rm -rf gpurun_out/${T}_prof1
cat gpurun_out/${T}_kernel_table_bench_1sim.txt | head -34; cat gpurun_out/${T}_kernel_gaps_bench_1sim.txt | head -20
grep "^{" gpurun_out/${T}_prof1_bench.json.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -- python bench.py --steps 3 --warmup 3 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > gpurun_out/${T}_prof_bench.json.log 2>&1
python tools/kernel_table.py gpurun_out/${T}_prof > gpurun_out/${T}_kernel_table_bench_576sims.txt
rm -rf gpurun_out/${T}_prof
head -16 gpurun_out/${T}_kernel_table_bench_576sims.txt

#!/bin/bash
# kernel table + idle gaps of a small batch as it runs (usage on the GPU box: tools/small_prof.sh <tag> <replicas> [ENV=.. ...])
T=${1:-r06_x}; N=${2:-1}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
[ -f $C ] || python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
[ $# -gt 0 ] && export "$@"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof$N -- python bench.py --sims $N --steps 10 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > gpurun_out/${T}_prof${N}_bench.json.log 2>&1
python tools/kernel_table.py gpurun_out/${T}_prof$N > gpurun_out/${T}_kernel_table_bench_${N}sims.txt
python tools/kernel_gaps.py gpurun_out/${T}_prof$N 16 > gpurun_out/${T}_kernel_gaps_bench_${N}sims.txt
python tools/timeline.py gpurun_out/${T}_prof$N > gpurun_out/${T}_timeline_${N}sims.txt
rm -rf gpurun_out/${T}_prof$N
head -24 gpurun_out/${T}_kernel_table_bench_${N}sims.txt; head -14 gpurun_out/${T}_kernel_gaps_bench_${N}sims.txt

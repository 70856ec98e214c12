#!/bin/bash
# kernel table of the headline workload with the batch whole (usage: tools/r05_whole_prof.sh <tag> [lib])
T=${1:-r05_x}; L=${2:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
[ -n "$L" ] && export SCEMA_MD_LIB=$L
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
SCEMA_MD_SPLIT=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_wprof -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > gpurun_out/${T}_wprof_bench.json.log 2>&1
python tools/kernel_table.py gpurun_out/${T}_wprof > gpurun_out/${T}_kernel_table_bench_576sims_whole.txt
rm -rf gpurun_out/${T}_wprof
head -12 gpurun_out/${T}_kernel_table_bench_576sims_whole.txt | cut -c1-150

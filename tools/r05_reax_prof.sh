#!/bin/bash
# kernel table of the ReaxFF replica set (usage on the GPU box: tools/r05_reax_prof.sh <tag>)
T=${1:-r05_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_rprof -- python bench.py --force-field reax --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_rprof_bench.json.log 2>&1
cp gpurun_out/${T}_rprof/*/*kernel_stats.csv gpurun_out/${T}_kernel_stats_bench_reax_72sims.csv
python tools/kernel_table.py gpurun_out/${T}_rprof > gpurun_out/${T}_kernel_table_bench_reax_72sims.txt 2>/dev/null || python tools/kernel_median.py gpurun_out/${T}_rprof > gpurun_out/${T}_kernel_table_bench_reax_72sims.txt
python tools/kernel_gaps.py gpurun_out/${T}_rprof 10 k_rx_hrow > gpurun_out/${T}_kernel_gaps_bench_reax_72sims.txt
rm -rf gpurun_out/${T}_rprof
head -12 gpurun_out/${T}_kernel_table_bench_reax_72sims.txt; head -11 gpurun_out/${T}_kernel_gaps_bench_reax_72sims.txt
python bench.py --force-field reax --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('reax', round(d['value'],1), 'evals/s')"

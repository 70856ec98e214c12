#!/bin/bash
# usage (on the GPU box, from the repo root): tools/r03_reax.sh <tag>  -- BASELINE config 5: the ReaxFF replica set with its own roofline block and CPU baseline, kernel table
T=${1:-r03_b}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --force-field reax --steps 6 --warmup 2 > gpurun_out/${T}_bench_reax_72sims.json.log 2> gpurun_out/${T}_reax.err
grep "^{" gpurun_out/${T}_bench_reax_72sims.json.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_rprof -- python bench.py --force-field reax --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_rprof_bench.json.log 2>&1
cp gpurun_out/${T}_rprof/*/*kernel_stats.csv gpurun_out/${T}_kernel_stats_bench_reax_72sims.csv
python tools/kernel_table.py gpurun_out/${T}_rprof > gpurun_out/${T}_kernel_table_bench_reax_72sims.txt
rm -rf gpurun_out/${T}_rprof
head -24 gpurun_out/${T}_kernel_table_bench_reax_72sims.txt

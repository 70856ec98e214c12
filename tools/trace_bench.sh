#!/bin/bash
# usage (GPU box, repo root): tools/trace_bench.sh <tag> [extra bench args] -- rocprofv3 kernel trace of the default bench on a cached equilibrated state
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=gpurun_out/equil_pe10k.npz
[ -f $C ] || python bench.py --sims 1 --steps 1 --warmup 0 --nss 10 --no-cpu-baseline --equil-cache $C > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_$TAG -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --equil-cache $C "$@" > gpurun_out/trace_$TAG.log 2>&1
tail -1 gpurun_out/trace_$TAG.log | cut -c1-300
python tools/kernel_table.py gpurun_out/trace_$TAG | tee gpurun_out/trace_$TAG.table.txt
cp gpurun_out/trace_$TAG/*/*kernel_stats.csv gpurun_out/trace_${TAG}_kernel_stats.csv
rm -rf gpurun_out/trace_$TAG

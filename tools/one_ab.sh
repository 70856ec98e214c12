#!/bin/bash
# usage (on the GPU box): tools/one_ab.sh  -- single replica and the headline batch under the profiler: launches and kernel time per step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/one_prof -- python bench.py --sims 1 --steps 6 --warmup 2 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > gpurun_out/one.json.log 2>&1
echo "== 1 replica: $(grep '^{' gpurun_out/one.json.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), "evals/s", round(d["ms_per_step"],3), "ms per update")')"
python tools/kernel_table.py gpurun_out/one_prof | head -24; rm -rf gpurun_out/one_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nab_prof -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > gpurun_out/nab.json.log 2>&1
echo "== 576 replicas: $(grep '^{' gpurun_out/nab.json.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["avg_launch_ms"],3))')"
python tools/kernel_table.py gpurun_out/nab_prof | head -24; rm -rf gpurun_out/nab_prof

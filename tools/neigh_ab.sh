#!/bin/bash
# usage (on the GPU box): tools/neigh_ab.sh  -- parity tests of the list build, then a sustained profile of the headline bench (k_neigh_build / k_pair per step), then the wave clocks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py -x -q 2>&1 | tail -3 || exit 1
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nab_prof -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > gpurun_out/nab.json.log 2>&1
echo "== $(grep '^{' gpurun_out/nab.json.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["avg_launch_ms"],3))')"
python tools/kernel_table.py gpurun_out/nab_prof | grep -E "k_neigh_build|k_pair<true|kernel time|k_cell_sort|k_pack"; rm -rf gpurun_out/nab_prof
bash tools/neigh_timing.sh 576 2>&1 | grep neigh_build | tail -2

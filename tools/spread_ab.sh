#!/bin/bash
# usage (on the GPU box): tools/spread_ab.sh -- k_neigh_build with its tiles dealt over all XCDs (1) or pinned like k_pair's (0), sustained, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > /dev/null 2>&1
for V in 1 0 1 0; do
  export SCEMA_MD_NEIGH_SPREAD=$V
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sab_prof -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > gpurun_out/spr_$V.json.log 2>&1
  echo "== spread $V: $(grep '^{' gpurun_out/spr_$V.json.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["avg_launch_ms"],3))')"
  python tools/kernel_table.py gpurun_out/sab_prof | grep -E "k_neigh_build|k_pair<true|kernel time"; rm -rf gpurun_out/sab_prof
done

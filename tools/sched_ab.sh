#!/bin/bash
# usage (on the GPU box): tools/sched_ab.sh -- k_pair / k_neigh_build with the three row schedules (1 longest first + splits, 0 longest first, 2 round robin), sustained
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > /dev/null 2>&1
for V in 1 2 0 1 2; do
  export SCEMA_MD_ROW_SPLIT=$V
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sab_prof -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > gpurun_out/sab_$V.json.log 2>&1
  echo "== schedule $V: $(grep '^{' gpurun_out/sab_$V.json.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["avg_launch_ms"],3))')"
  python tools/kernel_table.py gpurun_out/sab_prof | grep -E "k_neigh_build|k_pair<true|kernel time"; rm -rf gpurun_out/sab_prof
done

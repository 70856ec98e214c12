#!/bin/bash
# usage (on the GPU box): tools/skin_scan.sh  -- the list skin as a performance knob (SCEMA_MD_SKIN_EXTRA, added to the reference's 2 A): sustained headline bench per value
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > /dev/null 2>&1
for X in ${@:-0 -0.3 0.3 -0.15 0}; do
  SCEMA_MD_SKIN_EXTRA=$X python bench.py --steps 6 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('extra $X:', round(d['value'],1), 'evals/s, pair launch', round(d['roofline']['avg_launch_ms'],3), 'ms, steps per rebuild', round(c['steps_per_list_rebuild'],1), 'skin', c['list_skin_A'])"
done

#!/bin/bash
# same-box A/B of the ReaxFF part batches (usage on the GPU box: tools/reax_parts_ab.sh "9 18 36 72" "1 2 3 4"), two rounds
SIZES=$1; PARTS=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for R in 1 2; do
for N in $SIZES; do
  for H in $PARTS; do
    SCEMA_REAX_HALVES=$H python bench.py --force-field reax --sims $N --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$N', 'parts$H', round(d['value'],1), 'evals/s', round(d['ms_per_step'],2), 'ms per update', flush=True)"
  done
done
done

#!/bin/bash
# round-6 one-off A/Bs on one box: CU masks for the two half batches, ragged update as several launch groups, host time per update at 72
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
[ -f $C ] || python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
run() { name=$1; envs=$2; shift; shift
  env $envs python bench.py "$@" --no-cpu-baseline --monotonic-updates 0 --share8-updates 0 --reax-leg off --equil-cache $C 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), 'evals/s', round(d['ms_per_step'],2), 'ms per update', round(d['replica_steps_per_s']), 'replica-steps/s', d['config']['md_steps_per_eval'], d['config']['env_overrides'], flush=True)"; }
for R in 1 2; do
  run "72 nomask" "A=1" --sims 72 --steps 10 --warmup 3
  run "72 cumask2" "SCEMA_MD_CU_MASK=2" --sims 72 --steps 10 --warmup 3
  run "576 nomask" "A=1" --sims 576 --steps 4 --warmup 2
  run "576 cumask2" "SCEMA_MD_CU_MASK=2" --sims 576 --steps 4 --warmup 2
  run "576 ragged G1" "A=1" --sims 576 --strain-set imbalanced --steps 4 --warmup 2
  run "576 ragged G2" "SCEMA_MD_RAGGED_GROUPS=2" --sims 576 --strain-set imbalanced --steps 4 --warmup 2
  run "576 ragged G3" "SCEMA_MD_RAGGED_GROUPS=3" --sims 576 --strain-set imbalanced --steps 4 --warmup 2
  run "576 ragged G4" "SCEMA_MD_RAGGED_GROUPS=4" --sims 576 --strain-set imbalanced --steps 4 --warmup 2
done
SCEMA_MD_TIMING=1 python bench.py --sims 72 --steps 3 --warmup 2 --no-cpu-baseline --monotonic-updates 0 --share8-updates 0 --reax-leg off --equil-cache $C 2>&1 | grep -E "chunk of|host:" | tail -8

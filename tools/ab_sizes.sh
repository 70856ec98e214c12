#!/bin/bash
# same-box A/B over batch sizes (usage on the GPU box: tools/ab_sizes.sh "1 9 72" "name:ENV=.. ENV2=.." "name2:" ...), two rounds
SIZES=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
[ -f $C ] || python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
for R in 1 2; do
for N in $SIZES; do
  S=$(( N >= 576 ? 5 : (N >= 144 ? 8 : (N >= 36 ? 12 : 30)) ))
  for V in "$@"; do
    name=${V%%:*}; envs=${V#*:}
    env $envs python bench.py --sims $N --steps $S --warmup 3 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$N', '$name', round(d['value'],1), 'evals/s', round(d['ms_per_step'],2), 'ms per update; k_pair whole us/replica', round(1e3*r['whole_avg_launch_ms']/max(r['whole_sims_per_launch'],1),2), d['config']['env_overrides'], 'chk', d['config']['stress_zz_checksum_Pa'], flush=True)"
  done
done
done

#!/bin/bash
# headline workload, same-box A/B of environment variants (usage: tools/r05_env_ab.sh "name:ENV=.." ...)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
for V in "$@"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --steps 6 --warmup 3 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), 'evals/s', round(d['ms_per_step'],1), 'ms per update', d['config']['env_overrides'], 'chk', d['config']['stress_zz_checksum_Pa'], flush=True)"
done

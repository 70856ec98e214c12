#!/bin/bash
# one replica (BASELINE config 2): same-box A/B of environment variants (usage: tools/r05_1_ab.sh "name:ENV=.." ...)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
for V in "$@"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --sims 1 --steps 20 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3), 'ms per evaluation', d['config']['env_overrides'], 'chk', d['config']['stress_zz_checksum_Pa'], flush=True)"
done

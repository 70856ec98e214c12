#!/bin/bash
T=r04_zs
for V in "a:" "b:SCEMA_MD_SPLIT_MAX=100000" "a2:" "b2:SCEMA_MD_SPLIT_MAX=100000"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --steps 6 --warmup 3 --no-cpu-baseline --reax-leg off --monotonic-updates 0 --equil-cache gpurun_out/equil_pe10k.npz > gpurun_out/${T}_$name.json.log 2> gpurun_out/${T}_$name.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_$name.json.log').read().strip().split('\n')[-1]); r=d['roofline']
print('$name', round(d['value'],1), 'evals/s; k_pair ms', round(r['avg_launch_ms'],3), d['config']['env_overrides'], flush=True)
PY
done

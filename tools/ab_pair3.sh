#!/bin/bash
# same-box A/B/C of three builds of the library on the headline workload (usage: tools/ab_pair3.sh <tag> [steps])
T=${1:-ab}; N=${2:-6}
for V in "a:" "b:SCEMA_MD_LIB=libscema_md_b.so" "c:SCEMA_MD_LIB=libscema_md_c.so" "a2:" "b2:SCEMA_MD_LIB=libscema_md_b.so" "c2:SCEMA_MD_LIB=libscema_md_c.so"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --steps $N --warmup 3 --no-cpu-baseline --reax-leg off --monotonic-updates 0 --equil-cache gpurun_out/equil_pe10k.npz > gpurun_out/${T}_$name.json.log 2> gpurun_out/${T}_$name.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_$name.json.log').read().strip().split('\n')[-1]); r=d['roofline']
print('$name', round(d['value'],1), 'evals/s; k_pair ms', round(r['avg_launch_ms'],3), d['config']['env_overrides'], flush=True)
PY
done

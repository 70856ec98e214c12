#!/bin/bash
# same-box A/B of one environment switch on the ReaxFF replica set (usage: tools/ab_env.sh <tag> VAR=value [steps])
T=${1:-abe}; SW=$2; N=${3:-6}
for V in "a:" "b:$SW" "a2:" "b2:$SW"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --force-field reax --steps $N --warmup 2 --no-cpu-baseline > gpurun_out/${T}_reax_$name.json.log 2> gpurun_out/${T}_reax_$name.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_reax_$name.json.log').read().strip().split('\n')[-1]); r=d['roofline']
print('$name', round(d['value'],1), 'evals/s; sweep avg ms', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), 'its/solve', round(r['qeq_iterations_per_solve'],2), d['config']['env_overrides'], flush=True)
PY
done

#!/bin/bash
# the ReaxFF set at 576 replicas: same-box A/B of environment variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in "$@"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --force-field reax --sims 576 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$name', round(d['value'],1), 'evals/s; sweep whole ms', round(r['whole_avg_launch_ms'],4), d['config'].get('env_overrides'), flush=True)"
done

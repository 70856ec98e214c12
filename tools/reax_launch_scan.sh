#!/bin/bash
for L in ${@:-0 12 13 14 15 16 18}; do
  if [ $L = 0 ]; then envs=""; else envs="SCEMA_REAX_QEQ_LAUNCH=$L"; fi
  env $envs python bench.py --force-field reax --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('launch=$L', round(d['value'],1), 'evals/s; sweep launches', r['launches'], 'its/solve', round(r['qeq_iterations_per_solve'],2), d['config'].get('env_overrides'), flush=True)"
done

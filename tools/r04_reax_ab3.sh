#!/bin/bash
# same-box A/B: positions staged in LDS for the matrix build of the charge equilibration (k_rx_hrow)
for V in "xlds:" "cache:SCEMA_REAX_HROW_XLDS=0" "xlds_again:" "cache_again:SCEMA_REAX_HROW_XLDS=0"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --force-field reax --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04_g_reax_$name.json.log 2> gpurun_out/r04_g_reax_$name.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04_g_reax_$name.json.log').read().strip().split('\n')[-1]); r=d['roofline']
print('$name', round(d['value'],1), 'evals/s; ms/update', round(d['ms_per_step'],1), d['config']['env_overrides'], flush=True)
PY
done

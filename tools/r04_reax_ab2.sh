#!/bin/bash
# same-box A/B: rows per workgroup of the charge-equilibration sweep (RX_SWR 32 = libscema_md.so, 64 = libscema_md_b.so)
for V in "swr32:" "swr64:SCEMA_MD_LIB=libscema_md_b.so" "swr32_again:" "swr64_again:SCEMA_MD_LIB=libscema_md_b.so"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --force-field reax --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04_e_reax_$name.json.log 2> gpurun_out/r04_e_reax_$name.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04_e_reax_$name.json.log').read().strip().split('\n')[-1]); r=d['roofline']
print('$name', round(d['value'],1), 'evals/s; sweep avg ms', round(r['avg_launch_ms'],4), 'launches', r['launches'], 'frac', round(r['frac'],3), 'its/solve', round(r['qeq_iterations_per_solve'],2), 'sweep share', round(r['rank0_sweep_share_of_wall'],3), d['config']['env_overrides'], flush=True)
PY
done

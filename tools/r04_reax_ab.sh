#!/bin/bash
# same-box A/B of the ReaxFF charge-equilibration sweep variants (round 4): LDS-staged gather vector, single-precision phase
mkdir -p gpurun_out
for V in "zlds_f64:SCEMA_REAX_QEQ_F32=0" "zlds_f32:SCEMA_REAX_QEQ_F32=1" "cache_f64:SCEMA_REAX_QEQ_F32=0 SCEMA_REAX_QEQ_ZLDS=0" "zlds_f64_again:SCEMA_REAX_QEQ_F32=0"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --force-field reax --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r04_d_reax_$name.json.log 2> gpurun_out/r04_d_reax_$name.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04_d_reax_$name.json.log').read().strip().split('\n')[-1]); r=d['roofline']
print('$name', round(d['value'],1), 'evals/s; sweep avg ms', round(r['avg_launch_ms'],4), 'launches', r['launches'], 'frac', round(r['frac'],3), 'its/solve', round(r['qeq_iterations_per_solve'],2), 'sweep share', round(r['rank0_sweep_share_of_wall'],3), flush=True)
PY
done

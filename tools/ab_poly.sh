#!/bin/bash
# same-box sensitivity: fit target of the real-space Ewald polynomial (number of Horner terms of the coulomb block)
for V in "tol2e-13:" "tol1e-9:SCEMA_MD_POLY_TOL=1e-9" "tol1e-6:SCEMA_MD_POLY_TOL=1e-6" "tol2e-13_again:" "tol1e-9_again:SCEMA_MD_POLY_TOL=1e-9"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --steps 5 --warmup 3 --no-cpu-baseline --reax-leg off --monotonic-updates 0 --equil-cache gpurun_out/equil_pe10k.npz > gpurun_out/r04_n_$name.json.log 2> gpurun_out/r04_n_$name.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04_n_$name.json.log').read().strip().split('\n')[-1]); r=d['roofline']
print('$name', round(d['value'],1), 'evals/s; k_pair ms', round(r['avg_launch_ms'],3), d['config']['env_overrides'], flush=True)
PY
done

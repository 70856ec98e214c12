#!/bin/bash
# the driver's command once more on whatever box this call landed on (usage: tools/driver_repeat.sh <tag>): value, k_pair launch time, ReaxFF leg
T=${1:-rep}
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_576sims_driver.json.log 2> gpurun_out/${T}.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/${T}_bench_576sims_driver.json.log') if l.startswith('{')][-1]); r=d['roofline']; x=d['config']['reax']
print('${T}', round(d['value'],1), 'evals/s; k_pair ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],3), '; reax', round(x['evals_per_s'],1), 'sweep frac', round(x['roofline']['frac'],3), '; cpu', round(d['cpu_baseline']['value'],2), d['config']['env_overrides'])
PY

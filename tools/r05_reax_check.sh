#!/bin/bash
# ReaxFF GPU tests, then same-box bench of the ReaxFF replica set with environment variants (usage: tools/r05_reax_check.sh <tag> "name:ENV=.." ...)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=$1; shift
timeout -k 10 900 python -m pytest tests/test_gpu_reax.py tests/test_gpu_reax_eval.py tests/test_gpu_reax_equil.py -x -q -m gpu > gpurun_out/${T}_reaxtests.log 2>&1; rc=$?; tail -3 gpurun_out/${T}_reaxtests.log; [ $rc -eq 0 ] || { grep -E "^E |Error" gpurun_out/${T}_reaxtests.log | head -20; exit $rc; }
for V in "$@"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --force-field reax --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_$name.json.log 2> gpurun_out/${T}_$name.err || { tail -5 gpurun_out/${T}_$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/${T}_$name.json.log').read().strip().split('\n')[-1]); r=d['roofline']
print('$name', round(d['value'],1), 'evals/s; sweep whole ms', round(r['whole_avg_launch_ms'],4), 'frac', round(r['frac'],3), 'timed launches', r['timed_launches'], 'its/solve', round(r['qeq_iterations_per_solve'],2), d['config']['env_overrides'], flush=True)
PY
done

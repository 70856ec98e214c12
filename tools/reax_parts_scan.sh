#!/bin/bash
# part batches of a ReaxFF run (SCEMA_REAX_HALVES = 1, 2, 3, 4, 6) on the 72-replica set and on 576 replicas
for L in 1 2 3 4 6; do
  SCEMA_REAX_HALVES=$L python bench.py --force-field reax --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('72 replicas, parts=$L', round(d['value'],1), 'evals/s; sweep as run', round(r['avg_launch_ms'],4), 'ms', round(r['frac'],3), '; alone', round(r['frac'],3), flush=True)"
done
for L in 2 4; do
  SCEMA_REAX_HALVES=$L python bench.py --force-field reax --sims 576 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('576 replicas, parts=$L', round(d['value'],1), 'evals/s; sweep as run', round(r['avg_launch_ms'],4), 'ms', round(r['frac'],3), '; alone', round(r['frac'],3), flush=True)"
done

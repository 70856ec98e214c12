#!/bin/bash
# usage (GPU box, repo root): tools/pmc_traffic.sh <round-tag>
# HBM traffic of the pair kernel from PMC counters, per MI355X_MICROARCH.md "HBM": separate --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), no tracing domains; gfx950 correction: FETCH_SIZE counts
# 128-byte requests as 64 bytes for coalesced streams -> x2 (checked against the known byte count of the
# one-row-per-atom kernel, whose rows are read exactly once: 2*FETCH_SIZE = 4.49 GB = algorithmic bytes).
TAG=${1:-r01}; SIMS=72
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-include-regex "k_pair" --output-format csv -d gpurun_out/traffic_${TAG}_$C -- python bench.py --sims $SIMS --steps 1 --warmup 0 --nss 10 --no-cpu-baseline > gpurun_out/traffic_${TAG}_$C.log 2>&1
done
python - <<PY
import csv, glob, json
def mean(counter, pat):
    f = glob.glob(f'gpurun_out/traffic_${TAG}_{counter}/*/*_counter_collection.csv')[0]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter and pat in r['Kernel_Name']]
    return sum(v) / len(v), len(v)
fetch, n = mean('FETCH_SIZE', 'k_pair<true')
write, _ = mean('WRITE_SIZE', 'k_pair<true')
out = {"kernel": "k_pair<true,false,*>", "sims_per_launch": $SIMS, "dispatches": n,
       "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
       "hbm_bytes_per_launch_corrected": (2.0 * fetch + write) * 1024.0,
       "hbm_bytes_per_sim_step_corrected": (2.0 * fetch + write) * 1024.0 / $SIMS,
       "correction": "gfx950: FETCH_SIZE x2 for coalesced streams (MI355X_MICROARCH.md, HBM); WRITE_SIZE as read",
       "command": "rocprofv3 --pmc <C> --kernel-include-regex k_pair -- python bench.py --sims 72 --steps 1 --warmup 0 --nss 10 --no-cpu-baseline"}
json.dump(out, open('gpurun_out/pair_traffic_${TAG}.json', 'w'), indent=1)
print(json.dumps(out))
PY

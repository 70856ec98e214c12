#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_reax.sh
# HBM traffic and instruction counts of the ReaxFF charge-equilibration sweep (k_rx_qeq_sweep, 48 % of a reax step): separate
# rocprofv3 --pmc passes (no tracing domains) on tools/reax_bench.py; launches of converged replicas (which leave at once) are
# told from full sweeps by their fetch size.  Writes gpurun_out/reax_pmc.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SCEMA_REAX_HALVES=1 SCEMA_REAX_OVERLAP=0   # whole batch per launch, one stream: the counters are per launch over all replicas
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD"; do
  t=$(echo $P | cut -d" " -f1)
  timeout 400 rocprofv3 --pmc $P --kernel-include-regex "k_rx_qeq_sweep" --output-format csv -d gpurun_out/pmcrx_$t -- python tools/reax_bench.py --updates 1 --warmup 0 --equil-steps 0 > gpurun_out/pmcrx_$t.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
per = collections.defaultdict(dict)   # dispatch -> counter -> value
for d in sorted(glob.glob('gpurun_out/pmcrx_*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        per[(d.split('/')[1], r['Dispatch_Id'])][r['Counter_Name']] = per[(d.split('/')[1], r['Dispatch_Id'])].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
def full(counter, frac=0.5):
    vals = [v[counter] for v in per.values() if counter in v]
    top = max(vals)
    sel = [x for x in vals if x > frac * top]
    return sum(sel) / len(sel), len(sel), len(vals)
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_VMEM_RD"):
    try:
        m, n, tot = full(c)
        out[c] = m; out["launches_" + c] = n; out["all_launches_" + c] = tot
    except ValueError:
        pass
if "FETCH_SIZE" in out:
    out["hbm_bytes_per_sweep_corrected"] = (2.0 * out["FETCH_SIZE"] + out.get("WRITE_SIZE", 0.0)) * 1024.0
    out["correction"] = "gfx950: FETCH_SIZE x2 for coalesced streams (MI355X_MICROARCH.md, HBM); WRITE_SIZE as read; per launch that sweeps all 72 replicas"
out["command"] = "rocprofv3 --pmc <group> --kernel-include-regex k_rx_qeq_sweep -- python tools/reax_bench.py --updates 1 --warmup 0 --equil-steps 0"
json.dump(out, open('gpurun_out/reax_pmc.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmcrx_*/

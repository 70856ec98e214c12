#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_rx_kernels.sh [regex]
# Issue / wait / memory-instruction counters of the per-step ReaxFF kernels other than the sweep (default: k_rx_hrow, k_rx_nonbonded_once,
# k_rx_bonds, k_rx_torsions, k_rx_angles): separate rocprofv3 --pmc passes (no tracing domains) on tools/reax_bench.py, averaged per launch.
# Writes gpurun_out/reax_kernels_pmc.json.
RE=${1:-"k_rx_hrow|k_rx_nonbonded_once|k_rx_bonds|k_rx_torsions|k_rx_angles"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SCEMA_REAX_HALVES=1 SCEMA_REAX_OVERLAP=0   # whole batch per launch, one stream: the counters are per launch over all replicas
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_WAVES_LT_64" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  t=$(echo $P | cut -d" " -f1)
  timeout 400 rocprofv3 --pmc $P --kernel-include-regex "$RE" --output-format csv -d gpurun_out/pmcrk_$t -- python tools/reax_bench.py --updates 1 --warmup 0 --equil-steps 0 > gpurun_out/pmcrk_$t.log 2>&1 || echo "pass $t failed"
done
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for d in sorted(glob.glob('gpurun_out/pmcrk_*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        k = r['Kernel_Name'].split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
out = {k: {c: v / len(cnt[k][c]) for c, v in cs.items()} for k, cs in acc.items()}
for k in out: out[k]['launches'] = max(len(s) for s in cnt[k].values())
json.dump(out, open('gpurun_out/reax_kernels_pmc.json', 'w'), indent=1)
for k, v in out.items():
    print(k, {c: round(x) for c, x in v.items()})
PY
rm -rf gpurun_out/pmcrk_*/

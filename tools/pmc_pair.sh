#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_pair.sh <tag> [kernel-regex]
# runs separate rocprofv3 --pmc passes (no tracing domains) on a short bench and prints per-dispatch means
TAG=${1:-x}; RE=${2:-k_pair}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE"; do
  t=$(echo $P | cut -d" " -f1)
  timeout 300 rocprofv3 --pmc $P --kernel-include-regex "$RE" --output-format csv -d gpurun_out/pmc_${TAG}_$t -- python bench.py --sims 72 --steps 1 --warmup 0 --nss 10 --no-cpu-baseline > gpurun_out/pmc_${TAG}_$t.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_${TAG}_*/*/*_counter_collection.csv')):
    rows=list(csv.DictReader(open(d)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
    for r in rows:
        k=r['Kernel_Name'][:34]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
    for k,v in agg.items():
        for c,val in v.items(): print(f"{k:34s} {c:32s} {val/len(disp[k]):16.0f}")
PY

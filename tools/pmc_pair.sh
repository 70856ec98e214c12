#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_pair.sh <tag> [kernel-regex] [sims]
# Separate rocprofv3 --pmc passes (no tracing domains) on a short bench (equilibrated replica, one update of 10+10 steps);
# prints per-dispatch means of the batch launches only (the single-replica launches of the equilibration run have a
# smaller grid and are filtered out) and writes gpurun_out/pmc_<tag>.json.
TAG=${1:-x}; RE=${2:-k_pair}; SIMS=${3:-72}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CACHE=gpurun_out/equil_pe10k.npz
# the equilibration (2 000 single-replica steps) runs once, outside the profiler
[ -f $CACHE ] || python bench.py --sims 1 --steps 1 --warmup 0 --nss 10 --no-cpu-baseline --equil-cache $CACHE > /dev/null 2>&1
export SCEMA_MD_SPLIT=0   # whole batch per launch: the counters are divided by the batch size
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES_LT_64" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  t=$(echo $P | cut -d" " -f1)
  timeout 400 rocprofv3 --pmc $P --kernel-include-regex "$RE" --output-format csv -d gpurun_out/pmc_${TAG}_$t -- python bench.py --sims $SIMS --steps 1 --warmup 0 --nss 10 --no-cpu-baseline --equil-cache $CACHE > gpurun_out/pmc_${TAG}_$t.log 2>&1
done
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
from bench import kernel_source_hashes
out = collections.defaultdict(dict)
for d in sorted(glob.glob('gpurun_out/pmc_${TAG}_*/*/*_counter_collection.csv')):
    rows = list(csv.DictReader(open(d)))
    gmax = collections.defaultdict(int)
    for r in rows:
        gmax[r['Kernel_Name']] = max(gmax[r['Kernel_Name']], int(r['Grid_Size']))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in rows:
        if int(r['Grid_Size']) != gmax[r['Kernel_Name']]: continue
        k = r['Kernel_Name'].split('(')[0]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
    for k, v in agg.items():
        for c, val in v.items():
            out[k][c] = val / len(disp[k]); out[k]['dispatches_' + c] = len(disp[k])
            print(f"{k:40s} {c:32s} {val/len(disp[k]):16.0f}  ({len(disp[k])} dispatches)")
kp = next((k for k in out if k.startswith('void k_pair<true, false')), None)
if kp and 'FETCH_SIZE' in out[kp] and 'WRITE_SIZE' in out[kp] and 'SQ_INSTS_VALU' in out[kp]:
    o = out[kp]
    hbm = (2.0 * o['FETCH_SIZE'] + o['WRITE_SIZE']) * 1024.0
    json.dump({"kernel": kp, "sims_per_launch": $SIMS, "dispatches": o['dispatches_SQ_INSTS_VALU'],
               "FETCH_SIZE_KB_per_launch": o['FETCH_SIZE'], "WRITE_SIZE_KB_per_launch": o['WRITE_SIZE'],
               "hbm_bytes_per_launch_corrected": hbm, "hbm_bytes_per_sim_step_corrected": hbm / $SIMS,
               "valu_insts_per_launch": o['SQ_INSTS_VALU'], "valu_insts_per_sim_step": o['SQ_INSTS_VALU'] / $SIMS,
               "cycles_per_valu_inst": 4.0,
               "kernel_sources": kernel_source_hashes(),   # git blob hashes of the kernel's sources: bench.py drops these counters when the tree's differ
               "SQ_WAVE_CYCLES": o.get('SQ_WAVE_CYCLES'), "SQ_ACTIVE_INST_VALU": o.get('SQ_ACTIVE_INST_VALU'), "SQ_WAIT_ANY": o.get('SQ_WAIT_ANY'),
               "GRBM_GUI_ACTIVE": o.get('GRBM_GUI_ACTIVE'),
               "correction": "gfx950: FETCH_SIZE x2 for coalesced streams (MI355X_MICROARCH.md, HBM); WRITE_SIZE as read",
               "source": "profiles/pair_pmc.json: tools/pmc_pair.sh, rocprofv3 --pmc passes (one counter group each, no tracing) on bench.py --sims $SIMS --steps 1 --warmup 0 --nss 10, equilibrated replica; batch launches only"},
              open('gpurun_out/pair_pmc_${TAG}.json', 'w'), indent=1)
json.dump({"sims_per_launch": $SIMS, "command": "rocprofv3 --pmc <group> --kernel-include-regex $RE -- python bench.py --sims $SIMS --steps 1 --warmup 0 --nss 10 --no-cpu-baseline --equil-cache <state after 2000 NVT+SHAKE steps>", "kernels": out}, open('gpurun_out/pmc_${TAG}.json', 'w'), indent=1)
PY

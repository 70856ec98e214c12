#!/bin/bash
# 72 replicas (one GPU's share of 576 on 8 GPUs): PPPM next to the Ewald sum, kernel tables and idle gaps (on the GPU box)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02_l}; N=${2:-72}
C=gpurun_out/equil_pe10k.npz
for ks in pppm ewald; do
  python bench.py --sims $N --steps 8 --warmup 2 --no-cpu-baseline --equil-cache $C --kspace $ks 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$ks', $N, round(d['value'],1), round(d['ms_per_step'],1))" || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp_prof -- python bench.py --sims $N --steps 4 --warmup 1 --no-cpu-baseline --equil-cache $C --kspace $ks > gpurun_out/pp_prof.log 2>&1 || exit 1
  python tools/kernel_table.py gpurun_out/pp_prof > gpurun_out/${TAG}_kernel_table_bench_${N}sims_${ks}.txt
  python tools/kernel_gaps.py gpurun_out/pp_prof 8 > gpurun_out/${TAG}_kernel_gaps_${N}sims_${ks}.txt; cat gpurun_out/${TAG}_kernel_gaps_${N}sims_${ks}.txt
  rm -rf gpurun_out/pp_prof
  head -22 gpurun_out/${TAG}_kernel_table_bench_${N}sims_${ks}.txt
done

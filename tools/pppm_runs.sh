#!/bin/bash
# PPPM at full size next to the Ewald sum: bench lines and kernel tables (run on the GPU box; output under gpurun_out/)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02_k}
C=gpurun_out/equil_pe10k.npz
for ks in pppm ewald; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --equil-cache $C --kspace $ks 2>gpurun_out/${ks}_bench.err > gpurun_out/${TAG}_bench_576sims_${ks}.json || exit 1
  python -c "
import sys,json
d=json.loads(open('gpurun_out/${TAG}_bench_576sims_${ks}.json').read()); print('$ks', round(d['value'],1), round(d['ms_per_step'],1), d['config']['stress_zz_checksum_Pa'])"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp_prof -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --equil-cache $C --kspace $ks > gpurun_out/pp_prof.log 2>&1 || exit 1
  python tools/kernel_table.py gpurun_out/pp_prof > gpurun_out/${TAG}_kernel_table_bench_576sims_${ks}.txt
  python tools/kernel_gaps.py gpurun_out/pp_prof > gpurun_out/${TAG}_kernel_gaps_576sims_${ks}.txt; cat gpurun_out/${TAG}_kernel_gaps_576sims_${ks}.txt
  rm -rf gpurun_out/pp_prof
  head -12 gpurun_out/${TAG}_kernel_table_bench_576sims_${ks}.txt
done

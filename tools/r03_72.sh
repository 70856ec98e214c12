#!/bin/bash
# usage (on the GPU box): tools/r03_72.sh <tag> -- the 72-replica batch (one GPU's share of 576 on 8): kernel table, with and without the two-half pipeline
T=${1:-r03_d}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > /dev/null 2>&1
for V in split nosplit; do
  if [ $V = nosplit ]; then export SCEMA_MD_SPLIT=0; else unset SCEMA_MD_SPLIT; fi
  python bench.py --sims 72 --steps 10 --warmup 3 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C 2>/dev/null | grep "^{" > gpurun_out/${T}_bench_72sims_$V.json.log
  python -c "import json; d=json.loads(open('gpurun_out/${T}_bench_72sims_$V.json.log').read()); print('$V', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['avg_launch_ms'], d['roofline']['sims_per_launch'])"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof72 -- python bench.py --sims 72 --steps 6 --warmup 3 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C > /dev/null 2>&1
  python tools/kernel_table.py gpurun_out/${T}_prof72 > gpurun_out/${T}_kernel_table_bench_72sims_$V.txt
  python tools/kernel_gaps.py gpurun_out/${T}_prof72 10 > gpurun_out/${T}_kernel_gaps_bench_72sims_$V.txt
  rm -rf gpurun_out/${T}_prof72
  head -16 gpurun_out/${T}_kernel_table_bench_72sims_$V.txt; head -6 gpurun_out/${T}_kernel_gaps_bench_72sims_$V.txt
done

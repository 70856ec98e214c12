#!/bin/bash
# usage (GPU box, repo root): tools/quick_bench.sh [extra bench args]   -- default bench on a cached equilibrated state, twice
cd $GRAFT_REPO_ROOT
C=gpurun_out/equil_pe10k.npz
[ -f $C ] || python bench.py --sims 1 --steps 1 --warmup 0 --nss 10 --no-cpu-baseline --equil-cache $C > /dev/null 2>&1
for i in 1 2; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --equil-cache $C "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'evals/s, ms/step', round(d['ms_per_step'],1), 'pair ms', round(d['roofline']['avg_launch_ms'],3), 'sims/launch', d['roofline']['sims_per_launch'], 'share', round(d['roofline']['rank0_pair_share_of_wall'],3), 'chk', d['config']['stress_zz_checksum_Pa'])"
done

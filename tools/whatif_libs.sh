#!/bin/bash
# same-box what-if builds of the library on short runs (usage on the GPU box: tools/whatif_libs.sh "1 9 72 576" lib1.so lib2.so ...): the pair
# kernel's chip-exclusive time per launch (bench.py's own HIP events, batch whole) with each library; results of what-if builds are WRONG by design
SIZES=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
[ -f $C ] || python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
for N in $SIZES; do
  for L in "$@"; do
    SCEMA_MD_LIB=$L python bench.py --sims $N --steps 2 --warmup 1 --nss 10 --no-cpu-baseline --monotonic-updates 0 --share8-updates 0 --reax-leg off --equil-cache $C 2>gpurun_out/whatif.err | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$N', '$L', 'k_pair whole ms per launch', round(r['whole_avg_launch_ms'],4), 'us per replica', round(1e3*r['whole_avg_launch_ms']/max(r['whole_sims_per_launch'],1),2), '; ms per update', round(d['ms_per_step'],2), flush=True)" || tail -2 gpurun_out/whatif.err
  done
done

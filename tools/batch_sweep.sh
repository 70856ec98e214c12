#!/bin/bash
# The batch-size curve.  usage on the GPU box, from the repo root: tools/batch_sweep.sh <tag> [sizes...]
# OPLS PE-10k at 1 2 4 9 12 18 24 36 48 72 144 576 replicas per update and the ReaxFF set at 9 and 72, on ONE box: per size a plain bench run
# (evaluations/s, ms per update, the pair kernel's chip-exclusive time per replica from the bench's own HIP events) and a profiled run
# of the same workload (kernel launches, busy and idle time per MD step).  Table -> gpurun_out/<tag>_batch_sweep.txt
T=${1:-r06_a}; shift
SIZES=${@:-1 2 4 9 12 18 24 36 48 72 144 576}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
OUT=gpurun_out/${T}_batch_sweep.txt
echo "# batch-size curve, tag $T, $(date -u +%FT%TZ), lib ${SCEMA_MD_LIB:-libscema_md.so}; OPLS PE-10k (10+100 MD steps per evaluation)" > $OUT
HDR=--header
for N in $SIZES; do
  S=$(( N >= 576 ? 6 : (N >= 144 ? 10 : (N >= 36 ? 16 : 30)) ))
  python bench.py --sims $N --steps $S --warmup 3 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > gpurun_out/${T}_sw_${N}.json.log 2> gpurun_out/${T}_sw_${N}.err || { tail -3 gpurun_out/${T}_sw_${N}.err; exit 1; }
  P=$(( S / 2 ))
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_swprof -- python bench.py --sims $N --steps $P --warmup 2 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
  python tools/sweep_row.py gpurun_out/${T}_sw_${N}.json.log gpurun_out/${T}_swprof k_pair $HDR | tee -a $OUT
  HDR=
  rm -rf gpurun_out/${T}_swprof
done
echo "# ReaxFF PE-1620 (10+20 MD steps per evaluation), anchor kernel k_rx_hrow (one launch per MD step and part batch)" | tee -a $OUT
for N in 9 72; do
  python bench.py --force-field reax --sims $N --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_swrx_${N}.json.log 2> gpurun_out/${T}_swrx_${N}.err || { tail -3 gpurun_out/${T}_swrx_${N}.err; exit 1; }
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_swprof -- python bench.py --force-field reax --sims $N --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python tools/sweep_row.py gpurun_out/${T}_swrx_${N}.json.log gpurun_out/${T}_swprof k_rx_hrow | tee -a $OUT
  rm -rf gpurun_out/${T}_swprof
done

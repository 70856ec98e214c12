#!/bin/bash
# the two shapes of k_pppm_solve: PPPM parity tests with each, then same-box bench at 72 and 576 replicas
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for W in 0 1; do SCEMA_MD_PPPM_SOLVE_WIDE=$W timeout -k 10 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_equil.py -x -q -m gpu -k "pppm or static or full_evaluation or mesh" 2>&1 | tail -2; done
tools/r05_72_ab.sh "narrow:" "wide:SCEMA_MD_PPPM_SOLVE_WIDE=1" "narrow_nosplit:SCEMA_MD_SPLIT=0" "wide_nosplit:SCEMA_MD_SPLIT=0 SCEMA_MD_PPPM_SOLVE_WIDE=1" "narrow2:"
C=gpurun_out/equil_pe10k.npz
for V in "narrow:" "wide:SCEMA_MD_PPPM_SOLVE_WIDE=1" "narrow2:" "wide2:SCEMA_MD_PPPM_SOLVE_WIDE=1"; do
  name=${V%%:*}; envs=${V#*:}
  env $envs python bench.py --steps 6 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('576 $name', round(d['value'],1), 'evals/s', round(d['ms_per_step'],1), 'ms per update', d['config']['env_overrides'], flush=True)"
done

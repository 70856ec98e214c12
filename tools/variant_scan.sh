#!/bin/bash
# usage (GPU box, repo root): tools/variant_scan.sh <file.hip> "<flags1>" "<flags2>" ...
# rebuilds one kernel file with extra -D flags per variant and prints the bench line + the per-kernel table rows of interest
F=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=gpurun_out/equil_pe10k.npz
[ -f $C ] || python bench.py --sims 1 --steps 1 --warmup 0 --nss 10 --no-cpu-baseline --equil-cache $C > /dev/null 2>&1
for v in "$@"; do
  touch scema_amd/csrc/$F
  make -C scema_amd/csrc HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics $v" 2>&1 | grep -E "error|spill"
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/vs_tmp -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --equil-cache $C ${SCAN_ARGS} > gpurun_out/vs_tmp.log 2>&1
  echo "[$v] $(cat gpurun_out/vs_tmp.log | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'evals/s')")"
  python tools/kernel_table.py gpurun_out/vs_tmp | grep -E "${SCAN_GREP:-k_}" | head -8
  rm -rf gpurun_out/vs_tmp
done

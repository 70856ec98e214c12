#!/bin/bash
# same-box comparison of builds of the library under rocprofv3 (batch whole): k_neigh_build / k_pair per launch
# usage on the GPU box: tools/neigh_whatif.sh <tag> "a:" "b:SCEMA_MD_LIB=libscema_md_b.so" ...
T=${1:-nwi}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > /dev/null 2>&1
for V in "$@"; do
  name=${V%%:*}; envs=${V#*:}
  export SCEMA_MD_SPLIT=0
  for kv in $envs; do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_${name}_prof -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > gpurun_out/${T}_$name.json.log 2>&1 || { tail -5 gpurun_out/${T}_$name.json.log; exit 1; }
  for kv in $envs; do unset ${kv%%=*}; done
  echo "== $name [$envs] $(grep '^{' gpurun_out/${T}_$name.json.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "evals/s")')"
  python tools/kernel_table.py gpurun_out/${T}_${name}_prof > gpurun_out/${T}_${name}_kernel_table.txt
  grep -E "k_neigh_build|k_pair<true|kernel time" gpurun_out/${T}_${name}_kernel_table.txt
  rm -rf gpurun_out/${T}_${name}_prof
done

#!/bin/bash
# exclusive ReaxFF kernel table (one part batch, one stream) of an A/B build: tools/r05_reax_excl_lib.sh <tag> [lib]
T=${1:-r05_rxw}; L=${2:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SCEMA_REAX_HALVES=1 SCEMA_REAX_OVERLAP=0
[ -n "$L" ] && export SCEMA_MD_LIB=$L
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_rprof -- python tools/reax_bench.py --updates 2 --warmup 1 --equil-steps 0 > gpurun_out/${T}_excl_bench.log 2>&1
python tools/kernel_table.py gpurun_out/${T}_rprof > gpurun_out/${T}_kernel_table_reax_72sims_exclusive.txt
rm -rf gpurun_out/${T}_rprof
head -12 gpurun_out/${T}_kernel_table_reax_72sims_exclusive.txt | cut -c1-150

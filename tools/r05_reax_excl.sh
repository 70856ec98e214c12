#!/bin/bash
# ReaxFF set, every kernel with the chip to itself (one part batch, one stream): kernel table + counters per launch
T=${1:-r05_rx}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SCEMA_REAX_HALVES=1 SCEMA_REAX_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_rprof -- python tools/reax_bench.py --updates 2 --warmup 1 --equil-steps 0 > gpurun_out/${T}_excl_bench.log 2>&1
python tools/kernel_table.py gpurun_out/${T}_rprof > gpurun_out/${T}_kernel_table_reax_72sims_exclusive.txt
rm -rf gpurun_out/${T}_rprof
head -30 gpurun_out/${T}_kernel_table_reax_72sims_exclusive.txt
tools/pmc_rx_kernels.sh "k_rx_" > gpurun_out/${T}_pmc_rx.log 2>&1
cp gpurun_out/reax_kernels_pmc.json gpurun_out/${T}_reax_kernels_pmc.json
tail -3 gpurun_out/${T}_pmc_rx.log | cut -c1-300

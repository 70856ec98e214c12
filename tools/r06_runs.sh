#!/bin/bash
# usage (on the GPU box, from the repo root): tools/r06_runs.sh <tag> [a|b]  -- the measurement set of round 6 (logs under gpurun_out/);
# a: the bench lines and kernel tables (steps 1-6), b: the batch-size curves of this tree and of the round-5 library (steps 7-8); default both
T=${1:-r06_z}
PART=${2:-ab}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C=gpurun_out/equil_pe10k.npz
if [[ $PART == *a* ]]; then
# 1. the driver's command: headline + the all-tensile set timed after it (config.strain_set_monotonic_evals_per_s) + CPU baseline on every host core
python bench.py --steps 20 --warmup 5 --equil-cache $C > gpurun_out/${T}_bench_576sims_driver.json.log 2> gpurun_out/${T}.err
grep "^{" gpurun_out/${T}_bench_576sims_driver.json.log | cut -c1-220
# 2. kernel table and idle gaps of the same workload
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > gpurun_out/${T}_prof_bench.json.log 2>&1
cp gpurun_out/${T}_prof/*/*kernel_stats.csv gpurun_out/${T}_kernel_stats_bench_576sims.csv
python tools/kernel_table.py gpurun_out/${T}_prof > gpurun_out/${T}_kernel_table_bench_576sims.txt
python tools/kernel_gaps.py gpurun_out/${T}_prof 16 > gpurun_out/${T}_kernel_gaps_bench_576sims.txt
rm -rf gpurun_out/${T}_prof
head -14 gpurun_out/${T}_kernel_table_bench_576sims.txt; head -8 gpurun_out/${T}_kernel_gaps_bench_576sims.txt
# 2b. the same with the batch whole (SCEMA_MD_SPLIT=0): the per-kernel times roofline.whole_* must agree with
SCEMA_MD_SPLIT=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_wprof -- python bench.py --steps 4 --warmup 4 --no-cpu-baseline --monotonic-updates 0 --reax-leg off --equil-cache $C > gpurun_out/${T}_wprof_bench.json.log 2>&1
python tools/kernel_table.py gpurun_out/${T}_wprof > gpurun_out/${T}_kernel_table_bench_576sims_whole.txt
rm -rf gpurun_out/${T}_wprof
head -8 gpurun_out/${T}_kernel_table_bench_576sims_whole.txt
# 3. one GPU's share of 8, the ragged strain set, one replica (BASELINE configs 3, 4 share, 2)
python bench.py --sims 72 --steps 10 --warmup 2 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C 2>/dev/null | grep "^{" > gpurun_out/${T}_bench_72sims_10updates.json.log
python bench.py --strain-set imbalanced --steps 6 --warmup 2 --no-cpu-baseline --reax-leg off --equil-cache $C 2>/dev/null | grep "^{" > gpurun_out/${T}_bench_576sims_imbalanced.json.log
python bench.py --sims 1 --steps 10 --warmup 2 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C 2>/dev/null | grep "^{" > gpurun_out/${T}_bench_1sim.json.log
# 4. the dogbone_file3D mesh: 4 864 quadrature points in one update (several launch groups), inputs_dogbone_file3D.json:36
python bench.py --sims 4864 --steps 2 --warmup 1 --no-cpu-baseline --monotonic-updates 0 --equil-cache $C 2>gpurun_out/${T}_4864.err | grep "^{" > gpurun_out/${T}_bench_4864sims.json.log
# 5. BASELINE config 5: the ReaxFF replica set, with its own roofline block and CPU baseline; kernel table of the same
python bench.py --force-field reax --steps 6 --warmup 2 > gpurun_out/${T}_bench_reax_72sims.json.log 2> gpurun_out/${T}_reax.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_rprof -- python bench.py --force-field reax --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${T}_rprof_bench.json.log 2>&1
cp gpurun_out/${T}_rprof/*/*kernel_stats.csv gpurun_out/${T}_kernel_stats_bench_reax_72sims.csv
python tools/kernel_table.py gpurun_out/${T}_rprof > gpurun_out/${T}_kernel_table_bench_reax_72sims.txt 2>/dev/null || python tools/kernel_median.py gpurun_out/${T}_rprof > gpurun_out/${T}_kernel_table_bench_reax_72sims.txt
python tools/kernel_gaps.py gpurun_out/${T}_rprof 10 k_rx_hrow > gpurun_out/${T}_kernel_gaps_bench_reax_72sims.txt
rm -rf gpurun_out/${T}_rprof
for f in gpurun_out/${T}_bench_72sims_10updates.json.log gpurun_out/${T}_bench_576sims_imbalanced.json.log gpurun_out/${T}_bench_1sim.json.log gpurun_out/${T}_bench_4864sims.json.log gpurun_out/${T}_bench_reax_72sims.json.log; do
  python -c "import sys,json; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); r=d['roofline']; print('$f', round(d['value'],1), round(d['ms_per_step'],2), d['config']['md_steps_per_eval'], r['bound'], r['frac'] and round(r['frac'],3), round(r['avg_launch_ms'],3), (d.get('cpu_baseline') or {}).get('value'))"
done
# 6. the ReaxFF set at the size of the headline batch
python bench.py --force-field reax --sims 576 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/${T}_bench_reax_576sims.json.log
python -c "import json; d=json.loads(open(\"gpurun_out/${T}_bench_reax_576sims.json.log\").read()); print(\"reax 576\", round(d[\"value\"],1), round(d[\"roofline\"][\"frac\"],3))"
fi
if [[ $PART == *b* ]]; then
# 7. the batch-size curve of the same tree on the same box, kernel tables of an 8-replica (the largest batch that runs whole) and a single-replica batch as they run
bash tools/batch_sweep.sh ${T} > gpurun_out/${T}_sweep.log 2>&1
bash tools/small_prof.sh ${T} 8 > /dev/null 2>&1
bash tools/small_prof.sh ${T} 1 > /dev/null 2>&1
cat gpurun_out/${T}_batch_sweep.txt
# 8. the curve of the round-5 library on the same box (scema_amd/libscema_md_r05.so: the tree of commit 9953bdf built with the same Makefile)
[ -f scema_amd/libscema_md_r05.so ] && SCEMA_MD_LIB=libscema_md_r05.so bash tools/batch_sweep.sh ${T}_r05lib 1 2 4 9 12 18 24 36 48 72 144 576 > gpurun_out/${T}_r05lib_sweep.log 2>&1 && cat gpurun_out/${T}_r05lib_batch_sweep.txt
fi

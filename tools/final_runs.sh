cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_u_bench_576sims_driver.json.log 2> gpurun_out/r02_u.err
grep "^{" gpurun_out/r02_u_bench_576sims_driver.json.log | cut -c1-200
C=gpurun_out/equil_pe10k.npz
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_u_prof -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline --equil-cache $C > gpurun_out/r02_u_prof_bench.json.log 2>&1
cp gpurun_out/r02_u_prof/*/*kernel_stats.csv gpurun_out/r02_u_kernel_stats_bench_576sims.csv
python tools/kernel_table.py gpurun_out/r02_u_prof > gpurun_out/r02_u_kernel_table_bench_576sims.txt
rm -rf gpurun_out/r02_u_prof
head -12 gpurun_out/r02_u_kernel_table_bench_576sims.txt
python bench.py --sims 72 --steps 10 --warmup 2 --no-cpu-baseline --equil-cache $C 2>/dev/null | grep "^{" > gpurun_out/r02_u_bench_72sims_10updates.json.log
python bench.py --strain-set imbalanced --steps 6 --warmup 2 --no-cpu-baseline --equil-cache $C 2>/dev/null | grep "^{" > gpurun_out/r02_u_bench_576sims_imbalanced.json.log
python bench.py --sims 1 --steps 10 --warmup 2 --no-cpu-baseline --equil-cache $C 2>/dev/null | grep "^{" > gpurun_out/r02_u_bench_1sim.json.log
for f in gpurun_out/r02_u_bench_72sims_10updates.json.log gpurun_out/r02_u_bench_576sims_imbalanced.json.log gpurun_out/r02_u_bench_1sim.json.log; do python -c "import sys,json; d=json.loads(open('$f').read()); print('$f', round(d['value'],1), round(d['ms_per_step'],2), d['config']['md_steps_per_eval'])"; done

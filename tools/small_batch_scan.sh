cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; C=gpurun_out/equil_pe10k.npz
for s in 1 2 4 8 16; do echo sims $s
python bench.py --sims $s --steps 10 --warmup 2 --no-cpu-baseline --equil-cache $C 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))"
SCEMA_MD_BIG_CELLS=1 python bench.py --sims $s --steps 10 --warmup 2 --no-cpu-baseline --equil-cache $C 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bigcells', round(d['value'],1), round(d['ms_per_step'],2))"
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2

cd $GRAFT_REPO_ROOT
C=gpurun_out/equil_pe10k.npz
python bench.py --sims 1 --steps 1 --warmup 0 --nss 10 --no-cpu-baseline --equil-cache $C > /dev/null 2>&1
for v in 0 1; do for i in 1 2; do
SCEMA_MD_SPLIT=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --equil-cache $C 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split=$v', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'pair ms', round(d['roofline']['avg_launch_ms'],3), 'sims/launch', d['roofline']['sims_per_launch'], 'share', round(d['roofline']['rank0_pair_share_of_wall'],3), 'chk', d['config']['stress_zz_checksum_Pa'])"
done; done
SCEMA_MD_SPLIT=1 python bench.py --sims 72 --steps 3 --warmup 1 --no-cpu-baseline --equil-cache $C 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('72 split=1', round(d['value'],1))"
SCEMA_MD_SPLIT=0 python bench.py --sims 72 --steps 3 --warmup 1 --no-cpu-baseline --equil-cache $C 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('72 split=0', round(d['value'],1))"

#!/bin/bash
# usage: tools/variant_scan.sh "<flags1>" "<flags2>" ...   (run on the GPU box: rebuilds md_pair.hip with extra -D flags, benches 72 sims)
for v in "$@"; do
  touch scema_amd/csrc/md_pair.hip
  make -C scema_amd/csrc HIPFLAGS="--offload-arch=gfx950 -munsafe-fp-atomics $v" 2>&1 | grep -E "error|spill" 
  for i in 1 2; do
    timeout 200 python bench.py --sims 72 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4))"
  done
done

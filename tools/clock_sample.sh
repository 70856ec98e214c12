#!/bin/bash
# sustained clock and power of the headline loop: rocm-smi sampled every 2 s next to a short bench run (read-only queries)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 12 --warmup 3 --no-cpu-baseline --reax-leg off --monotonic-updates 0 --equil-cache gpurun_out/equil_pe10k.npz > gpurun_out/r04_clock_bench.json.log 2>/dev/null &
BP=$!
: > gpurun_out/r04_clock_samples.txt
while kill -0 $BP 2>/dev/null; do
  (date +%s.%N; rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use|fclk") >> gpurun_out/r04_clock_samples.txt
  sleep 2
done
wait $BP
grep -E "sclk" gpurun_out/r04_clock_samples.txt | sort | uniq -c | sort -rn | head -8
grep -E "Power" gpurun_out/r04_clock_samples.txt | sort | uniq -c | sort -rn | head -8

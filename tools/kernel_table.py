#!/usr/bin/env python3
"""usage: tools/kernel_table.py <rocprofv3 output dir> -- per-kernel table of the batch launches (largest grid per kernel name only)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
gmax = collections.defaultdict(int)
for r in rows:
    g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y'])
    gmax[r['Kernel_Name']] = max(gmax[r['Kernel_Name']], g)
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for r in rows:
    if int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) != gmax[r['Kernel_Name']]:
        continue
    n = r['Kernel_Name'].split('(')[0]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[n][0] += 1; agg[n][1] += d; agg[n][2] = max(agg[n][2], d)
tot = sum(v[1] for v in agg.values())
nstep = (sum(v[0] for n, v in agg.items() if 'k_pair' in n) or sum(v[0] for n, v in agg.items() if n.strip().endswith('k_rx_bonds'))
         or max(v[0] for v in agg.values()))   # one k_pair (OPLS) or k_rx_bonds (ReaxFF) launch per MD step of the batch
print(f"kernel time {tot/1e9:.3f} s over {nstep} steps = {tot/nstep/1e3:.1f} us per step")
for n, v in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{n[:44]:44s} calls {v[0]:5d}  avg {v[1]/v[0]/1e3:9.1f} us  max {v[2]/1e3:9.1f} us  share {100*v[1]/tot:5.1f}%  per-step {v[1]/nstep/1e3:8.1f} us")

#!/usr/bin/env python3
"""usage: tools/kernel_median.py <rocprofv3 output dir> [name ...] -- median / mean / max duration of the batch launches (largest grid) of the
named kernels: the median of a kernel that exits early on most steps is the price of launching it for nothing."""
import collections, csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
want = sys.argv[2:]
rows = list(csv.DictReader(open(f)))
gmax = collections.defaultdict(int)
for r in rows:
    gmax[r['Kernel_Name']] = max(gmax[r['Kernel_Name']], int(r['Grid_Size_X']) * int(r['Grid_Size_Y']))
d = collections.defaultdict(list)
for r in rows:
    if int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) != gmax[r['Kernel_Name']]:
        continue
    n = r['Kernel_Name'].split('(')[0]
    if not want or any(w in n for w in want):
        d[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in sorted(d.items(), key=lambda x: -sum(x[1])):
    print(f"{n[:40]:40s} calls {len(v):5d}  min {min(v):8.1f} us  median {statistics.median(v):9.1f} us  mean {sum(v)/len(v):9.1f} us  max {max(v):9.1f} us")

#!/usr/bin/env python3
"""usage: tools/kernel_gaps.py <rocprofv3 output dir> [n] [anchor kernel, default k_pair] -- where the GPU idles: busy time (union of the kernel
intervals of all streams) against the wall between the first and the last batch launch of the anchor kernel, and the kernels that precede the
longest gaps."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
anchor = sys.argv[3] if len(sys.argv) > 3 else 'k_pair'
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0], int(r['Grid_Size_X']) * int(r['Grid_Size_Y'])) for r in csv.DictReader(open(f))]
rows.sort()
gmax = max(g for s, e, n, g in rows if anchor in n)
big = [i for i, r in enumerate(rows) if anchor in r[2] and r[3] == gmax]
lo, hi = big[len(big) // 3], big[-1]        # skip the set-up and the first third
t0, t1 = rows[lo][0], rows[hi][0]
busy, cur_e, gaps = 0, t0, collections.defaultdict(lambda: [0, 0])
last = None
for s, e, n, g in rows[lo:hi]:
    if s > cur_e:
        if last is not None:
            gaps[(last, n)][0] += 1; gaps[(last, n)][1] += s - cur_e
        cur_s = s
    busy += max(0, e - max(s, cur_e))
    if e > cur_e:
        cur_e, last = e, n
nk = sum(1 for i in big if lo <= i < hi)
print(f"wall {(t1-t0)/1e6:.2f} ms, busy {busy/1e6:.2f} ms, idle {(t1-t0-busy)/1e6:.2f} ms over {nk} {anchor} launches = {(t1-t0-busy)/nk/1e3:.1f} us idle per launch")
for (a, b), v in sorted(gaps.items(), key=lambda x: -x[1][1])[:top]:
    print(f"  {v[1]/1e3:9.1f} us in {v[0]:5d} gaps (avg {v[1]/v[0]/1e3:6.1f} us)  {a[:40]} -> {b[:40]}")

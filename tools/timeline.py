#!/usr/bin/env python3
"""usage: tools/timeline.py <rocprofv3 output dir> [first k_pair launch to show, default the middle] [launches to show, default 40]
-- the kernels of a few MD steps as they ran: start and end (us, relative), stream / queue, grid, name.  For reading how the streams of a
small batch interleave (tools/small_prof.sh leaves the trace when KEEP=1)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Stream_Id') or r.get('Queue_Id') or '0', int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1),
               r['Kernel_Name'].split('(')[0]) for r in csv.DictReader(open(f))))
pairs = [i for i, r in enumerate(rows) if 'k_pair' in r[4]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(pairs) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
i0 = pairs[first] - 6
t0 = rows[i0][0]
streams = sorted({r[2] for r in rows[i0:i0 + n]})
for s, e, q, g, name in rows[i0:i0 + n]:
    col = streams.index(q)
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  grid {g:8d}  " + "    " * col + f"[{q}] {name[:60]}")

#!/usr/bin/env python3
"""usage: tools/sweep_row.py <bench json log> <rocprofv3 output dir> [anchor kernel, default k_pair] -- one row of the batch-size curve
(tools/batch_sweep.sh): evaluations/s and ms per update from the plain bench run; from the kernel trace of a second, profiled run of the
same workload: the anchor kernel's time per replica, launches per MD step, busy and idle time per MD step (busy = union of all kernel
intervals between the first and the last anchor launch of the window; the first third of the anchor launches is skipped: set-up)."""
import csv, glob, json, sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
anchor = sys.argv[3] if len(sys.argv) > 3 else 'k_pair'
n = d['config']['n_sims']
r = d['roofline']
f = glob.glob(sys.argv[2] + '/*/*kernel_trace.csv')
row = {'sims': n, 'evals_per_s': d['value'], 'ms_per_update': d['ms_per_step'], 'md_steps_per_eval': d['config']['md_steps_per_eval']}
row['anchor_us_per_replica_whole'] = 1e3 * r['whole_avg_launch_ms'] / max(r.get('whole_sims_per_launch') or n, 1) if r.get('whole_avg_launch_ms') else None
if f:
    rows = [(int(x['Start_Timestamp']), int(x['End_Timestamp']), x['Kernel_Name'].split('(')[0], int(x['Grid_Size_X']) * int(x.get('Grid_Size_Y', 1) or 1),
             x.get('Stream_Id') or x.get('Queue_Id') or '0') for x in csv.DictReader(open(f[0]))]
    rows.sort()
    gmax = max(g for s, e, k, g, q in rows if anchor in k)
    # whole batches and part batches (run_phase: three parts for 9 replicas, four for 10-63, two halves beyond; the
    # smallest part of p holds at least n // p of the n replicas); not the one-replica equilibration
    pmax = 1 if n < 9 or anchor != 'k_pair' else 3 if n < 10 else 4 if n < 64 else 2
    if anchor != 'k_pair': pmax = 2
    fmin = (n // pmax) / n * 0.999 if n >= 2 * pmax else 0.5
    fmin = min(fmin, 0.5) if pmax > 1 else 0.999 if n > 1 else 0.5
    is_big = lambda x: anchor in x[2] and x[3] >= fmin * gmax
    big = [i for i, x in enumerate(rows) if is_big(x)]
    lo, hi = big[len(big) // 3], big[-1]
    win = rows[lo:hi]
    streams = {x[4] for x in win if is_big(x)}
    nanchor = sum(1 for x in win if is_big(x))
    steps = nanchor / max(len(streams), 1)          # MD steps of the batch in the window (each part batch has its own stream)
    busy, cur = 0, rows[lo][0]
    for s, e, k, g, q in win:
        busy += max(0, e - max(s, cur))
        cur = max(cur, e)
    wall = rows[hi][0] - rows[lo][0]
    ak = [x for x in win if is_big(x)]
    row.update({'launches_per_step': len(win) / steps, 'busy_us_per_step': busy / steps / 1e3, 'idle_us_per_step': (wall - busy) / steps / 1e3,
                'wall_us_per_step': wall / steps / 1e3, 'anchor_avg_us': sum(x[1] - x[0] for x in ak) / len(ak) / 1e3,
                'anchor_streams': len(streams), 'anchor_us_per_replica_as_run': sum(x[1] - x[0] for x in ak) / len(ak) / 1e3 / (n / max(len(streams), 1))})
fmt = lambda v: '-' if v is None else (f"{v:.1f}" if isinstance(v, float) else str(v))
cols = ['sims', 'evals_per_s', 'ms_per_update', 'anchor_us_per_replica_whole', 'anchor_us_per_replica_as_run', 'anchor_avg_us', 'anchor_streams', 'launches_per_step',
        'busy_us_per_step', 'idle_us_per_step', 'wall_us_per_step']
if '--header' in sys.argv:
    print(' '.join(f"{c:>14s}" for c in ['sims', 'evals/s', 'ms/update', 'anch us/rep W', 'anch us/rep R', 'anchor avg us', 'streams', 'launch/step', 'busy us/step', 'idle us/step', 'wall us/step']))
print(' '.join(f"{fmt(row.get(c)):>14s}" for c in cols), flush=True)

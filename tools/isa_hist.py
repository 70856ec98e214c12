#!/usr/bin/env python3
"""usage: tools/isa_hist.py <file.s> <kernel name substring> [--loops] -- opcode histogram of one kernel of an `hipcc -S` listing (the habit that
found the 330 wasted instructions of the PPPM weights in round 5); with --loops also per basic block, largest first, so that the loop bodies show"""
import collections, re, sys
src, name = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(name) + r'\w*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
tot, blocks, cur = collections.Counter(), [], ['entry', collections.Counter()]
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') and not t.endswith(':'):
        continue
    if t.endswith(':') or re.match(r'^\.LBB\w+:', t):
        blocks.append(cur); cur = [t.split(':')[0], collections.Counter()]; continue
    op = t.split()[0]
    if not re.match(r'^[a-z_0-9]+$', op):
        continue
    tot[op] += 1; cur[1][op] += 1
blocks.append(cur)
cls = lambda op: ('valu_f64' if re.search(r'_f64|_rsq_f64|_rcp_f64', op) else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') and not op.startswith('s_load') and not op.startswith('s_waitcnt') else
                  'smem' if op.startswith('s_load') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'wait' if op.startswith('s_waitcnt') else 'other')
by = collections.Counter()
for op, n in tot.items():
    by[cls(op)] += n
print(f"kernel {name}: {sum(tot.values())} instructions;", ', '.join(f"{k} {v}" for k, v in by.most_common()))
for op, n in tot.most_common(40):
    print(f"  {n:6d} {op}")
if '--loops' in sys.argv:
    for b in sorted(blocks, key=lambda b: -sum(b[1].values()))[:8]:
        n = sum(b[1].values())
        c = collections.Counter()
        for op, k in b[1].items():
            c[cls(op)] += k
        print(f"block {b[0]}: {n} instructions;", ', '.join(f"{k} {v}" for k, v in c.most_common()), '; top:', ', '.join(f"{op} {k}" for op, k in b[1].most_common(8)))

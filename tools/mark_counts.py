#!/usr/bin/env python3
"""Static instruction counts between the phase marks of a render kernel (build with -DKY_MARKS -S): tools/mark_counts.py <file.s> <mangled-name prefix>
Approximate by nature (linear assembly order, loops counted once); meant to show which region is heavy per visit."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
prefix = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ':' in l.split(';')[0])
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
cur = 'pre'; counts = {cur: [0, 0, 0, 0]}; order = [cur]
trans = ('v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos', 'v_exp', 'v_log')
for l in lines[start:end]:
    m = re.search(r'; KYMARK (-?\d+)', l)
    if m:
        cur = 'after mark %s #%d' % (m.group(1), len(order)); order.append(cur); counts[cur] = [0, 0, 0, 0]; continue
    t = l.strip().split(' ')[0] if l.strip() else ''
    if t.startswith('v_'):
        counts[cur][0] += 1
        if t.startswith(trans): counts[cur][1] += 1
    elif t.startswith('s_'): counts[cur][2] += 1
    elif t.startswith(('ds_', 'global_', 'scratch_', 'buffer_', 'flat_')): counts[cur][3] += 1
for k in order:
    print("%-22s VALU %5d (trans %3d)  SALU %5d  MEM %4d" % (k, *counts[k]))
print("total VALU", sum(c[0] for c in counts.values()))

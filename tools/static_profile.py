#!/usr/bin/env python3
"""Static VALU profile of one kernel from a `hipcc -S -gline-tables-only` listing: VALU instructions per source line range.
usage: tools/static_profile.py <file.s> <kernel symbol prefix> [--lines]"""
import re, sys, collections
path, sym = sys.argv[1], sys.argv[2]
txt = open(path).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt):
    files[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]
start = txt.index("\n" + sym)
end = txt.index("s_endpgm", start)
loc = ("?", 0)
per_line = collections.Counter(); per_line_trans = collections.Counter()
TR = ("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log", "v_mul_lo", "v_mul_hi", "v_mad_u64")
tot = 0
for l in txt[start:end].split("\n"):
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        loc = (files.get(m.group(1), m.group(1)), int(m.group(2))); continue
    m = re.match(r'\s+(v_[a-z0-9_]+)', l)
    if m:
        op = m.group(1)
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): continue
        per_line[loc] += 1; tot += 1
        if op.startswith(TR): per_line_trans[loc] += 1
print("total static VALU:", tot)
# group by function ranges given on the command line as file:lo-hi=name, else per line top 60
if "--lines" in sys.argv:
    for k, v in per_line.most_common(70):
        print("%-18s %5d  valu %4d  quarter-rate %3d" % (k[0], k[1], v, per_line_trans[k]))
else:
    import bisect
    # function start lines from the sources
    import os
    root = "/root/repo/ky_amd/csrc/"
    funcs = {}
    for fn in ("ky_device.hpp", "kyhip.hip"):
        starts = []
        for i, line in enumerate(open(root + fn), 1):
            m = re.match(r'(?:template\s*<[^>]*>\s*)?(?:KY_DEV|__global__|static|inline)\b.*?\b([A-Za-z_0-9]+)\s*\(', line)
            if m and not line.startswith(" "): starts.append((i, m.group(1)))
        funcs[fn] = starts
    agg = collections.Counter(); aggt = collections.Counter()
    for (f, ln), v in per_line.items():
        st = funcs.get(f)
        name = f + ":?"
        if st:
            idx = bisect.bisect_right([s[0] for s in st], ln) - 1
            if idx >= 0: name = st[idx][1]
        agg[name] += v; aggt[name] += per_line_trans[(f, ln)]
    for k, v in agg.most_common(50):
        print("%-32s valu %5d  quarter-rate %4d" % (k, v, aggt[k]))

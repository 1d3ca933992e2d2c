#!/usr/bin/env python3
"""Where does a kernel use its highest VGPRs?  usage: tools/vgpr_peaks.py <listing.s from -S -gline-tables-only [-DKY_MARKS]> <kernel symbol prefix> [threshold]"""
import re, sys, collections
path, sym = sys.argv[1], sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 80
txt = open(path).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt):
    files[m.group(1)] = (m.group(3) or m.group(2)).split('/')[-1]
start = txt.index("\n" + sym); end = txt.index("s_endpgm", start)
loc = None; mark = None; hi = collections.Counter(); top = 0
for l in txt[start:end].split("\n"):
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m: loc = (files.get(m.group(1), '?'), int(m.group(2))); continue
    m = re.search(r'; KYMARK (-?\d+)', l)
    if m: mark = m.group(1); continue
    if not re.match(r'\s+[vsdgbf]', l): continue
    regs = [int(x) for x in re.findall(r'\bv(\d+)\b', l)] + [int(b) for a, b in re.findall(r'v\[(\d+):(\d+)\]', l)]
    if regs:
        top = max(top, max(regs))
        if max(regs) >= thr: hi[(mark, loc)] += 1
print("highest VGPR index:", top)
for k, v in sorted(hi.items(), key=lambda x: -x[1])[:30]: print(k, v)

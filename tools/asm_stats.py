#!/usr/bin/env python3
"""tools/asm_stats.py <listing.s> : per render kernel: instructions, code bytes (from .size or estimated), by class (valu / salu / smem / vmem / lds / scratch)"""
import re, sys, collections
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\n(_Z\w*render_kernel\w*):", txt):
    sym = m.group(1)
    start = m.end()
    end = txt.index("s_endpgm", start)
    body = txt[start:end]
    cls = collections.Counter()
    n = 0
    for l in body.split("\n"):
        mm = re.match(r"\s+([a-z][a-z0-9_]+)\s", l + " ")
        if not mm: continue
        op = mm.group(1)
        if op.startswith("."): continue
        n += 1
        if op.startswith("v_"): cls["valu"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer"): cls["smem"] += 1
        elif op.startswith("s_"): cls["salu"] += 1
        elif op.startswith("ds_"): cls["lds"] += 1
        elif op.startswith("scratch_"): cls["scratch"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): cls["vmem"] += 1
        else: cls["other"] += 1
    short = re.sub(r"EvPKN3kyd.*", "", sym.replace("_Z13render_kernel", "").replace("_ZN3kyd15render_kernel_q", "q"))
    print("%-28s instr %6d  " % (short, n) + "  ".join("%s %d" % kv for kv in sorted(cls.items())))

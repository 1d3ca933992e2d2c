#!/usr/bin/env python3
"""tools/asm_by_func.py <listing.s (tools/asm.sh)> <kernel symbol substring> : static instructions of one kernel attributed to the source
function whose lines they were generated from (.loc), split VALU / SALU / memory.  Inlined copies are summed."""
import re, sys, collections, os
path, key = sys.argv[1], sys.argv[2]
txt = open(path).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt):
    files[m.group(1)] = os.path.basename(m.group(3) or m.group(2))
# function line ranges of our sources
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ky_amd", "csrc")
ranges = {}
for fn in ("ky_device.hpp", "ky_render.hpp", "ky_launch.hip", "ky_queue.hpp"):
    src = open(os.path.join(os.environ.get("KY_SRC_DIR", root), fn)).read().split("\n")
    cur = []
    for i, l in enumerate(src, 1):
        m = re.match(r"(?:template\s*<[^>]*>\s*)?(?:KY_DEV|__global__|static|__device__)[^;{]*?\b([A-Za-z_]\w*)\s*\([^;]*$", l)
        if m and not l.startswith(" "):
            cur.append((i, m.group(1)))
    ranges[fn] = cur
def func_of(fn, line):
    best = "?"
    for start, name in ranges.get(fn, []):
        if start <= line: best = name
        else: break
    return best
m = re.search(r"\n(_Z\w*%s\w*):" % re.escape(key), txt)
start = m.end(); end = txt.index("s_endpgm", start)
loc = ("?", 0)
last = "(kernel prologue)"
HELPER_END = next(i for i, l in enumerate(open(os.path.join(os.environ.get("KY_SRC_DIR", root), "ky_device.hpp")).read().split("\n"), 1) if "device scene layout" in l or "// random numbers:" in l)
cnt = collections.defaultdict(collections.Counter)
for l in txt[start:end].split("\n"):
    mm = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if mm: loc = (files.get(mm.group(1), "?"), int(mm.group(2))); continue
    mm = re.match(r"\s+([a-z][a-z0-9_]+)\s", l + " ")
    if not mm or mm.group(1).startswith("."): continue
    op = mm.group(1)
    c = "valu" if op.startswith("v_") else "salu" if (op.startswith("s_") and not op.startswith("s_load")) else "mem"
    f = func_of(*loc) if loc[0] in ranges else None
    # the vector / scalar helpers at the top of ky_device.hpp and the HIP headers are transparent: charged to the last real function seen
    if f is None or f == "?" or (loc[0] == "ky_device.hpp" and loc[1] < HELPER_END) or f in ("mix32",):
        f = last
    else:
        last = f
    cnt[f][c] += 1
tot = collections.Counter()
for f, c in sorted(cnt.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-34s valu %5d  salu %5d  mem %4d" % (f, c["valu"], c["salu"], c["mem"]))
    tot.update(c)
print("%-34s valu %5d  salu %5d  mem %4d" % ("TOTAL", tot["valu"], tot["salu"], tot["mem"]))

#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output for the render kernels: tools/variant_usage.py <log>..."""
import re, sys
for path in sys.argv[1:]:
    name = path.split("/")[-1].rsplit(".", 1)[0]
    cur = None; rows = {}
    for line in open(path, errors="replace"):
        m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
        if m:
            cur = m.group(1); rows[cur] = {}; continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
    for fn, r in rows.items():
        if "render_kernelILb0" not in fn or "render_kernel_q" in fn: continue
        tag = "hot" if "ELi48E" in fn else "gen"
        print("%-14s %s  vgpr %3s spill %3s | sgpr %3s spill %3s | scratch %4s occ %s" % (name, tag, r.get("VGPRs"), r.get("VGPRs Spill"), r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("ScratchSize"), r.get("Occupancy")))

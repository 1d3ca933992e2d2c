#!/usr/bin/env python3
"""tools/resources.py <hipcc -Rpass-analysis=kernel-resource-usage log> : one line per render kernel (registers, spills, scratch, LDS)."""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: Function Name: ", txt)[1:]
for b in blocks:
    name = b.split()[0]
    if "render_kernel" not in name and "--all" not in sys.argv:
        continue
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        name = name.replace("void ", "").split(">(")[0] + ">"
    except Exception:
        pass
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print("%-62s VGPR %3s spill %3s  SGPR %3s spill %3s  scratch %4s B  LDS %6s  occ %s" % (
        name[:70], g("VGPRs"), g("VGPRs Spill"), g("TotalSGPRs"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))

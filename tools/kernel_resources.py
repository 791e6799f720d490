#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (one line per kernel)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
K_SCR = r"ScratchSize \[bytes/lane\]"
K_OCC = r"Occupancy \[waves/SIMD\]"
for b in blocks:
    name = b.split(" ")[0]

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"

    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("rgcn::", "").split("(")[0]
    print("%-50s sgpr=%s vgpr=%s agpr=%s scratch=%s spill=%s occ=%s" % (
        dn[:50], g("TotalSGPRs"), g("VGPRs"), g("AGPRs"), g(K_SCR), g("VGPRs Spill"), g(K_OCC)))

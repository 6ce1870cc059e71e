#!/usr/bin/env python3
"""VGPRs / spills / scratch / occupancy per kernel from a -Rpass-analysis=kernel-resource-usage log
(`make -C snowmocap_amd/csrc asm` writes build/resource_usage.txt).  usage: resource_usage.py <log> [name filter ...]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
filters = sys.argv[2:]
rows = []
for b in re.split(r"remark: [^\n]*?Function Name: ", txt)[1:]:
    name = b.split()[0]
    g = lambda k: (lambda m: int(m.group(1)) if m else -1)(re.search(k + r": (\d+)", b))
    rows.append((name, g("VGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = n.replace("snowtri::", "").replace("void ", "").split("(")[0]
    if not filters or any(k in n for k in filters):
        print("%-64s VGPR %3d  spilled %3d  scratch %4d B  waves/SIMD %d" % (n[:64], r[1], r[2], r[3], r[4]))

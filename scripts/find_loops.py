#!/usr/bin/env python3
"""Dev aid: list the loops of a kernel (from `make -C snowmocap_amd/csrc asm`) that contain a given instruction,
with their static VALU / fp64 / LDS / scratch counts.   usage: find_loops.py <kernel substring> [needle=v_rsq_f64] [maxlen=600]"""
import re, sys, os
ASM = os.environ.get("SNOWTRI_ASM", "snowmocap_amd/csrc/build/snowtri-hip-amdgcn-amd-amdhsa-gfx950.s")
name = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "v_rsq_f64"
maxlen = int(sys.argv[3]) if len(sys.argv) > 3 else 600
lines = open(ASM).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and name in l][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
k = lines[start:end]
labels = {}
for i, l in enumerate(k):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
for i, l in enumerate(k):
    m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a, b = labels[m.group(1)], i
        body = [x for x in k[a:b + 1] if re.match(r"\s+[a-z]", x)]
        n = sum(needle in x for x in body)
        if n and b - a < maxlen:
            cnt = lambda rx: sum(1 for x in body if re.match(rx, x))
            pats = {"valu": r"\s+v_", "f64": r"\s+v_\w+_f64", "ds": r"\s+ds_", "vmem": r"\s+(global|buffer|flat)_", "scratch": r"\s+scratch_",
                    "salu": r"\s+s_", "lane": r"\s+v_(read|write)lane"}
            print("lines %d-%d (%d insts): %s x%d  " % (start + a, start + b, len(body), needle, n) + "  ".join("%s %d" % (kk, cnt(v)) for kk, v in pats.items()))

#!/usr/bin/env python3
"""Dev aid: per-basic-block instruction / VGPR statistics of one kernel in the -save-temps ISA.
usage: blockstats.py <asm.s> <mangled-name-prefix> [min_instrs]"""
import re, sys
from collections import Counter
path, prefix = sys.argv[1], sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ':' in l and not l.startswith('\t'))
end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i])
blocks = []; cur = ['entry', [], start]; blocks.append(cur)
for n in range(start + 1, end):
    l = lines[n]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = [m.group(1), [], n]; blocks.append(cur); continue
    t = l.strip()
    if t and not t.startswith(';') and not t.startswith('.'):
        cur[1].append(t)
for name, ins, n in blocks:
    if len(ins) < minins: continue
    regs = set()
    for i in ins:
        for m in re.finditer(r'\bv(\d+)\b', i): regs.add(int(m.group(1)))
        for m in re.finditer(r'\bv\[(\d+):(\d+)\]', i):
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    c = Counter(i.split()[0] for i in ins)
    f64 = sum(v for k, v in c.items() if 'f64' in k)
    trans = sum(v for k, v in c.items() if re.match(r'v_(rcp|rsq|sqrt)_f64', k))
    print(f"{name:10s} line {n - start:5d} n={len(ins):4d} f64={f64:4d} trans={trans:2d} vgprs={len(regs):3d} max=v{max(regs) if regs else -1:<3d}",
          dict(c.most_common(6)))

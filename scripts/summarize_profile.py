#!/usr/bin/env python3
"""Condense rocprofv3 output directories (scripts/profile.sh) into a short text + JSON summary."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
summary = {}


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


for path in find("*kernel_stats.csv"):
    rows = list(csv.DictReader(open(path)))
    print(f"== {os.path.relpath(path, out)}")
    keep = []
    for r in rows[:12]:
        name = r.get("Name", "")[:90]
        print(f"  {name:90s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} total_ns={r.get('TotalDurationNs')} pct={r.get('Percentage')}")
        keep.append({k: r.get(k) for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
    summary["kernel_stats"] = keep

for tag in ("pmc_fetch", "pmc_write", "pmc_sq"):
    files = [p for p in find("*counter_collection.csv") if f"/{tag}/" in p]
    for path in files:
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(path)):
            key = r.get("Kernel_Name", "")[:80] + " grid=" + str(r.get("Grid_Size") or r.get("Grid_Size_X") or "?")
            acc[key][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
        print(f"== {os.path.relpath(path, out)}")
        for kname, ctrs in acc.items():
            if "snowtri" not in kname:
                continue
            line = {c: sum(v) / len(v) for c, v in ctrs.items()}
            n = len(next(iter(ctrs.values())))
            print(f"  {kname}  launches={n}  " + "  ".join(f"{c}={x:.6g}" for c, x in line.items()))
            summary.setdefault(tag, {})[kname] = dict(launches=n, **line)
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)

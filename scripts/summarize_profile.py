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
    summary.setdefault("kernel_stats", {})[os.path.relpath(path, out).split(os.sep)[0]] = keep

for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
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

# HBM bytes per fused launch, scaled by the calibration streams of the same passes (known byte counts, same access shapes)
def mean_of(tag, needle, counter):
    for kname, line in summary.get(tag, {}).items():
        if needle in kname:
            return line.get(counter)
    return None


try:
    known = None
    for ln in open(os.path.join(out, "pmc_fetch.log")):
        if ln.startswith("known_bytes"):
            _, rb, wb, _, frames = ln.split()
            known = (int(rb), int(wb), int(frames))
    fr_cal, fr_k = mean_of("pmc_fetch", "k_calib_read12", "FETCH_SIZE"), mean_of("pmc_fetch", "k_fused_lean", "FETCH_SIZE")
    wr_cal, wr_k = mean_of("pmc_write", "k_calib_write16", "WRITE_SIZE"), mean_of("pmc_write", "k_fused_lean", "WRITE_SIZE")
    if known and fr_cal and wr_cal and fr_k and wr_k:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        scale_r, scale_w = known[0] / (fr_cal * 1024.0), known[1] / (wr_cal * 1024.0)
        import re
        kfull = next(k for k in summary["pmc_fetch"] if "k_fused_lean" in k)
        kshort = re.search(r"(k_fused_lean\w*<[^(]*>)", kfull).group(1).replace(" ", "")      # what snowtri_last_kernel_names() reports ...
        kshort = re.sub(r"^(k_fused_lean(?:_coop)?<\d+,\w+,\d+),float>$", r"\1>", kshort)   # ... which omits the defaulted output type of the float32 instances
        traffic = {"kernel": kshort, "frames_per_launch": known[2],
                   "workload": "BASELINE configs[1]: %d frames per launch" % known[2],
                   "FETCH_SIZE_KB_raw": fr_k, "WRITE_SIZE_KB_raw": wr_k,
                   "calibration": {"how": "snowtri_calib_stream in the same rocprofv3 pass: 12-byte records read per lane / 16-byte records "
                                          "written per lane, known byte counts", "read_bytes": known[0], "FETCH_SIZE_KB_raw": fr_cal,
                                   "read_scale": scale_r, "write_bytes": known[1], "WRITE_SIZE_KB_raw": wr_cal, "write_scale": scale_w},
                   "hbm_read_bytes_per_launch": fr_k * 1024.0 * scale_r, "hbm_write_bytes_per_launch": wr_k * 1024.0 * scale_w,
                   "hbm_bytes_per_launch": fr_k * 1024.0 * scale_r + wr_k * 1024.0 * scale_w,
                   "algorithmic_bytes_per_launch": 8512 * known[2], "source_sha256": bench.kernel_source_hash()}
        # VALU wave-instructions per 64 joints, from the SQ pass on the large launch (its frame count from that run's bench line)
        try:
            frames_large = None
            for ln in open(os.path.join(out, "pmc_sq.log")):
                if ln.startswith("{") and '"large_batch"' in ln:
                    frames_large = json.loads(ln)["large_batch"]["frames"]
            insts = None
            for kname, line in summary.get("pmc_sq", {}).items():
                if "k_fused_lean" in kname and (insts is None or line.get("SQ_INSTS_VALU", 0) > insts):
                    insts = line.get("SQ_INSTS_VALU")          # the large launch is the one with the most instructions
            if frames_large and insts:
                items = frames_large * 133 / 64.0
                traffic["valu"] = {"SQ_INSTS_VALU_per_launch": insts, "frames": frames_large, "wave_items": items, "per_64_joints": insts / items,
                                   "kernel": "k_fused_lean (the wave-autonomous kernel of large launches; k_fused_lean_coop runs the same item)"}
                print("== VALU wave-instructions per 64 joints: %.1f (SQ_INSTS_VALU %.6g on %d frames)" % (insts / items, insts, frames_large))
        except Exception as e:
            print("valu summary failed:", repr(e))
        json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
        print("== HBM traffic per fused launch: %.2f MB read + %.2f MB written = %.2f MB vs %.2f MB algorithmic (read scale %.3f, write scale %.3f)" % (
            traffic["hbm_read_bytes_per_launch"] / 1e6, traffic["hbm_write_bytes_per_launch"] / 1e6, traffic["hbm_bytes_per_launch"] / 1e6,
            traffic["algorithmic_bytes_per_launch"] / 1e6, scale_r, scale_w))
except Exception as e:      # a summary without the traffic block is still useful
    print("traffic summary failed:", repr(e))
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)

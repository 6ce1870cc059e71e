#!/usr/bin/env python3
"""profiles/<tag>/summary.json: every number the documents quote, derived from the files committed beside it.

    python scripts/build_summary.py r03

Inputs (all under profiles/<tag>/, copied there by scripts/collect_profiles.sh): the rocprofv3 kernel-stats CSVs, the
individual durations of the large launches, the counter means (counters.json, multi_person_counters.txt), pmc_traffic.json
and the two bench lines taken without the profiler.  scripts/make_tables.py renders the tables of README.md, DESIGN.md and
profiles/README.md from summary.json; tests/test_docs_tables.py fails when a table and summary.json disagree, and when
summary.json and the CSVs disagree.
"""
import ast
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8.0e12
FP64_PEAK = 78.6e12
FLOP_PER_SOLVE = 90.0
J = 133
BYTES_PER_FRAME = 8512            # 4 cameras x 1 person: 12 C P J in + 16 Pout J out


def short(name):
    """snowtri kernel name without its argument list."""
    m = re.search(r"snowtri::(\w+(?:<[^(]*>)?)\(", name)
    n = m.group(1).replace(" ", "") if m else name[:60]
    # the lean kernels carry the output type as a defaulted fourth template argument since round 5: the float32 instances keep
    # the name snowtri_last_kernel_names (and every earlier record) gives them
    return re.sub(r"^(k_fused_lean(?:_coop)?<\d+,\w+,\d+),float>$", r"\1>", n)


def kernel_stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        if "snowtri::" in r["Name"]:
            out[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                     "max_us": float(r["MaxNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6}
    return out


def multi_counters(path):
    """multi_person_counters.txt lines `a3 <kernel> grid=N {counter: 'value', ...} n=K` -> {cfg: {kernel: {...}}} (the
    largest grid of a kernel = its full-size launch)."""
    res = {}
    for ln in open(path):
        m = re.match(r"^([ab])([35]) (\S.*?) grid=(\d+) (\{.*\}) n=(\d+)", ln)
        if not m:
            continue
        cfg, kern, grid, ctr = "cfg" + m.group(2), m.group(3).replace(" ", ""), int(m.group(4)), ast.literal_eval(m.group(5))
        slot = res.setdefault(cfg, {}).setdefault(kern, {"grid": 0})
        if grid >= slot["grid"]:
            if grid > slot["grid"]:
                slot.clear()
            slot["grid"] = grid
            slot.update({k: float(v) for k, v in ctr.items()})
    for cfg in res.values():
        for k in cfg.values():
            if "SQ_ACTIVE_INST_VALU" in k and "GRBM_GUI_ACTIVE" in k and k["GRBM_GUI_ACTIVE"] > 0:
                # VALU busy: active VALU cycles (x 4: one count per 4-cycle issue) over the SIMD cycles of the launch
                k["valu_busy"] = k["SQ_ACTIVE_INST_VALU"] * 4.0 / (k["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            if k.get("SQ_ACTIVE_INST_LDS"):
                k["lds_conflict_ratio"] = k.get("SQ_LDS_BANK_CONFLICT", 0.0) / k["SQ_ACTIVE_INST_LDS"]
    return res


def bench_digest(line):
    d = json.loads(line)
    r, rep = d.get("roofline") or {}, d.get("repeats") or {}     # (the driver's record keeps the contract keys only: the rest may be absent)
    out = {"value": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
           "ms_per_step_min": rep.get("ms_per_step_min", d["ms_per_step"]), "ms_per_step_max": rep.get("ms_per_step_max", d["ms_per_step"]),
           "roofline_frac": r.get("frac"), "roofline_kernel": r.get("kernel"),
           "kernel_us_mean": (r.get("kernel_ms_mean") or 0.0) * 1e3, "kernel_us_min": (r.get("kernel_ms_min") or 0.0) * 1e3,
           "kernel_us_overlapped": (r.get("kernel_ms_mean_overlapped") or 0.0) * 1e3,
           "kernel_us_bracketed": (r.get("kernel_ms_mean_bracketed") or 0.0) * 1e3,
           "kernel_us_step_one_stream": (r.get("kernel_ms_step_one_stream") or 0.0) * 1e3,
           "copy_GBs": r.get("measured_device_copy_GBs"), "frac_of_copy": r.get("frac_of_measured_copy"),
           "frac_bracketed": r.get("frac_bracketed"), "frac_step_one_stream": r.get("frac_step_one_stream"),
           "roofline_traffic": r.get("traffic"), "roofline_region_frac": (d.get("roofline_region") or {}).get("frac"),
           "valu_per_64": (d.get("fp64_valu_issue") or {}).get("valu_insts_per_64_joints")}
    if d.get("large_batch"):
        out["large"] = {"frames": d["large_batch"]["frames"], "kernel_ms": d["large_batch"]["kernel_ms"], "frac": d["large_batch"]["frac"],
                        "joints_per_s": d["large_batch"]["joints_per_s"]}
    if d.get("extra_workloads"):
        out["extra"] = [{"workload": e["workload"], "kernel": e["kernel"], "frames": e["frames"], "kernel_ms": e["kernel_ms"],
                         "frames_per_s": e["frames_per_s"], "pair_solves_per_s": e["pair_solves_per_s"], "frac": e["roofline"]["frac"],
                         "fall_backs": e.get("fall_back_frames_last_segment"),
                         "two_streams_frames_per_s": (e.get("two_streams") or {}).get("frames_per_s"),
                         "two_streams_frac": (e.get("two_streams") or {}).get("frac"),
                         # (round 6: the lines are found by what they are, not by their position)
                         "method": e.get("method", "pairwise"), "zero_fill": e.get("zero_fill", True), "pout_max": e.get("pout_max"),
                         "written_bytes_per_frame": e.get("output_bytes_written_per_frame"), "joints_per_s": e.get("joints_per_s"),
                         "hbm_frac": (e.get("hbm") or {}).get("frac_of_8TBs")}
                        for e in d["extra_workloads"]]
    if d.get("cpu_baseline"):
        c = d["cpu_baseline"]
        out["cpu"] = {"value": c["value"], "cores": c["cores"], "gpu_vs_oracle_max_abs_m": c["gpu_vs_oracle_max_abs_m"]}
    if d.get("per_frame_api"):
        p = d["per_frame_api"]
        out["per_frame"] = {"api_sequence_us": p["api_sequence_us_median"], "fused_host_call_us": p["fused_host_call_us_median"],
                            "reference_ms": p["reference_ms_per_frame"]}
    return out


def build(tag):
    d = os.path.join(ROOT, "profiles", tag)
    s = {"tag": tag}
    fast = kernel_stats(os.path.join(d, "kernel_stats_cfg2_10k.csv"))
    name = next(k for k in fast if k.startswith("k_fused_lean"))
    f = fast[name]
    s["fast_kernel_10k"] = dict(f, kernel=name, frames=10000, algorithmic_bytes=BYTES_PER_FRAME * 10000,
                                hbm_frac=BYTES_PER_FRAME * 10000 / (f["avg_us"] * 1e-6) / HBM_PEAK)
    big = json.load(open(os.path.join(d, "large_launches.json")))["durations_ns"]
    avg = sum(big) / len(big)
    large = kernel_stats(os.path.join(d, "kernel_stats_large_2M.csv"))
    name_2m = next(k for k, v in large.items() if k.startswith("k_fused_lean") and v["max_us"] > 1000.0)
    s["fast_kernel_2M"] = {"kernel": name_2m, "frames": 2000000, "launches": len(big), "avg_ms": avg / 1e6, "min_ms": min(big) / 1e6,
                           "max_ms": max(big) / 1e6, "hbm_frac": BYTES_PER_FRAME * 2000000 / (avg * 1e-9) / HBM_PEAK,
                           "joints_per_s": 2000000 * J / (avg * 1e-9)}
    t = json.load(open(os.path.join(d, "pmc_traffic.json")))
    s["traffic"] = {"read_MB": t["hbm_read_bytes_per_launch"] / 1e6, "write_MB": t["hbm_write_bytes_per_launch"] / 1e6,
                    "total_MB": t["hbm_bytes_per_launch"] / 1e6, "algorithmic_MB": t["algorithmic_bytes_per_launch"] / 1e6,
                    "ratio": t["hbm_bytes_per_launch"] / t["algorithmic_bytes_per_launch"], "source_sha256": t["source_sha256"]}
    if "valu" in t:
        s["valu_per_64_joints"] = t["valu"]["per_64_joints"]
    c = json.load(open(os.path.join(d, "counters.json")))
    for kname, line in c.get("pmc_sq", {}).items():
        if "k_fused_lean" in kname and line.get("SQ_WAVES", 0) > 4096:       # the 2 000 000-frame launch
            s["fast_kernel_2M_counters"] = {"valu_busy": line["SQ_ACTIVE_INST_VALU"] * 4.0 / (line["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
                                            "clock_GHz": line["GRBM_GUI_ACTIVE"] / 8.0 / (avg * 1e-9) / 1e9,
                                            "SQ_INSTS_VALU": line["SQ_INSTS_VALU"]}
    s["multi"] = {}
    for cfg, fn, frames, kc in (("cfg3", "kernel_stats_multi_cfg3_8x4_10000.csv", 10000, 448), ("cfg5", "kernel_stats_multi_cfg5_16x8_12000.csv", 12000, 7680)):
        ks = kernel_stats(os.path.join(d, fn))
        runs = 4      # scripts/bench_configs.py: one warm-up + three timed fused calls (each = one launch of every kernel per segment)
        per_call_ms = sum(v["total_ms"] for v in ks.values()) / runs
        s["multi"][cfg] = {"frames": frames, "fused_calls_in_trace": runs, "kernels": ks, "sum_of_kernels_ms_per_call": per_call_ms,
                           "frames_per_s": frames / (per_call_ms * 1e-3),
                           "fp64_frac": frames * kc * J * FLOP_PER_SOLVE / (per_call_ms * 1e-3) / FP64_PEAK}
    mc = multi_counters(os.path.join(d, "multi_person_counters.txt"))
    for cfg in mc:
        s["multi"].setdefault(cfg, {})["counters"] = mc[cfg]
    s["bench_default"] = bench_digest(open(os.path.join(d, "bench_default.json")).read())
    s["bench_steps20"] = bench_digest(open(os.path.join(d, "bench_steps20.json")).read())
    # the driver's own record of bench.py (copied beside the evidence by collect_profiles.sh: the newest BENCH_r*.json then)
    drv = os.path.join(d, "driver_bench.json")
    if os.path.exists(drv):
        j = json.load(open(drv))
        if j.get("parsed"):
            s["driver"] = dict(bench_digest(json.dumps(j["parsed"])), file=j["file"], head=j.get("head"))
    nr = os.path.join(d, "next_rows.jsonl")      # rows N1 / N2 / N4 and the whole pipeline (scripts/bench_next_rows.py)
    if os.path.exists(nr):
        s["next_rows"] = {}
        for ln in open(nr):
            if ln.startswith("{"):
                r = json.loads(ln)
                s["next_rows"][r["row"]] = {k: v for k, v in r.items() if k != "row"}
    mf = os.path.join(d, "multiproc_full.jsonl")  # tests/test_gpu_multiproc.py: BASELINE configs[3] / [4] at full size, 8 ranks on the one GPU
    if os.path.exists(mf):
        s["multiproc_full"] = []
        for ln in open(mf):
            if ln.startswith("{"):
                r = json.loads(ln)
                ranks = r["ranks"]
                ent = {"config": r["config"], "world": r["world"], "frames_total": ranks[0]["frames_total"], "pout_max": ranks[0]["pout_max"]}
                for kind in ("padded", "compact"):
                    rs = [q[kind] for q in ranks if kind in q]
                    if rs:
                        ent[kind] = {"run_wall_s_max": max(x["run_wall_s"] for x in rs), "kernels_only_wall_s_max": max(x["kernels_only_wall_s"] for x in rs),
                                     "pieces": rs[0]["pieces"], "gather_MB_received_per_rank": rs[0]["gather_bytes_received"] / 1e6}
                s["multiproc_full"].append(ent)
    pw = os.path.join(d, "power_trace.json")
    if os.path.exists(pw):
        j = json.load(open(pw))
        cap = j.get("power_cap") or {}
        s["power"] = {"cap_W": (cap.get("power_cap") or 0) / 1e6 if isinstance(cap, dict) else None,
                      "workloads": {w["workload"]: {"gfxclk_MHz": (w.get("gfxclk_MHz") or {}).get("median"),
                                                    "socket_power_W": (w.get("socket_power_W") or {}).get("median"),
                                                    "achieved": w.get("achieved")} for w in j["workloads"]}}
    return s


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    out = build(tag)
    json.dump(out, open(os.path.join(ROOT, "profiles", tag, "summary.json"), "w"), indent=1, sort_keys=True)
    print("profiles/%s/summary.json written" % tag)

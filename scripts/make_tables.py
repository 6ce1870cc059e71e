#!/usr/bin/env python3
"""The measured tables of README.md, DESIGN.md and profiles/README.md, rendered from profiles/<tag>/summary.json.

    python scripts/make_tables.py --write      rewrite the blocks between `<!-- BEGIN measured:... -->` / `<!-- END ... -->`
    python scripts/make_tables.py --check      exit 1 if a document's block differs from what summary.json renders to

tests/test_docs_tables.py runs the check, so a number in a table cannot drift from the committed evidence.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r05"
DOCS = {"README.md": ["headline"], "DESIGN.md": ["headline", "detail"], os.path.join("profiles", "README.md"): ["detail"]}


def load(tag=TAG):
    return json.load(open(os.path.join(ROOT, "profiles", tag, "summary.json")))


def e(x, nd=2):
    """1.23e10 style"""
    return ("%." + str(nd) + "e") % x


def headline(s):
    """Three columns: the quantity, this round's figure (profiles/<tag>/, our boxes), and the same quantity in the newest
    record of the DRIVER's own run of bench.py that existed when the evidence was collected (profiles/<tag>/driver_bench.json:
    an independently run figure; it belongs to the sources of the round it names)."""
    b, b20 = s["bench_default"], s["bench_steps20"]
    k10, k2m = s["fast_kernel_10k"], s["fast_kernel_2M"]
    m3, m5 = s["multi"]["cfg3"], s["multi"]["cfg5"]
    d = s.get("driver")
    dx = (d or {}).get("extra") or []
    pw = (s.get("power") or {}).get("workloads", {})

    def drv(fn, *need):
        try:
            return fn(d) if d else "-"
        except (KeyError, IndexError, TypeError):
            return "-"

    def fb(x):
        f = x.get("fall_backs") or {}
        return "fall-back frames %d / %d / %d" % (f.get("second_association_launch", -1), f.get("exact_candidate_sums", -1), f.get("k_frame_recompute", -1))

    def clk(name):
        w = pw.get(name) or {}
        return "%.0f MHz at %.0f W" % (w.get("gfxclk_MHz") or 0, w.get("socket_power_W") or 0)

    rows = [
        ("`value`, BASELINE configs[1] (10 000 frames per call; a plain loop of calls in the library's overlap mode, 2 internal streams; median of 7 regions)",
         "%s joints/s (`python bench.py`), %s (`--steps 20 --warmup 5`, the driver's command); regions %.4f-%.4f ms per step"
         % (e(b["value"]), e(b20["value"]), b["ms_per_step_min"], b["ms_per_step_max"]),
         drv(lambda d: "%s joints/s (two contexts alternated by hand)" % e(d["value"]))),
        ("`roofline` (one stream, launches back to back): `%s`" % k10["kernel"],
         "%.2f us per launch by rocprofv3 (%d launches of the one-stream command) -> %.0f GB/s = **%.3f of 8 TB/s**; HIP events of `bench.py`: "
         "%.2f us attached to the dispatch (`roofline.frac` %.3f / %.3f in the two bench lines), %.2f us launch to launch on one stream "
         "(%.3f), %.2f us bracketed by event records (%.3f)"
         % (k10["avg_us"], k10["calls"], 85.12e6 / (k10["avg_us"] * 1e-6) / 1e9, k10["hbm_frac"], b["kernel_us_mean"], b["roofline_frac"],
            b20["roofline_frac"], b["kernel_us_step_one_stream"], b["frac_step_one_stream"], b["kernel_us_bracketed"], b["frac_bracketed"]),
         drv(lambda d: "%.2f us attached -> `roofline.frac` %.3f" % (d["kernel_us_mean"], d["roofline_frac"]))),
        ("the same against the device-copy bandwidth measured on the box (1 GiB `copy_`, read + written bytes)",
         "copy %.0f GB/s -> the 10 000-frame launch runs at %.3f of it" % (b.get("copy_GBs") or 0.0, b.get("frac_of_copy") or 0.0),
         drv(lambda d: "%.3f of %.0f GB/s" % (d["frac_of_copy"], d["copy_GBs"]))),
        ("`roofline_region` (consecutive launches overlap)", "%.3f / %.3f of 8 TB/s" % (b["roofline_region_frac"], b20["roofline_region_frac"]),
         drv(lambda d: "%.3f" % d["roofline_region_frac"])),
        ("one 2 000 000-frame launch (SURVEY 8d's roofline run): `%s`" % k2m["kernel"],
         "%.2f ms (rocprofv3, %d launches: %.2f-%.2f) -> %s joints/s, **%.3f of 8 TB/s**; `large_batch.frac` %.3f"
         % (k2m["avg_ms"], k2m["launches"], k2m["min_ms"], k2m["max_ms"], e(k2m["joints_per_s"]), k2m["hbm_frac"], b["large"]["frac"]),
         drv(lambda d: "`large_batch.frac` %.3f" % d["large"]["frac"])),
        ("HBM traffic per 10 000-frame launch (PMC, calibrated in the same pass)",
         "%.2f MB read + %.2f MB written = %.2f MB against %.2f MB algorithmic (x %.3f)"
         % (s["traffic"]["read_MB"], s["traffic"]["write_MB"], s["traffic"]["total_MB"], s["traffic"]["algorithmic_MB"], s["traffic"]["ratio"]), "-"),
        ("VALU wave-instructions per 64 joints (`SQ_INSTS_VALU`, 2 000 000 frames)",
         "%.0f; VALU busy %.0f %%; `GRBM_GUI_ACTIVE` / duration = %.2f GHz" % (s["valu_per_64_joints"], 100 * s["fast_kernel_2M_counters"]["valu_busy"], s["fast_kernel_2M_counters"]["clock_GHz"]), "-"),
        ("shader clock and socket power WHILE the kernel runs (amdsmi every 4 ms over 4 s, `power_trace.json`; cap %.0f W)" % ((s.get("power") or {}).get("cap_W") or 0),
         "2 000 000-frame launches: %s; 10 000-frame launches on two streams: %s; `v_fma_f64` loop: %s; 8 x 4: %s; 16 x 8: %s"
         % (clk("lean_2M"), clk("lean_10k_two_streams"), clk("fp64_fma"), clk("multi_8x4"), clk("multi_16x8")), "-"),
        ("multi-person, 8 cameras x 4 persons, 10 000 frames (BASELINE configs[2]), ONE call",
         "%s frames/s (%.3f ms), %s pair solves/s = **%.3f of the fp64 vector peak** (`extra_workloads[0]`), %s; sum of the kernels under rocprofv3 (one stream) %.3f ms"
         % (e(b["extra"][0]["frames_per_s"]), b["extra"][0]["kernel_ms"], e(b["extra"][0]["pair_solves_per_s"]), b["extra"][0]["frac"], fb(b["extra"][0]), m3["sum_of_kernels_ms_per_call"]),
         drv(lambda d: "%s frames/s (%.3f); two calls in flight %s" % (e(dx[0]["frames_per_s"]), dx[0]["frac"], e(dx[0]["two_streams_frames_per_s"])))),
        ("multi-person, 16 x 8, 12 500 frames (one GPU's share of configs[4]), ONE call",
         "%s frames/s (%.2f ms), %s pair solves/s = **%.3f of the fp64 vector peak** (`extra_workloads[1]`), %s; 12 000 frames under rocprofv3: %.2f ms"
         % (e(b["extra"][1]["frames_per_s"]), b["extra"][1]["kernel_ms"], e(b["extra"][1]["pair_solves_per_s"]), b["extra"][1]["frac"], fb(b["extra"][1]), m5["sum_of_kernels_ms_per_call"]),
         drv(lambda d: "%s frames/s (%.3f); two calls in flight %s" % (e(dx[1]["frames_per_s"]), dx[1]["frac"], e(dx[1]["two_streams_frames_per_s"])))),
        ("8 x 4 x 10 000 frames with float64 outputs (the reference's output type), ONE call",
         ("%s frames/s (%.3f ms; float32: %.3f ms), same route" % (e(b["extra"][2]["frames_per_s"]), b["extra"][2]["kernel_ms"], b["extra"][0]["kernel_ms"]))
         if len(b["extra"]) > 2 else "-", "- (k_frame_recompute then)"),
        ("one detection per camera, other shapes, 10 000 frames per call (`extra_workloads[3..]`; fp64-vector roof against the reference's 90 flop per pair solve + 15 per ray)",
         "; ".join("%s: %.1f us = %s joints/s, **%.3f**" % (x["workload"].split(" x 133")[0].replace(" cameras x 1 person", " x 1") + (" float64 out" if "double" in x["kernel"] or "k_associate" in x["kernel"] else ""),
                                                        x["kernel_ms"] * 1e3, e(x["frames_per_s"] * 133), x["frac"]) for x in b["extra"][3:]) or "-", "-"),
        ("per-frame API (`main.py:50-71,106`, floor rig, 300 frames one by one)",
         "%.0f us per frame through the reference-named calls, %.0f us as one F = 1 fused host call (reference: %.1f ms per frame)"
         % (b["per_frame"]["api_sequence_us"], b["per_frame"]["fused_host_call_us"], b["per_frame"]["reference_ms"]),
         drv(lambda d: "%.0f us / %.0f us" % (d["per_frame"]["api_sequence_us"], d["per_frame"]["fused_host_call_us"]))),
        ("CPU oracle on the box (OpenMP, %d threads)" % b["cpu"]["cores"],
         "%s joints/s; GPU batch vs oracle %.1e m" % (e(b["cpu"]["value"]), b["cpu"]["gpu_vs_oracle_max_abs_m"]),
         drv(lambda d: "%s joints/s" % e(d["cpu"]["value"]))),
    ]
    out = ["| Quantity | Round 5 (`profiles/%s/`, our boxes) | Driver's record `%s` (round-%s sources) |"
           % (s["tag"], (d or {}).get("file", "-"), ((d or {}).get("file", "BENCH_r??")[7:9])), "|---|---|---|"]
    out += ["| %s | %s | %s |" % r for r in rows]
    return "\n".join(out)


def detail(s):
    out = ["| Config | Kernel | calls | avg us | share | VALU busy | LDS conflict cycles / active LDS cycles |", "|---|---|---|---|---|---|---|"]
    for cfg, label in (("cfg3", "8 x 4, 10 000 frames"), ("cfg5", "16 x 8, 12 000 frames")):
        m = s["multi"][cfg]
        tot = sum(k["total_ms"] for k in m["kernels"].values())
        for name, k in sorted(m["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
            c = m.get("counters", {}).get(name, {})
            out.append("| %s | `%s` | %d | %.1f | %.1f %% | %s | %s |" % (
                label, name, k["calls"], k["avg_us"], 100 * k["total_ms"] / tot,
                ("%.0f %%" % (100 * c["valu_busy"])) if "valu_busy" in c else "-",
                ("%.3f" % c["lds_conflict_ratio"]) if "lds_conflict_ratio" in c else "-"))
    return "\n".join(out)


RENDER = {"headline": headline, "detail": detail}


def blocks(s):
    return {name: "<!-- BEGIN measured:%s (generated by scripts/make_tables.py from profiles/%s/summary.json: do not edit) -->\n%s\n<!-- END measured:%s -->"
                  % (name, s["tag"], fn(s), name) for name, fn in RENDER.items()}


def main():
    s = load()
    want = blocks(s)
    bad = []
    for doc, names in DOCS.items():
        path = os.path.join(ROOT, doc)
        text = open(path).read()
        new = text
        for name in names:
            rx = re.compile(r"<!-- BEGIN measured:%s .*?<!-- END measured:%s -->" % (name, name), re.S)
            if not rx.search(new):
                bad.append("%s: no block `measured:%s`" % (doc, name))
                continue
            new = rx.sub(lambda m: want[name], new)
        if new != text:
            if "--write" in sys.argv:
                open(path, "w").write(new)
                print("updated", doc)
            else:
                bad.append("%s: a measured table differs from profiles/%s/summary.json (run scripts/make_tables.py --write)" % (doc, s["tag"]))
    if bad:
        print("\n".join(bad))
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())

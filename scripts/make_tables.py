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
TAG = "r06"
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
    ex = b.get("extra") or []

    def find(*needles, method="pairwise", zero_fill=True, lines=None):
        for x in (ex if lines is None else lines):
            if all(n in x["workload"] for n in needles) and x.get("method", "pairwise") == method and x.get("zero_fill", True) == zero_fill:
                return x
        return None
    x_f32, x_168, x_f64 = find("configs[2]: 8 cameras"), find("configs[4]"), find("configs[2] with float64")
    x_nzf, x_dlt84 = find("NO_ZERO_FILL", zero_fill=False), find("DLT", "configs[2]", method="dlt")
    singles = [x for x in ex if " x 1 person" in x["workload"] and x.get("method", "pairwise") == "pairwise" and "configs[" not in x["workload"]]
    dlts = [x for x in ex if " x 1 person" in x["workload"] and x.get("method") == "dlt"]
    dx0, dx1 = find("configs[2]: 8 cameras", lines=dx), find("configs[4]", lines=dx)
    nrw = s.get("next_rows") or {}

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
         "%s frames/s (%.3f ms), %s pair solves/s = **%.3f of the fp64 vector peak** (nominal: the reference's 90 flop per candidate joint), %s; sum of the kernels under rocprofv3 (one stream) %.3f ms"
         % (e(x_f32["frames_per_s"]), x_f32["kernel_ms"], e(x_f32["pair_solves_per_s"]), x_f32["frac"], fb(x_f32), m3["sum_of_kernels_ms_per_call"]),
         drv(lambda d: "%s frames/s (%.3f); two calls in flight %s" % (e(dx0["frames_per_s"]), dx0["frac"], e(dx0["two_streams_frames_per_s"])))),
        ("multi-person, 16 x 8, 12 500 frames (one GPU's share of configs[4]), ONE call",
         "%s frames/s (%.2f ms), %s pair solves/s = **%.3f of the fp64 vector peak** (nominal), %s; 12 000 frames under rocprofv3: %.2f ms"
         % (e(x_168["frames_per_s"]), x_168["kernel_ms"], e(x_168["pair_solves_per_s"]), x_168["frac"], fb(x_168), m5["sum_of_kernels_ms_per_call"]),
         drv(lambda d: "%s frames/s (%.3f); two calls in flight %s" % (e(dx1["frames_per_s"]), dx1["frac"], e(dx1["two_streams_frames_per_s"])))),
        ("8 x 4 x 10 000 frames with float64 outputs (the reference's output type), ONE call",
         ("%s frames/s (%.3f ms; float32: %.3f ms), same route" % (e(x_f64["frames_per_s"]), x_f64["kernel_ms"], x_f32["kernel_ms"])) if x_f64 else "-",
         drv(lambda d: "%s frames/s" % e(find("configs[2] with float64", lines=dx)["frames_per_s"]))),
        ("8 x 4 x 10 000 frames, `Pout_max` 16, with `SNOWTRI_CALL_NO_ZERO_FILL` (the slots behind `out_count[f]` are not written)",
         ("%s frames/s (%.3f ms); the call writes %.0f MB of joints and scores instead of %.0f MB; two calls in flight %s frames/s"
          % (e(x_nzf["frames_per_s"]), x_nzf["kernel_ms"], x_nzf["written_bytes_per_frame"] * x_nzf["frames"] / 1e6, x_nzf["pout_max"] * (133 * 16 + 4) * x_nzf["frames"] / 1e6,
             e(x_nzf["two_streams_frames_per_s"]))) if x_nzf else "-", "-"),
        ("one detection per camera, other shapes, 10 000 frames per call (fp64-vector roof against the reference's 90 flop per pair solve + 15 per ray)",
         "; ".join("%s: %.1f us = %s joints/s, **%.3f**" % (x["workload"].split(" x 133")[0].replace(" cameras x 1 person", " x 1") + (" float64 out" if "double" in x["kernel"] or "k_associate" in x["kernel"] else ""),
                                                        x["kernel_ms"] * 1e3, e(x["frames_per_s"] * 133), x["frac"]) for x in singles) or "-", "-"),
        ("`method = SNOWTRI_DLT` (north_star's N-view DLT; NOT the reference's algorithm), 10 000 frames per call; roofs: HBM (64 / 112 B per joint) and fp64 vector against the stated 64 C + 310 flop per joint",
         ("; ".join("%s: %.1f us = %s joints/s, %.3f of 8 TB/s, %.3f of the fp64 peak" % (x["workload"].split(": ")[1].split(" x 133")[0].replace(" cameras x 1 person", " x 1"), x["kernel_ms"] * 1e3,
                                                                                        e(x["joints_per_s"]), x.get("hbm_frac") or 0.0, x["frac"]) for x in dlts)
          + ("; 8 x 4 behind the reference's association (DLT per cluster): %s frames/s (%.3f ms)" % (e(x_dlt84["frames_per_s"]), x_dlt84["kernel_ms"]) if x_dlt84 else "")) if dlts else "-", "-"),
        ("rows after the path, device-resident, 100 000 frames (`next_rows.jsonl`): N1 smoothing of 399 lanes / N2 per-bone smoothing of 4 persons / the whole chain A1-A4 + N1 + N2",
         ("N1 %.3f ms = %.0f GB/s algorithmic = **%.3f of 8 TB/s** (one pass: 16 B per lane-frame); N2 %.3f ms = %.0f GB/s (%.3f); `TrackPipeline` %.3f ms (%.3f ms from raw-frame detections)"
          % (nrw["N1 smooth_track"]["ms"], nrw["N1 smooth_track"]["algorithmic_GBs"], nrw["N1 smooth_track"]["algorithmic_GBs"] / 8000.0,
             nrw["N2 blender_smooth"]["ms"], nrw["N2 blender_smooth"]["algorithmic_GBs"], nrw["N2 blender_smooth"]["algorithmic_GBs"] / 8000.0,
             nrw["pipeline A1-A4 + N1 + N2"]["ms"], nrw["pipeline A1-A4 + N1 + N2 + N4"]["ms"])) if "N1 smooth_track" in nrw else "-", "-"),
        ("BASELINE configs[3] / configs[4] at their FULL size over 8 ranks sharing the one GPU (gloo-staged collectives; `multiproc_full.jsonl`): wall time of `ShardedTriangulator.run`, slowest rank",
         "; ".join("%s frames of %s: %s" % ("{:,}".format(m["frames_total"]).replace(",", " "), m["config"].split(":")[1].split(",")[0].strip(),
                                            ", ".join("%s gather %.2f s (%.0f MB received per rank, %d pieces; the kernels alone %.3f s)" % (k, m[k]["run_wall_s_max"], m[k]["gather_MB_received_per_rank"], m[k]["pieces"], m[k]["kernels_only_wall_s_max"])
                                                      for k in ("padded", "compact") if k in m)) for m in (s.get("multiproc_full") or [])) or "-", "-"),
        ("per-frame API (`main.py:50-71,106`, floor rig, 300 frames one by one)",
         "%.0f us per frame through the reference-named calls, %.0f us as one F = 1 fused host call (reference: %.1f ms per frame)"
         % (b["per_frame"]["api_sequence_us"], b["per_frame"]["fused_host_call_us"], b["per_frame"]["reference_ms"]),
         drv(lambda d: "%.0f us / %.0f us" % (d["per_frame"]["api_sequence_us"], d["per_frame"]["fused_host_call_us"]))),
        ("CPU oracle on the box (OpenMP, %d threads)" % b["cpu"]["cores"],
         "%s joints/s; GPU batch vs oracle %.1e m" % (e(b["cpu"]["value"]), b["cpu"]["gpu_vs_oracle_max_abs_m"]),
         drv(lambda d: "%s joints/s" % e(d["cpu"]["value"]))),
    ]
    out = ["| Quantity | Round %d (`profiles/%s/`, our boxes) | Driver's record `%s` (round-%s sources) |"
           % (int(s["tag"][1:3]), s["tag"], (d or {}).get("file", "-"), ((d or {}).get("file", "BENCH_r??")[7:9])), "|---|---|---|"]
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

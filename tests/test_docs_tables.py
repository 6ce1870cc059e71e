"""The measured tables of README.md, DESIGN.md and profiles/README.md are generated from profiles/<tag>/summary.json
(scripts/make_tables.py), and summary.json is derived from the CSVs / JSON lines committed beside it
(scripts/build_summary.py).  These tests fail when a document quotes a number the committed evidence does not hold."""
import json
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_tables_in_the_documents_equal_what_summary_json_renders_to():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_tables.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr


def test_summary_json_equals_what_the_committed_csvs_derive_to():
    import build_summary
    import make_tables
    have = make_tables.load()
    want = json.loads(json.dumps(build_summary.build(have["tag"]), sort_keys=True))
    assert have == want, "profiles/%s/summary.json is stale: run scripts/build_summary.py %s" % (have["tag"], have["tag"])


def test_headline_numbers_are_the_csv_numbers():
    """Spot checks straight from the raw files, independent of build_summary's code."""
    import csv
    import make_tables
    s = make_tables.load()
    d = os.path.join(ROOT, "profiles", s["tag"])
    rows = [r for r in csv.DictReader(open(os.path.join(d, "kernel_stats_cfg2_10k.csv"))) if "k_fused_lean_coop" in r["Name"]]
    assert len(rows) == 1
    avg_us = float(rows[0]["AverageNs"]) / 1e3
    assert abs(s["fast_kernel_10k"]["avg_us"] - avg_us) < 1e-9
    assert abs(s["fast_kernel_10k"]["hbm_frac"] - 8512 * 10000 / (avg_us * 1e-6) / 8e12) < 1e-12
    text = open(os.path.join(ROOT, "README.md")).read()
    assert "%.2f us per launch by rocprofv3" % avg_us in text and "**%.3f of 8 TB/s**" % s["fast_kernel_10k"]["hbm_frac"] in text
    big = json.load(open(os.path.join(d, "large_launches.json")))["durations_ns"]
    assert abs(s["fast_kernel_2M"]["hbm_frac"] - 8512 * 2000000 / (sum(big) / len(big) * 1e-9) / 8e12) < 1e-12
    b = json.loads(open(os.path.join(d, "bench_default.json")).read())
    assert s["bench_default"]["value"] == b["value"] and s["bench_default"]["roofline_frac"] == b["roofline"]["frac"]
    assert b["roofline"]["kernel"] == "k_fused_lean_coop<4,float,133>" == s["fast_kernel_10k"]["kernel"]
    assert b["roofline"]["traffic"] in (None, json.load(open(os.path.join(d, "pmc_traffic.json")))["hbm_bytes_per_launch"])
    # the traffic and the VALU count the bench line quotes belong to the profiled kernel sources
    t = json.load(open(os.path.join(d, "pmc_traffic.json")))
    assert t["source_sha256"] == json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["source_sha256"]
    assert abs(s["valu_per_64_joints"] - t["valu"]["SQ_INSTS_VALU_per_launch"] / (t["valu"]["frames"] * 133 / 64.0)) < 1e-9

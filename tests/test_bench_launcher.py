"""bench.py's launch logic on CPU: `python bench.py --gpus 2 --dry-run` without a launcher re-executes itself under
torch.distributed.run, both ranks rendezvous (gloo on 127.0.0.1) and rank 0 prints one JSON line naming them; started
by a launcher whose WORLD_SIZE disagrees with --gpus it refuses instead of hanging."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_self_launches_two_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=_env(),
                       capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["rccl_ranks"] == 2
    assert sorted(r["rank"] for r in d["ranks"]) == [0, 1]
    assert len({r["pid"] for r in d["ranks"]}) == 2          # two processes, one per (would-be) GPU


def test_bench_gather_leg_is_the_products_sharded_entry():
    """`with_track_allgather` goes through snowmocap_amd.sharded.gather_track_chunked (what ShardedTriangulator.run
    calls): the dry run drives bench.gather_leg over two gloo ranks with a stand-in for the kernels, for every piece
    count of a --chunks sweep, and checks the gathered track names every (rank, frame) once, in order."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--chunks", "1,3,4", "--frames", "41"],
                       env=_env(), capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert sorted(d["gather_leg"]) == ["1", "3", "4"]
    for chunks, g in d["gather_leg"].items():
        assert g["ok"] and g["frames_gathered"] == 2 * 41, (chunks, g)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "ShardedTriangulator" in src and "all_gather_into_tensor(gbuf" not in src      # no private gather in the bench


def test_bench_refuses_a_world_size_mismatch():
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_bench_single_rank_dry_run_needs_no_launcher():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], env=_env(),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["rccl_ranks"] == 1


def test_summary_is_the_last_key_and_small():
    """Round-5 review, item 3: the driver keeps the last 8 KB of bench.py's stdout; the `summary` object -- every BASELINE
    config that fits a GPU, the DLT method, per-frame API, CPU baseline -- is the LAST key of the line and stays under 3 KB.
    Driven here with a stand-in line of the real shapes (no GPU)."""
    sys.path.insert(0, ROOT)
    import bench
    multi = lambda w, **kw: dict(dict(workload=w, frames=10000, kernel_ms=1.0, frames_per_s=1e7, roofline={"frac": 0.68}, mean_persons_per_frame=5.08,
                                      two_streams={"frames_per_s": 1.05e7}, method="pairwise", zero_fill=True, pout_max=16,
                                      output_bytes_written_per_frame=34112.0), **kw)
    single = lambda w, **kw: dict(dict(workload=w, kernel="k_fused_single<4,1,float,float>", kernel_ms=0.03, joints_per_s=4e10, roofline={"frac": 0.3},
                                       hbm={"frac_of_8TBs": 0.33}, out_dtype="float32", method="pairwise"), **kw)
    line = {"metric": "joint-triangulations/sec", "value": 6.3e10, "ms_per_step": 0.0211, "n_gpus": 1, "dtype": "f64",
            "roofline": {"kernel_ms_mean": 0.0252, "frac": 0.42, "frac_of_measured_copy": 0.61, "traffic": 8.8e7, "algorithmic_bytes_per_launch": 8.512e7},
            "large_batch": {"frac": 0.52}, "cpu_baseline": {"value": 1.7e8, "cores": 16, "gpu_vs_oracle_max_abs_m": 6e-8},
            "per_frame_api": {"api_sequence_us_median": 78.0, "fused_host_call_us_median": 31.0, "what": "..."},
            "extra_workloads": [
                multi("BASELINE configs[2]: 8 cameras x 4 persons x 133 joints x 10 000 frames"),
                multi("BASELINE configs[4] per-GPU share: 16 cameras x 8 persons x 133 joints x 12 500 frames"),
                multi("BASELINE configs[2] with float64 outputs: 8 cameras x 4 persons"),
                multi("BASELINE configs[2] with SNOWTRI_CALL_NO_ZERO_FILL: 8 cameras", zero_fill=False, output_bytes_written_per_frame=10830.0),
                multi("DLT (method = SNOWTRI_DLT) on BASELINE configs[2]: 8 cameras x 4 persons", method="dlt"),
                single("4 cameras x 1 person x 133 joints x 10000 frames, the floor rig", out_dtype="float64"),
                single("6 cameras x 1 person x 133 joints"), single("8 cameras x 1 person x 133 joints"),
                single("DLT (method = SNOWTRI_DLT, NOT the reference's algorithm): 4 cameras x 1 person", method="dlt"),
                single("DLT (method = SNOWTRI_DLT, NOT the reference's algorithm): 8 cameras x 1 person", method="dlt")]}
    s = bench.summary_of(line)
    assert len(json.dumps(s)) < 3000, len(json.dumps(s))
    for key in ("configs1_4x1_10000_frames", "configs2_8x4_10000_frames_f32", "configs2_8x4_f64_out", "configs2_8x4_no_zero_fill",
                "configs4_share_16x8_12500_frames", "dlt_4x1", "dlt_8x1", "dlt_8x4_with_association", "single_4x1_f64_out", "single_6x1",
                "single_8x1", "per_frame_api_us", "cpu_baseline"):
        assert s[key] is not None, key
    assert s["configs2_8x4_no_zero_fill"]["output_MB_written_per_call"] < s["configs2_8x4_no_zero_fill"]["output_MB_written_per_call_default"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('line["summary"] = summary_of(line)') > src.index('"extra_workloads": extra,')      # appended behind everything else

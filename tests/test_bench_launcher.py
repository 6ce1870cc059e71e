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

"""bench.py's N > 1 path with the REAL kernels on a one-GPU box: `--one-device` puts both ranks on cuda:0 and runs
torch.distributed over gloo (RCCL refuses two ranks on one device).  Everything else is the code the driver's
`--gpus N` run executes: the fences, the MAX over ranks, the per-rank times, the sharded entry with its all-gather."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_device_print_one_whole_job_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    K, F = 6, 4000
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--steps", str(K), "--warmup", "2",
                        "--repeats", "3", "--frames", str(F), "--pool", "4", "--large-frames", "0", "--device-warmup-ms", "50",
                        "--no-extra", "--no-cpu-baseline", "--no-per-frame", "--chunks", "auto,2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["scaling"] == "weak"
    assert sorted(r["rank"] for r in d["ranks"]) == [0, 1]
    assert d["steps"] == K and len(d["ms_per_step_per_rank"]) == 2
    # whole-job value: the frames of BOTH ranks over the slower rank's time
    assert abs(d["value"] - 2 * F * 133 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["ms_per_step"] >= max(d["ms_per_step_per_rank"]) * (1 - 1e-9)
    g = d["with_track_allgather"]
    assert g["gathered_track_ok"] and [s["chunks"] for s in g["sweep"]] == ["auto", 2] and all(s["gathered_track_ok"] for s in g["sweep"])
    assert g["sweep"][1]["pieces"] == 2
    assert list(d)[-1] == "summary"

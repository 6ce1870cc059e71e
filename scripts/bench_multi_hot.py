#!/usr/bin/env python3
"""GPU box: the multi-person configurations (BASELINE configs[2] = 8 x 4 x 10 000 frames, one GPU's share of configs[4] =
16 x 8 x 12 500 frames) timed the way the headline kernel is timed: calls queued BACK TO BACK on one stream and their
event pairs read afterwards (a synchronize between calls lets the chip idle and clock down: round 3's `kernel_ms` of a
single synchronised call was 10 % above the same call in a loop).  Also: two calls in flight on two contexts / streams.

    python scripts/bench_multi_hot.py [--only=3|5] [--calls=N] [--out64] [--kn=K] [--center=I] [--no-two] [--pout=N] [--no-zero-fill]
    python scripts/bench_multi_hot.py --sweep-split=1,2,4      (snowtri_ctx_set_split, each value twice, interleaved)
A/B of development builds: SNOWTRI_LIB=.../ab/libsnowtri_<tag>.so python scripts/bench_multi_hot.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from snowmocap_amd import synth
from snowmocap_amd.batch import BatchTriangulator

ONLY = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--only=")]
CALLS = ([int(a.split("=")[1]) for a in sys.argv if a.startswith("--calls=")] or [0])[0]
KN = ([int(a.split("=")[1]) for a in sys.argv if a.startswith("--kn=")] or [0])[0]
CENTER = ([int(a.split("=")[1]) for a in sys.argv if a.startswith("--center=")] or [-1])[0]
OUT64 = "--out64" in sys.argv
NO_TWO = "--no-two" in sys.argv
POUT = ([int(a.split("=")[1]) for a in sys.argv if a.startswith("--pout=")] or [0])[0]
ZERO_FILL = "--no-zero-fill" not in sys.argv


def measure(cfg, F, gen, pout, calls, split=None):
    dev = torch.device("cuda", 0)
    wl = synth.config_workload(cfg, gen)
    K, R, t = wl["rig"]
    params = dict(wl["params"])
    if KN:
        params["keypoint_num"] = KN
    if CENTER >= 0:
        params["center_point_index"] = CENTER
    kp = torch.from_numpy(wl["kpts"]).to(dev).repeat(F // gen, 1, 1, 1, 1).contiguous()
    npers = torch.from_numpy(wl["n_persons"]).to(dev).repeat(F // gen, 1).contiguous()
    odt = np.float64 if OUT64 else np.float32
    bts = [BatchTriangulator(K, R, t, params, pout_max=pout, out_dtype=odt, zero_fill=ZERO_FILL) for _ in range(2)]
    if split is not None:
        for b in bts:
            b.ctx.set_split(split)
    outs = [b.alloc_outputs(F, dev) for b in bts]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    bts[0].run_torch(kp, npers, out=outs[0])
    torch.cuda.synchronize(dev)
    # warm the clocks, then `calls` calls back to back with the context's event ring around each
    for _ in range(3):
        bts[0].run_torch(kp, npers, out=outs[0])
    bts[0].ctx.set_timing(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(calls):
        bts[0].run_torch(kp, npers, out=outs[0])
    e1.record()
    torch.cuda.synchronize(dev)
    per_call = bts[0].ctx.timing_collect()
    bts[0].ctx.set_timing(False)
    loop_ms = e0.elapsed_time(e1) / calls
    # the round-3 way: one call, synchronise, read its events
    bts[0].ctx.set_timing(True)
    alone = []
    for _ in range(4):
        bts[0].run_torch(kp, npers, out=outs[0])
        alone.append(bts[0].ctx.last_kernel_ms()[0])
    bts[0].ctx.set_timing(False)
    # two calls in flight
    def two(n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(n):
            bts[i & 1].run_torch(kp, npers, out=outs[i & 1], stream=streams[i & 1].cuda_stream)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3
    if NO_TWO:      # (under rocprofv3: the kernels of two calls in flight stretch one another's durations)
        two_ms = float("nan")
    else:
        two(4)
        two_ms = float(np.median([two(2 * calls) for _ in range(3)]))
    same = bool(torch.equal(outs[0]["xyzs"], outs[1]["xyzs"])) and bool(torch.equal(outs[0]["count"], outs[1]["count"]))
    cnt = outs[0]["count"].cpu().numpy()
    res = {"cfg": cfg, "frames": F, "out": "f64" if OUT64 else "f32", "kn": params["keypoint_num"], "lib": os.environ.get("SNOWTRI_LIB", "production"),
           "pout_max": pout, "zero_fill": ZERO_FILL,
           "kernels": bts[0].ctx.last_kernel_names(),
           "ms_per_call_loop": loop_ms, "ms_per_call_events_median": float(np.median(per_call)), "ms_per_call_events_min": float(np.min(per_call)),
           "frames_per_s_loop": F / (loop_ms * 1e-3), "ms_single_synchronised_call": float(np.median(alone[1:])),
           "two_streams_ms_per_call": two_ms, "two_streams_frames_per_s": F / (two_ms * 1e-3), "two_streams_identical": same,
           "mean_persons": float(cnt.mean()), "handed": bts[0].ctx.last_handover_persons()}
    for b in bts:
        b.close()
    return res


# --sweep-split=v1,v2,...: the whole measurement once per value of snowtri_ctx_set_split, interleaved twice so that a drifting box shows
SWEEP = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--sweep-split=")]
settings = [None]
if SWEEP:
    settings = [int(v) for v in SWEEP[0].split(",")] * 2
for cfg, F, gen, pout in ((3, 10000, 1000, 16), (5, 12500, 250, 32)):
    if ONLY and cfg not in ONLY:
        continue
    for st in settings:
        r = measure(cfg, F, gen, POUT or pout, CALLS or (40 if cfg == 3 else 8), split=st)
        if st is not None:
            r["split"] = st
        print(json.dumps(r), flush=True)

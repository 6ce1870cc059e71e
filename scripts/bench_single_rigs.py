#!/usr/bin/env python3
"""GPU box: ONE detection per camera on ring rigs of 4 .. 8 cameras (the reference's shipped thresholds make every rig a
single-cluster rig, configs/snowmocap_default_config.json:13), timed like the multi-person lines of bench.py: calls queued
back to back on one stream, the context's event pair around each, median.

    python scripts/bench_single_rigs.py [--cams=6,8] [--frames=10000,100000] [--out64] [--calls=N] [--kn=K] [--dlt]
A/B of development builds: SNOWTRI_LIB=.../ab/libsnowtri_<tag>.so python scripts/bench_single_rigs.py

Roofline of a line: fp64 vector peak 78.6 TFLOP/s against the REFERENCE's work per output joint, 90 flop per pair solve
(triangulation.py:24-31,70-75) x C(C,2) + 15 flop per ray (camera.py:241-243) x C  (SURVEY.md 8d's counts).
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


def arg(name, default):
    v = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--" + name + "=")]
    return v[0] if v else default


CAMS = [int(c) for c in arg("cams", "4,6,8").split(",")]
FRAMES = [int(f) for f in arg("frames", "10000").split(",")]
CALLS = int(arg("calls", "0"))
KN = int(arg("kn", "0"))
OUT64 = "--out64" in sys.argv
FP64_PEAK = 78.6e12


def flop_per_joint(C):
    return 90.0 * (C * (C - 1) // 2) + 15.0 * C


def dlt_flop_per_joint(C):
    """Stated flop model of the DLT item (snowtri_fused.hpp: dlt_add_observation, dlt_inverse_iteration): two rows per camera into
    the upper triangle of A^T A (64 flop per observation), Cholesky of the shifted 4 x 4 (~60), four steps of inverse iteration
    (two triangular solves + normalisation, ~60 each; the loop leaves when the wave has settled: 3-5 on noisy data), dehomogenise (~10)."""
    return 64.0 * C + 60.0 + 4 * 60.0 + 10.0


def measure(C, F, out_dtype, calls=0, kn=0, gen=500, seed=7, rig="ring", device_index=0, method=0):
    """rig = "ring": C cameras on the synthetic ring; "floor": the reference's 4-camera rig (BASELINE configs[1]'s frames)."""
    dev = torch.device("cuda", device_index)
    rng = np.random.default_rng(seed + C)
    if rig == "floor":
        wl = synth.config_workload(2, gen, seed=seed)
        (K, R, t), kp, npers = wl["rig"], wl["kpts"], wl["n_persons"]
        C = K.shape[0]
    else:
        K, R, t = synth.ring_rig(C)
        X = synth.make_people(rng, gen, 1)
        kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(3.5, 8.0))
    prm = dict(synth.default_thresholds())
    if kn:
        prm["keypoint_num"] = kn
    reps = (F + gen - 1) // gen
    kpd = torch.from_numpy(kp).to(dev).repeat(reps, 1, 1, 1, 1)[:F].contiguous()
    npd = torch.from_numpy(npers).to(dev).repeat(reps, 1)[:F].contiguous()
    bt = BatchTriangulator(K, R, t, prm, pout_max=1, out_dtype=out_dtype, device=device_index, method=method)
    out = bt.alloc_outputs(F, dev)
    bt.run_torch(kpd, npd, out=out)
    torch.cuda.synchronize(dev)
    names = bt.ctx.last_kernel_names()
    fast = int(((out["flags"] & 4) != 0).sum().item())
    cnt1 = int((out["count"] == 1).sum().item())
    J = kp.shape[3]
    knn = prm["keypoint_num"]
    if not calls:
        calls = max(10, min(200, int(2e8 / (F * C * C))))
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.1:      # device warm-up as in bench.py: the set-up above left the chip idle
        for _ in range(8):
            bt.run_torch(kpd, npd, out=out)
        torch.cuda.synchronize(dev)
    bt.ctx.set_timing(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(calls):
        bt.run_torch(kpd, npd, out=out)
    e1.record()
    torch.cuda.synchronize(dev)
    per_call = np.asarray(bt.ctx.timing_collect(), dtype=np.float64)
    bt.ctx.set_timing(False)
    loop_ms = e0.elapsed_time(e1) / calls
    ms = float(np.median(per_call)) if per_call.size else loop_ms
    ms = min(ms, loop_ms) if per_call.size else ms
    joints = F * knn
    # the whole item runs for every joint < keypoint_num; rays of all J joints are never built for the others
    flops = joints * (dlt_flop_per_joint(C) if method else flop_per_joint(C))
    line = dict(workload=("DLT (method = SNOWTRI_DLT, NOT the reference's algorithm): " if method else "") +
                         f"{C} cameras x 1 person x {J} joints x {F} frames, {'the floor rig of the reference' if rig == 'floor' else 'ring rig'}, default thresholds",
                frames=F, method="dlt" if method else "pairwise",
                out_dtype=np.dtype(out_dtype).name, keypoint_num=knn, kernels=names, calls=calls,
                ms_per_call=ms, ms_per_call_loop=loop_ms, frames_per_s=F / (ms * 1e-3), joints_per_s=joints / (ms * 1e-3),
                pair_solves_per_s=joints * (C * (C - 1) // 2) / (ms * 1e-3),
                roofline=dict(bound=("fp64 VALU (stated DLT flops per joint: %d)" % dlt_flop_per_joint(C)) if method else ("fp64 VALU (reference flops per joint: %d)" % flop_per_joint(C)),
                              achieved=flops / (ms * 1e-3) / 1e12,
                              peak=FP64_PEAK / 1e12, unit="TFLOP/s", frac=flops / (ms * 1e-3) / FP64_PEAK),
                hbm=dict(bytes_per_joint=12 * C + (16 if out_dtype == np.float32 else 32),
                         achieved_GBps=joints * (12 * C + (16 if out_dtype == np.float32 else 32)) / (ms * 1e-3) / 1e9,
                         frac_of_8TBs=joints * (12 * C + (16 if out_dtype == np.float32 else 32)) / (ms * 1e-3) / 8e12),
                fast_frames=fast, one_person_frames=cnt1, overrides=bt.ctx.overrides())
    bt.close()
    return line


if __name__ == "__main__":
    for C in CAMS:
        for F in FRAMES:
            for odt in ([np.float64] if OUT64 else [np.float32, np.float64] if "--both" in sys.argv else [np.float32]):
                ln = measure(C, F, odt, CALLS, KN, method=(1 if "--dlt" in sys.argv else 0), rig=arg("rig", "ring"))
                print(json.dumps(ln))
                sys.stdout.flush()

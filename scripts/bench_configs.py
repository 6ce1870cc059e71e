#!/usr/bin/env python3
"""Dev/measurement aid (GPU box): throughput of the multi-person configs (BASELINE configs[2], [4] shapes)
through the fused entry, for both general kernels, with the oracle timed on a few frames beside it."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from snowmocap_amd import synth, _lib
from snowmocap_amd.batch import BatchTriangulator
from oracle import oracle as orc

J = 133
SIZES = ((3, 2000, 16), (5, 256, 32)) if "--small" in sys.argv else ((3, 10000, 16), (5, 2000, 32))
REPEAT = {5: 6} if "--full" in sys.argv else {}   # cfg5: 2 000 generated frames tiled to 12 000 = its per-GPU share
ONE_STREAM = "--one-stream" in sys.argv     # snowtri_ctx_set_split(1): every kernel of a call on the caller's stream (per-kernel profiles)
METHOD = _lib.DLT if "--dlt" in sys.argv else _lib.PAIRWISE
ONLY = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--only=")]
for cfg, F, pout in SIZES:
    if ONLY and cfg not in ONLY:
        continue
    wl = synth.config_workload(cfg, F)
    K, R, t = wl["rig"]
    C, P = K.shape[0], wl["kpts"].shape[2]
    dev = torch.device("cuda", 0)
    kp = torch.from_numpy(wl["kpts"]).to(dev)
    npers = torch.from_numpy(wl["n_persons"]).to(dev)
    if cfg in REPEAT:
        kp = kp.repeat(REPEAT[cfg], 1, 1, 1, 1).contiguous()
        npers = npers.repeat(REPEAT[cfg], 1).contiguous()
        F = F * REPEAT[cfg]
    for mode in ("2",):
        bt = BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=np.float32, method=METHOD)
        if ONE_STREAM:
            bt.ctx.set_split(1)
        out = bt.run_torch(kp, npers)
        torch.cuda.synchronize()
        bt.ctx.set_timing(True)
        ms = []
        for _ in range(3):
            bt.run_torch(kp, npers, out=out)
            ms.append(bt.ctx.last_kernel_ms()[0])
        handed = bt.ctx.last_handover_persons()
        cnt = out["count"].cpu().numpy()
        joints = int(cnt.clip(max=pout).sum()) * J
        kc = C * (C - 1) // 2 * P * P
        m = float(np.median(ms))
        print(json.dumps({"cfg": cfg, "frames": F, "kernel": ("recompute" if mode == "2" else "spill") + (" dlt" if METHOD == _lib.DLT else ""), "ms": m,
                          "frames_per_s": F / (m * 1e-3), "output_joints_per_s": joints / (m * 1e-3),
                          "pair_solves_per_s": F * kc * J / (m * 1e-3), "mean_persons": float(cnt.mean()), "handed_last_segment": handed}))
        bt.close()
    if "--no-oracle" in sys.argv:
        continue
    nf = 8 if cfg == 3 else 2
    t0 = time.perf_counter()
    ref = orc.triangulate_condense_batch(K, R, t, wl["kpts"][:nf], wl["n_persons"][:nf], orc.make_params(**wl["params"]), pout, nthreads=1)
    dt = time.perf_counter() - t0
    print(json.dumps({"cfg": cfg, "oracle_1thread_frames_per_s": nf / dt, "output_joints_per_s": int(ref["count"].sum()) * J / dt}))

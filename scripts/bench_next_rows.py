#!/usr/bin/env python3
"""Dev/measurement aid (GPU box): device-resident throughput of the rows after the hot path --
N1 snowtri_smooth_track, N2 snowtri_blender_points / snowtri_blender_smooth -- with HIP events on the
launch stream, next to the oracle's CPU time for the same work.  One JSON line per row."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from snowmocap_amd import _lib
from oracle import oracle as orc, blender as ob

dev = torch.device("cuda", 0)
L = _lib.lib()
ctx = _lib.scratch_context()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# N1: a 100 000-frame single-person track (133 joints x 3 axes), fp64
T, n = 100000, 399
x = torch.cumsum(torch.randn((T, n), dtype=torch.float64, device=dev) * 0.01, 0)
y = torch.empty_like(x)
ms = timed(lambda: _lib.check(L.snowtri_smooth_track(ctx.handle, T, n, x.data_ptr(), 2.5, 0.75, 0.0, 1 / 30, y.data_ptr(),
                                                     _lib.DEVICE, st), "smooth"))
xs = x[:5000].cpu().numpy()
t0 = time.perf_counter(); orc.second_order_track(xs, 2.5, 0.75, 0.0, 1 / 30); cpu = time.perf_counter() - t0
print(json.dumps({"row": "N1 smooth_track", "frames": T, "lanes": n, "ms": ms, "lane_frames_per_s": T * n / (ms * 1e-3),
                  "algorithmic_GBs": 16 * T * n / (ms * 1e-3) / 1e9, "oracle_1thread_lane_frames_per_s": xs.size / cpu}))

# N2 points: 1 000 000 skeletons of float32 [133][4] records (the fused kernel's output layout)
N = 1000000
rec = torch.randn((N, 133, 4), dtype=torch.float32, device=dev)
pts = torch.empty((N, 24, 4), dtype=torch.float64, device=dev)
val = torch.empty((N, 24), dtype=torch.uint8, device=dev)
ms = timed(lambda: _lib.check(L.snowtri_blender_points(ctx.handle, N, 133, rec.data_ptr(), _lib.F32, pts.data_ptr(),
                                                       val.data_ptr(), _lib.DEVICE, st), "points"))
small = rec[:2000, :, :3].double().cpu().numpy()
t0 = time.perf_counter(); ob.control_points_track(small); cpu = time.perf_counter() - t0
print(json.dumps({"row": "N2 blender_points", "skeletons": N, "ms": ms, "skeletons_per_s": N / (ms * 1e-3),
                  "algorithmic_GBs": (28 * 16 + 24 * 33) * N / (ms * 1e-3) / 1e9,
                  "record_GBs": (133 * 16 + 24 * 33) * N / (ms * 1e-3) / 1e9,
                  "oracle_1thread_skeletons_per_s": 2000 / cpu}))

# N2 smoothing: 100 000 frames x 4 persons of control points
T, P = 100000, 4
p4 = torch.cumsum(torch.randn((T, P, 24, 4), dtype=torch.float64, device=dev) * 0.01, 0)
v4 = (torch.rand((T, P, 24), device=dev) > 0.02).to(torch.uint8)
o4 = torch.empty_like(p4)
fzr = np.ascontiguousarray(np.tile([2.5, 0.75, 0.0], (24, 1)))
ms = timed(lambda: _lib.check(L.snowtri_blender_smooth(ctx.handle, T, P, p4.data_ptr(), v4.data_ptr(), _lib.ptr(fzr), 1 / 30,
                                                       o4.data_ptr(), _lib.DEVICE, st), "bsmooth"))
ps, vs = p4[:5000].cpu().numpy(), v4[:5000].cpu().numpy()
t0 = time.perf_counter(); ob.smooth_track(ps, vs, fzr, 1 / 30); cpu = time.perf_counter() - t0
print(json.dumps({"row": "N2 blender_smooth", "frames": T, "persons": P, "ms": ms,
                  "point_frames_per_s": T * P * 24 / (ms * 1e-3), "algorithmic_GBs": (64 + 1) * T * P * 24 / (ms * 1e-3) / 1e9,
                  "oracle_1thread_point_frames_per_s": 5000 * P * 24 / cpu}))

# N4: undistort the keypoints of 100 000 frames x 4 cameras x 133 joints (float32 [u, v, score] records), in place
from snowmocap_amd import synth
K, R, t = synth.load_rig_json()
lctx = _lib.Context(K, R, t)
lctx.set_distortion(synth.load_rig_distortion())
F = 100000
kp = torch.rand((F, 4, 1, 133, 3), dtype=torch.float32, device=dev) * torch.tensor([1280.0, 720.0, 8.0], device=dev)
ms = timed(lambda: _lib.check(L.snowtri_undistort_keypoints(lctx.handle, F, 1, 133, kp.data_ptr(), kp.data_ptr(), _lib.F32,
                                                            _lib.DEVICE, st), "undistort"), reps=5)
from oracle import undistort as ou
small = kp[:2000, 0, 0, :, :2].double().cpu().numpy()
D0 = synth.load_rig_distortion()[0]
t0 = time.perf_counter(); ou.undistort_pixels(K[0], D0, small, iters=5); cpu = time.perf_counter() - t0
nobs = F * 4 * 133
print(json.dumps({"row": "N4 undistort_keypoints", "frames": F, "observations": nobs, "ms": ms,
                  "observations_per_s": nobs / (ms * 1e-3), "algorithmic_GBs": 24 * nobs / (ms * 1e-3) / 1e9,
                  "oracle_numpy_1thread_observations_per_s": small.shape[0] * small.shape[1] / cpu}))

# the whole-recording pipeline: 100 000 frames, 4 cameras, 1 person, float32 detections resident in HBM
from snowmocap_amd import TrackPipeline
th = synth.default_thresholds()
smo = {n: [2.5, 0.75, 0.0] for n in __import__("snowmocap_amd.blender", fromlist=["x"]).CONTROL_POINT_NAMES}
wl = synth.config_workload(2, 10000)
kp = torch.from_numpy(wl["kpts"]).to(dev).repeat(10, 1, 1, 1, 1).contiguous()
for D, tag in ((None, "undistorted detections"), (synth.load_rig_distortion() * 0.0 + synth.load_rig_distortion(), "raw-frame detections + lens")):
    pipe = TrackPipeline(K, R, t, th, smo, n_persons_out=1, D=D)
    ms = timed(lambda: pipe.run(kp, check=False), reps=5)
    print(json.dumps({"row": "pipeline A1-A4 + N1 + N2" + (" + N4" if D is not None else ""), "input": tag, "frames": kp.shape[0],
                      "ms": ms, "frames_per_s": kp.shape[0] / (ms * 1e-3), "joints_per_s": kp.shape[0] * 133 / (ms * 1e-3)}))
    pipe.close()

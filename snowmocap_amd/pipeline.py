"""Additive whole-recording form of main.py:47-106: every stage after 2D detection on the GPU, device-resident.

    [undistort raw-frame keypoints (N4)] -> triangulate + condense (A1-A4, the hot path) -> temporal smoothing (N1)
    -> Blender control points (N2) -> per-bone smoothing (N2) -> the reference's JSON track

One call per recording instead of one Python iteration per frame.  The per-frame protocol of the reference
identifies persons by list index and lets the list length vary from frame to frame; a track needs a fixed set of
persons, so `run` requires every frame to resolve to exactly `n_persons_out` persons (the shipped configuration,
condense_distance_tol = 10 m, always yields one) and raises otherwise.  PyTorch is used for device memory and the
stream only.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _lib
from .batch import BatchTriangulator
from .blender import CONTROL_POINT_NAMES, _WIDTH


class TrackPipeline:
    def __init__(self, K, R, t, thresholds, blender_smooth_profile, n_persons_out=1, D=None, device=0,
                 method=_lib.PAIRWISE):
        """thresholds: the dict of configs/snowmocap_default_config.json (8 triangulation keys + smooth_f, smooth_z,
        smooth_r, smooth_delta_time); blender_smooth_profile: {control point: [f, z, r]}."""
        self.th = dict(thresholds)
        self.P = int(n_persons_out)
        self.bt = BatchTriangulator(K, R, t, {k: self.th[k] for k in (
            "keypoint_score_threshold", "average_score_threshold", "distance_threshold", "condense_distance_tol",
            "condense_person_num_tol", "condense_score_tol", "center_point_index", "keypoint_num")},
            pout_max=self.P, out_dtype=np.float64, device=device, method=method, D=D)
        self.kn = self.bt.params.keypoint_num
        self.fzr = np.ascontiguousarray([blender_smooth_profile[n] for n in CONTROL_POINT_NAMES], dtype=np.float64)
        self.device = device

    def close(self):
        self.bt.close()

    def run(self, kpts, n_persons=None, check=True):
        """kpts [F, C, Pmax, J, 3] (NumPy or CUDA tensor; raw-frame pixels if D was given) ->
        dict of CUDA tensors: xyzs [F, P, kn, 4] (triangulated), smoothed [F, P, kn, 4], points [F, P, 24, 4],
        valid [F, P, 24], points_smoothed [F, P, 24, 4], count [F], flags [F]."""
        import torch
        dev = torch.device("cuda", self.device)
        if not torch.is_tensor(kpts):
            kpts = torch.from_numpy(np.ascontiguousarray(kpts)).to(dev)
        if n_persons is not None and not torch.is_tensor(n_persons):
            n_persons = torch.from_numpy(np.ascontiguousarray(n_persons, dtype=np.int32)).to(dev)
        F = kpts.shape[0]
        L, h = _lib.lib(), self.bt.ctx.handle
        st = ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        tri = self.bt.run_torch(kpts, n_persons)
        if check:
            cnt = tri["count"].cpu().numpy()
            flg = tri["flags"].cpu().numpy()
            if (flg & _lib.FLAG_SINGULAR).any():    # the reference's np.linalg.inv raises (triangulation.py:26)
                raise np.linalg.LinAlgError(f"Singular matrix (frame {int(np.argmax((flg & _lib.FLAG_SINGULAR) != 0))})")
            if not (cnt == self.P).all():
                bad = int(np.argmax(cnt != self.P))
                raise ValueError(f"frame {bad} resolved to {int(cnt[bad])} persons, the track is built for {self.P}")
        xyzs = tri["xyzs"]
        n = self.P * self.kn * 4
        sm = torch.empty_like(xyzs)
        th = self.th
        _lib.check(L.snowtri_smooth_track(h, F, n, ct.c_void_p(xyzs.data_ptr()), float(th["smooth_f"]),
                                          float(th["smooth_z"]), float(th["smooth_r"]), float(th["smooth_delta_time"]),
                                          ct.c_void_p(sm.data_ptr()), _lib.DEVICE, st), "snowtri_smooth_track")
        sm[..., 3] = xyzs[..., 3]                      # only the points are filtered (triangulation.py:169-184)
        pts = torch.empty((F, self.P, 24, 4), dtype=torch.float64, device=dev)
        val = torch.empty((F, self.P, 24), dtype=torch.uint8, device=dev)
        _lib.check(L.snowtri_blender_points(h, F * self.P, self.kn, ct.c_void_p(sm.data_ptr()), _lib.F64,
                                            ct.c_void_p(pts.data_ptr()), ct.c_void_p(val.data_ptr()), _lib.DEVICE, st),
                   "snowtri_blender_points")
        pts_s = torch.empty_like(pts)
        _lib.check(L.snowtri_blender_smooth(h, F, self.P, ct.c_void_p(pts.data_ptr()), ct.c_void_p(val.data_ptr()),
                                            _lib.ptr(self.fzr), float(th["smooth_delta_time"]),
                                            ct.c_void_p(pts_s.data_ptr()), _lib.DEVICE, st), "snowtri_blender_smooth")
        return dict(xyzs=xyzs, smoothed=sm, points=pts, valid=val, points_smoothed=pts_s, count=tri["count"],
                    flags=tri["flags"])

    @staticmethod
    def to_blender_result(points_smoothed, valid, armature_profile=None):
        """Device (or NumPy) track -> the list the reference dumps with save_blender_result (blender.py:180-187):
        one {'armature': [per person {name: list}], 'score': [per person {name: 0/1}]} per frame."""
        pts = points_smoothed.cpu().numpy() if hasattr(points_smoothed, "cpu") else np.asarray(points_smoothed)
        val = valid.cpu().numpy() if hasattr(valid, "cpu") else np.asarray(valid)
        if not val[..., 1].all():
            raise np.linalg.LinAlgError("SVD did not converge")     # the reference raises on a NaN pelvis matrix
        names = list(armature_profile.keys()) if armature_profile is not None else list(CONTROL_POINT_NAMES)
        slot = {n: i for i, n in enumerate(CONTROL_POINT_NAMES)}
        frames = []
        for f in range(pts.shape[0]):
            frames.append({
                "armature": [{n: pts[f, p, slot[n], :_WIDTH[n]].tolist() for n in names} for p in range(pts.shape[1])],
                "score": [{n: int(val[f, p, slot[n]]) for n in names} for p in range(pts.shape[1])]})
        return frames

"""Rig + per-frame detection state behind the reference's `CameraGroup` API (snowvision/camera.py).

Mirrors what the triangulation path touches: `Camera` rig fields (camera.py:17-59),
`CameraGroup(camera_group_info_path=...)` (camera.py:142-170), `add_human_2D_points`
(camera.py:234-253) and `clear_2D_points` (camera.py:255-261).  Capture / recording / chessboard
calibration are device I/O outside the path and are not provided.

Difference in mechanism, not in behaviour: the reference turns every keypoint into a world ray on
the CPU at add-time (one 3x3 inverse per keypoint).  Here add-time only records the detector's
arrays; rays are built inside the HIP kernels from M_c = R_c inv(K_c) held by the native context.
`Camera.hrnet_point_rays` is still available and is computed on the GPU on first access.
"""
from __future__ import annotations

import json

import numpy as np

from . import _lib


class Camera:
    def __init__(self, cap_id=0, frame_width=1280, frame_height=720, camera_info_path=None,
                 camera_info_dict=None):
        self.cap_id, self.frame_width, self.frame_height = cap_id, frame_width, frame_height
        self.K = np.zeros((3, 3))
        self.R = np.eye(3)
        self.t = np.zeros((3, 1))
        self.D = np.zeros((1, 5))
        self._group = None
        self._index = -1
        self._reset_frame_state()
        if camera_info_path is not None:
            with open(camera_info_path, "r") as fh:
                camera_info_dict = json.load(fh)
        if camera_info_dict is not None:
            for key in ("cap_id", "frame_width", "frame_height"):
                setattr(self, key, camera_info_dict[key])
            for key in ("K", "R", "t", "D"):
                setattr(self, key, np.array(camera_info_dict[key]))

    def _reset_frame_state(self):
        self.points, self.point_rays = [], []
        self.hrnet_points, self.hrnet_point_score = [], []
        self._rays_cache = None

    @property
    def hrnet_point_rays(self):
        """Per person, per joint (3,1) fp64 world rays R.inv(K).[u,v,1] (camera.py:241-247)."""
        if self._rays_cache is None or len(self._rays_cache) != len(self.hrnet_points):
            ctx = self._group.native_context()
            out = []
            for person in self.hrnet_points:
                uv = np.ascontiguousarray(np.asarray(person, dtype=np.float64)[:, :2])
                rays = np.empty((uv.shape[0], 3))
                _lib.check(ctx.L.snowtri_rays_from_pixels(ctx.handle, self._index, uv.shape[0],
                                                               _lib.ptr(uv), _lib.ptr(rays)),
                           "snowtri_rays_from_pixels")
                out.append([rays[j].reshape(3, 1) for j in range(rays.shape[0])])
            self._rays_cache = out
        return self._rays_cache

    def camera_info_dict(self):
        return {"cap_id": self.cap_id, "frame_width": self.frame_width, "frame_height": self.frame_height,
                "K": np.asarray(self.K).tolist(), "R": np.asarray(self.R).tolist(),
                "t": np.asarray(self.t).tolist(), "D": np.asarray(self.D).tolist()}

    def save_camera_info(self, camera_info_path):
        with open(camera_info_path, "w") as fh:
            fh.write(json.dumps(self.camera_info_dict()))


class CameraGroup:
    def __init__(self, cap_ids=[0, 1], resolutions=[(1280, 720), (1280, 720)], camera_group_info_path=None):
        self.cameras = []
        if camera_group_info_path is None:
            for cap_id, (w, h) in zip(cap_ids, resolutions):
                self.cameras.append(Camera(cap_id=cap_id, frame_width=w, frame_height=h))
            self.camera_num = len(cap_ids)
        else:
            with open(camera_group_info_path, "r") as fh:
                info = json.load(fh)
            self.camera_num = info["camera_num"]
            self.cameras = [Camera(camera_info_dict=d) for d in info["camera_group_info"]]
        for i, cam in enumerate(self.cameras):
            cam._group, cam._index = self, i
        self._ctx = None
        self._ctx_key = None

    # ---- rig ---------------------------------------------------------------------------------
    def rig_arrays(self):
        """K[C,3,3], R[C,3,3], t[C,3] as contiguous fp64."""
        n = self.camera_num
        K = np.stack([np.asarray(c.K, dtype=np.float64).reshape(3, 3) for c in self.cameras[:n]])
        R = np.stack([np.asarray(c.R, dtype=np.float64).reshape(3, 3) for c in self.cameras[:n]])
        t = np.stack([np.asarray(c.t, dtype=np.float64).reshape(3) for c in self.cameras[:n]])
        return K, R, t

    def native_context(self):
        """snowtri context for the CURRENT rig values (rebuilt if K/R/t were edited)."""
        K, R, t = self.rig_arrays()
        key = (K.tobytes(), R.tobytes(), t.tobytes())
        if self._ctx is None or key != self._ctx_key:
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = _lib.Context(K, R, t)
            self._ctx_key = key
        return self._ctx

    def undistort_keypoints(self, kpts):
        """Row N4 (additive): keypoints detected on RAW frames, kpts[F, C, Pmax, J, 3] or one frame
        [C, Pmax, J, 3], moved to where `cv2.undistort(frame, K, D)` (main.py:52) would have put them, using each
        camera's K and D -- so the whole-image undistortion can be skipped.  GPU, one lane per keypoint."""
        a = np.asarray(kpts)
        single = a.ndim == 4
        ctx = self.native_context()
        n = self.camera_num
        D = np.stack([np.asarray(c.D, dtype=np.float64).reshape(-1)[:5] for c in self.cameras[:n]])
        ctx.set_distortion(D)
        out = ctx.undistort_keypoints(a[None] if single else a)
        return out[0] if single else out

    def camera_group_info_dict(self):
        return {"camera_num": self.camera_num,
                "camera_group_info": [c.camera_info_dict() for c in self.cameras]}

    def save_camera_group_info(self, camera_group_info_path):
        with open(camera_group_info_path, "w") as fh:
            fh.write(json.dumps(self.camera_group_info_dict()))

    # ---- per-frame detections ----------------------------------------------------------------
    def add_human_2D_points(self, person, scores, camera_index, ax=None):
        """Record one detected person of camera `camera_index`: person[J,2] pixels, scores[J]
        (camera.py:234-253).  Call order defines the person order of that camera."""
        cam = self.cameras[camera_index]
        cam.hrnet_points.append(person)
        cam.hrnet_point_score.append(scores)
        cam._rays_cache = None
        if ax is not None:                      # optional debug drawing (camera.py:246-248)
            t = np.asarray(cam.t).reshape(3)
            for ray in cam.hrnet_point_rays[-1]:
                a = ray.reshape(3) * 10
                ax.quiver(t[0], t[1], t[2], a[0], a[1], a[2])

    def clear_2D_points(self):
        for cam in self.cameras:
            cam._reset_frame_state()

    def pack_frame(self):
        """Current detections -> (kpts[C,Pmax,J,3], n_persons[C]) in the C-ABI layout.

        dtype: float32 only when every pixel and score array is float32 -- then the reference adds
        the two confidences of a pair in float32 (NumPy scalar arithmetic, triangulation.py:72)
        and the F32 kernels reproduce exactly that; anything else is promoted to float64."""
        C = self.camera_num
        n_persons = np.array([len(self.cameras[c].hrnet_points) for c in range(C)], dtype=np.int32)
        pmax = max(1, int(n_persons.max()) if C else 1)
        people = [(c, p, np.asarray(self.cameras[c].hrnet_points[p]), np.asarray(self.cameras[c].hrnet_point_score[p]))
                  for c in range(C) for p in range(n_persons[c])]
        if not people:
            return np.zeros((C, pmax, 1, 3), dtype=np.float32), n_persons
        J = people[0][2].shape[0]
        all_f32 = all(uv.dtype == np.float32 and sc.dtype == np.float32 for _, _, uv, sc in people)
        kpts = np.zeros((C, pmax, J, 3), dtype=np.float32 if all_f32 else np.float64)
        for c, p, uv, sc in people:
            if uv.shape[0] != J or sc.shape[0] != J:
                raise ValueError("every detection must carry the same number of keypoints")
            kpts[c, p, :, :2] = uv[:, :2]
            kpts[c, p, :, 2] = sc
        return kpts, n_persons

"""Synthetic rigs and 2D keypoint batches for parity tests and bench.py.

Nothing here is on the product path: it only manufactures *inputs* (calibrated
rigs, exact projections + pixel noise, confidence scores) shaped like the
workloads BASELINE.json names.  Camera convention follows the
reference rig file (camera.py:41-44 + configs/camera_group_floor.json): `R` is
the camera->world rotation and `t` is the camera centre in world metres, so a
world point X projects as  uv ~ K . R^T (X - t).
"""
from __future__ import annotations

import json
import os

import numpy as np

J_WHOLEBODY = 133
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
FLOOR_RIG_PATH = os.path.join(_DATA, "camera_group_floor.json")
DEFAULT_THRESHOLDS_PATH = os.path.join(_DATA, "default_thresholds.json")


def load_rig_json(path=FLOOR_RIG_PATH):
    """Read a CameraGroup JSON (camera.py:142-157 schema) -> K[C,3,3], R[C,3,3], t[C,3] fp64."""
    with open(path, "r") as fh:
        info = json.load(fh)
    cams = info["camera_group_info"]
    K = np.array([c["K"] for c in cams], dtype=np.float64)
    R = np.array([c["R"] for c in cams], dtype=np.float64)
    t = np.array([c["t"] for c in cams], dtype=np.float64).reshape(len(cams), 3)
    assert info["camera_num"] == len(cams)
    return K, R, t


def load_rig_distortion(path=FLOOR_RIG_PATH):
    """Lens coefficients of a CameraGroup JSON -> D[C,5] = (k1, k2, p1, p2, k3) (camera_group_floor.json:53-61)."""
    with open(path, "r") as fh:
        info = json.load(fh)
    return np.array([np.asarray(c["D"], dtype=np.float64).reshape(-1)[:5] for c in info["camera_group_info"]])


def default_thresholds():
    with open(DEFAULT_THRESHOLDS_PATH, "r") as fh:
        return json.load(fh)


def ring_rig(C, radius=4.5, height=2.6, look_at=(0.0, 0.0, 1.0),
             fx=690.0, fy=695.0, cx=640.0, cy=360.0):
    """C cameras evenly spaced on a circle, all looking at `look_at`."""
    K = np.zeros((C, 3, 3))
    R = np.zeros((C, 3, 3))
    t = np.zeros((C, 3))
    target = np.asarray(look_at, dtype=np.float64)
    up = np.array([0.0, 0.0, 1.0])
    for c in range(C):
        ang = 2.0 * np.pi * c / C
        centre = np.array([radius * np.cos(ang), radius * np.sin(ang), height])
        z = target - centre
        z /= np.linalg.norm(z)
        x = np.cross(z, up)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)          # image y points "down"
        R[c] = np.stack([x, y, z], axis=1)      # columns = camera axes in world
        t[c] = centre
        K[c] = [[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]]
    return K, R, t


def project(K, R, t, X):
    """X[..., 3] world points -> uv[C, ..., 2] (float64), depth[C, ...]."""
    X = np.asarray(X, dtype=np.float64)
    C = K.shape[0]
    uv = np.empty((C,) + X.shape[:-1] + (2,))
    depth = np.empty((C,) + X.shape[:-1])
    for c in range(C):
        xc = (X - t[c]) @ R[c]          # = R^T (X - t)
        pix = xc @ K[c].T
        depth[c] = pix[..., 2]
        uv[c] = pix[..., :2] / pix[..., 2:3]
    return uv, depth


def person_centres(P, circle_radius=1.5):
    if P == 1:
        return np.array([[0.0, 0.0, 0.0]])
    ang = 2.0 * np.pi * np.arange(P) / P
    return np.stack([circle_radius * np.cos(ang), circle_radius * np.sin(ang), np.zeros(P)], axis=1)


def make_people(rng, F, P, J=J_WHOLEBODY, centres=None, box=(0.6, 0.6, 1.8)):
    """World joints X[F, P, J, 3]: uniform in a box standing on the floor around each centre."""
    if centres is None:
        centres = person_centres(P)
    X = rng.uniform(0.0, 1.0, size=(F, P, J, 3))
    X[..., 0] = (X[..., 0] - 0.5) * box[0]
    X[..., 1] = (X[..., 1] - 0.5) * box[1]
    X[..., 2] = X[..., 2] * box[2]
    X += np.asarray(centres)[None, :, None, :]
    return X


def make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(3.5, 8.0),
                   permute_persons=False, dtype=np.float32):
    """Project X[F,P,J,3] into every camera -> kpts[F,C,P,J,3] = (u, v, score), n_persons[F,C].

    With `permute_persons` each (frame, camera) lists its persons in an independent
    random order, as a per-view detector would.
    """
    F, P, J, _ = X.shape
    C = K.shape[0]
    uv, _ = project(K, R, t, X)                      # [C,F,P,J,2]
    uv = np.moveaxis(uv, 0, 1)                       # [F,C,P,J,2]
    if pixel_sigma > 0:
        uv = uv + rng.normal(0.0, pixel_sigma, size=uv.shape)
    sc = rng.uniform(score_range[0], score_range[1], size=(F, C, P, J))
    if permute_persons and P > 1:
        for f in range(F):
            for c in range(C):
                perm = rng.permutation(P)
                uv[f, c] = uv[f, c, perm]
                sc[f, c] = sc[f, c, perm]
    kpts = np.concatenate([uv, sc[..., None]], axis=-1).astype(dtype)
    n_persons = np.full((F, C), P, dtype=np.int32)
    return np.ascontiguousarray(kpts), n_persons


def config_workload(cfg, F, seed=None, dtype=np.float32):
    """The BASELINE.json workloads: returns dict(rig=(K,R,t), kpts, n_persons, params, X).

    cfg 1/2/4: floor rig, 4 cams, 1 person, default thresholds.
    cfg 3    : ring rig 8 cams, 4 persons, avg_thr=1.0, ctol=0.3.
    cfg 5    : ring rig 16 cams, 8 persons, as cfg 3 plus condense_person_num_tol=30.
    """
    rng = np.random.default_rng(cfg if seed is None else seed)
    params = default_thresholds()
    if cfg in (1, 2, 4):
        K, R, t = load_rig_json()
        P = 1
        permute = False
    elif cfg == 3:
        K, R, t = ring_rig(8)
        P = 4
        params.update(average_score_threshold=1.0, condense_distance_tol=0.3)
        permute = True
    elif cfg == 5:
        K, R, t = ring_rig(16)
        P = 8
        params.update(average_score_threshold=1.0, condense_distance_tol=0.3,
                      condense_person_num_tol=30)
        permute = True
    else:
        raise ValueError("cfg must be one of 1..5")
    X = make_people(rng, F, P)
    kpts, n_persons = make_keypoints(rng, K, R, t, X, pixel_sigma=1.0,
                                     score_range=(3.5, 8.0), permute_persons=permute, dtype=dtype)
    return dict(rig=(K, R, t), kpts=kpts, n_persons=n_persons, params=params, X=X)


def config_workload_device(cfg, F, seed, device, chunk=20000, dtype=None):
    """The BASELINE.json workloads generated ON the device (torch: device memory and its Philox generator; the distributions of
    config_workload / SURVEY.md 8d -- joints uniform in a 0.6 x 0.6 x 1.8 m box around each person's centre, exact projections +
    N(0, 1 px), confidences U(3.5, 8), per-(frame, camera) person order permuted for the multi-person rigs).  For batches too
    large to build on the host and upload: a frame block of the 1 000 000-frame configs[3] / the 100 000-frame configs[4] takes
    milliseconds, and the SAME (cfg, F, seed, device type, chunk) gives the same bits in every process -- each rank of a sharded
    run generates its own block from a per-block seed, a checker regenerates any block it wants to recompute.
    Returns dict(rig=(K, R, t) NumPy, params, kpts [F, C, P, J, 3] float32 CUDA tensor, n_persons [F, C] int32 CUDA tensor)."""
    import torch
    base = config_workload(cfg, 1)
    K, R, t = base["rig"]
    C, P, J = K.shape[0], base["kpts"].shape[2], J_WHOLEBODY
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    f64 = torch.float64
    Kt, Rt, tt = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (K, R, t))
    centres = torch.from_numpy(person_centres(P)).to(dev)
    box = torch.tensor([0.6, 0.6, 1.8], dtype=f64, device=dev)
    off = torch.tensor([0.5, 0.5, 0.0], dtype=f64, device=dev)
    kpts = torch.empty((F, C, P, J, 3), dtype=dtype or torch.float32, device=dev)
    for lo in range(0, F, chunk):
        n = min(chunk, F - lo)
        X = (torch.rand((n, P, J, 3), generator=g, dtype=f64, device=dev) - off) * box + centres[None, :, None, :]
        for c in range(C):
            pix = ((X - tt[c]) @ Rt[c]) @ Kt[c].T                   # K R^T (X - t)
            uv = pix[..., :2] / pix[..., 2:3] + torch.randn((n, P, J, 2), generator=g, dtype=f64, device=dev)
            sc = 3.5 + 4.5 * torch.rand((n, P, J, 1), generator=g, dtype=f64, device=dev)
            rec = torch.cat([uv, sc], dim=-1)
            if P > 1:                                               # a per-view detector lists its persons in its own order
                perm = torch.argsort(torch.rand((n, P), generator=g, device=dev), dim=1)
                rec = torch.gather(rec, 1, perm[:, :, None, None].expand(n, P, J, 3))
            kpts[lo:lo + n, c] = rec.to(kpts.dtype)
    n_persons = torch.full((F, C), P, dtype=torch.int32, device=dev)
    return dict(rig=(K, R, t), params=base["params"], kpts=kpts, n_persons=n_persons)

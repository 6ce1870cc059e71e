"""CPU restatement of row N2 (snowvision/blender.py): 133 joints -> 24 Blender IK control points, and the
per-control-point SecondOrderDynamic filtering of a whole track.

TEST INFRASTRUCTURE ONLY -- importable from tests/ and __graft_entry__.smoke(); the product package never
imports it.  Parity status: pinned against the reference through tests/golden/g6_smooth_blender.npz (control
points of 5 persons), g7_pipeline.npz (the main.py sequence) and g8_blender_track.npz (an 80-frame track with
invalid points through the filters) -- tests/test_oracle_golden.py.

The matrix -> quaternion step is SciPy's `Rotation.from_matrix` exactly as the reference calls it
(util.py:26-29; SciPy 1.15: orthogonal-Procrustes projection by SVD, then Markley's branch selection);
a NaN pelvis matrix makes SciPy's SVD raise LinAlgError in the reference and does so here.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

NAMES = ["root_position", "root_rotation", "clavicle_r_ik", "clavicle_l_ik", "arm_r_ik", "arm_r_pole",
         "arm_l_ik", "arm_l_pole", "leg_r_ik", "leg_r_pole", "leg_l_ik", "leg_l_pole", "hand_r_ik",
         "hand_r_pole", "hand_l_ik", "hand_l_pole", "foot_r_ik", "foot_r_pole", "foot_l_ik", "foot_l_pole",
         "chest_ik", "chest_pole", "head_ik", "head_pole"]          # configs/blender_armature_profile.json:2-25


def _unit(v):
    return v / np.linalg.norm(v)


def _mid(a, b):
    return (a + b) / 2


def _cross_pole(base, first, second):
    return base + _unit(np.cross(first, second))                     # blender.py:37-85 (one shape, eight uses)


def _joint_pole(joint, upper, lower):
    a, b, c = upper - joint, lower - joint, upper - lower              # blender.py:87-95
    return joint + _unit(np.cross(np.cross(b, a), c))


def _root_rotation(p5, p6, p11, p12):
    x = _unit(p11 - p12)                                               # blender.py:15-35
    y = _unit(_mid(p5, p6) - _mid(p11, p12))
    z = _unit(np.cross(x, y))
    q = Rotation.from_matrix(np.array([x, y, z]).T).as_quat()          # util.py:26-29  (x, y, z, w)
    return np.array([q[3], q[0], q[1], q[2]])                          # blender.py:28-29  (w, x, y, z)


def control_points(P):
    """P[J,3] (J >= 130) -> points[24,4] (3-vectors padded with 0), valid[24] uint8 (0 where any NaN)."""
    P = np.asarray(P, dtype=np.float64)
    sh, hip, ear = _mid(P[5], P[6]), _mid(P[11], P[12]), _mid(P[3], P[4])
    with np.errstate(all="ignore"):
        vals = [
            hip, _root_rotation(P[5], P[6], P[11], P[12]), P[6], P[5],
            P[10], _joint_pole(P[8], P[6], P[10]), P[9], _joint_pole(P[7], P[5], P[9]),
            P[16], _joint_pole(P[14], P[12], P[16]), P[15], _joint_pole(P[13], P[11], P[15]),
            P[121], _cross_pole(P[112], P[117] - P[112], P[129] - P[112]),
            P[100], _cross_pole(P[91], P[108] - P[91], P[96] - P[91]),
            _mid(P[20], P[21]), _cross_pole(P[22], P[20] - P[22], P[21] - P[22]),
            _mid(P[17], P[18]), _cross_pole(P[19], P[18] - P[19], P[17] - P[19]),
            sh, _cross_pole(sh, P[5] - P[6], sh - hip),
            sh + _unit(ear - sh), _cross_pole(ear, P[3] - P[4], ear - sh),
        ]
    out = np.zeros((24, 4))
    valid = np.ones(24, np.uint8)
    for i, v in enumerate(vals):
        out[i, :len(v)] = v
        valid[i] = 0 if np.isnan(v).any() else 1                       # blender.py:135-139
    return out, valid


def control_points_track(xyz):
    """xyz[..., J, 3] -> points[..., 24, 4], valid[..., 24]."""
    xyz = np.asarray(xyz, dtype=np.float64)
    lead = xyz.shape[:-2]
    flat = xyz.reshape((-1,) + xyz.shape[-2:])
    pts = np.zeros((flat.shape[0], 24, 4))
    val = np.zeros((flat.shape[0], 24), np.uint8)
    for i in range(flat.shape[0]):
        pts[i], val[i] = control_points(flat[i])
    return pts.reshape(lead + (24, 4)), val.reshape(lead + (24,))


def smooth_track(points, valid, fzr, dt):
    """blender.py:145-178 over a track: points[T,P,24,4], valid[T,P,24], fzr[24,3] -> smoothed[T,P,24,4].
    Frame 0 passes through untouched (NaNs included) and seeds each filter with the point, or zeros when the
    point is invalid; on a later invalid point the filter is fed its own previous input."""
    points = np.asarray(points, dtype=np.float64)
    T = points.shape[0]
    fzr = np.asarray(fzr, dtype=np.float64)
    f, z, r = fzr[:, 0, None], fzr[:, 1, None], fzr[:, 2, None]
    k1 = z / (np.pi * f)                                               # triangulation.py:6-8
    k2 = 1 / ((2 * np.pi * f) * (2 * np.pi * f))
    k3 = r * z / (2 * np.pi * f)
    ok = np.asarray(valid, dtype=bool)[..., None]
    out = np.empty_like(points)
    out[0] = points[0]
    xp = np.where(ok[0], points[0], 0.0)
    y, yd = xp.copy(), np.zeros_like(xp)
    for t in range(1, T):
        x = np.where(ok[t], points[t], xp)
        xd = (x - xp) / dt                                             # triangulation.py:16-21
        xp = x
        y = y + dt * yd
        yd = yd + dt * (x + k3 * xd - y - k1 * yd) / k2
        out[t] = y
    return out

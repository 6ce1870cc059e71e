"""Row N2 (after the hot path): 133 whole-body joints -> the 24 Blender IK control points of
snowvision/blender.py, per-control-point temporal filtering, and the on-disk JSON track.

Same names, signatures and dict schemas as the reference, so its caller sequence (main.py:80-87,104) runs on
this package.  The geometry runs on the GPU (snowtri_blender_points: one lane per skeleton); the per-frame
`Human_Triangulation_Blender` is a one-skeleton-batch call of it, `blender_points_track` / `blender_smooth_track`
are the additive whole-track forms (snowtri_blender_smooth: hold + per-bone filters as chunked scans).
`Human_Triangulation_Blender_Smooth` keeps the reference's stateful per-frame protocol (filter objects travel
inside the result dict, blender.py:152-178) and is host bookkeeping around those objects.
Pinned to the reference by tests/golden/g6_smooth_blender.npz, g7_pipeline.npz, g8_blender_track.npz.
"""
from __future__ import annotations

import copy
import json

import numpy as np

from . import _lib
from .triangulation import SecondOrderDynamic

# order of configs/blender_armature_profile.json = slot order of snowtri_blender_points (include/snowtri.h)
CONTROL_POINT_NAMES = ("root_position", "root_rotation", "clavicle_r_ik", "clavicle_l_ik", "arm_r_ik", "arm_r_pole",
                       "arm_l_ik", "arm_l_pole", "leg_r_ik", "leg_r_pole", "leg_l_ik", "leg_l_pole", "hand_r_ik",
                       "hand_r_pole", "hand_l_ik", "hand_l_pole", "foot_r_ik", "foot_r_pole", "foot_l_ik",
                       "foot_l_pole", "chest_ik", "chest_pole", "head_ik", "head_pole")
_SLOT = {n: i for i, n in enumerate(CONTROL_POINT_NAMES)}
_WIDTH = {n: (4 if n == "root_rotation" else 3) for n in CONTROL_POINT_NAMES}


def blender_points_track(xyzs):
    """xyzs[..., kn, 4] (x, y, z, score; float32 or float64, as the fused triangulation writes them) or
    xyzs[..., kn, 3]  ->  points[..., 24, 4] fp64 (3-vectors padded with 0; root_rotation = (w, x, y, z)),
    valid[..., 24] uint8.  GPU, one lane per skeleton."""
    a = np.asarray(xyzs)
    if a.shape[-1] == 3:
        a = np.concatenate([a, np.zeros(a.shape[:-1] + (1,), a.dtype)], axis=-1)
    if a.dtype != np.float32:
        a = a.astype(np.float64, copy=False)
    a = np.ascontiguousarray(a)
    lead, kn = a.shape[:-2], a.shape[-2]
    n = int(np.prod(lead)) if lead else 1
    pts = np.empty((n, 24, 4), np.float64)
    val = np.empty((n, 24), np.uint8)
    ctx = _lib.scratch_context()
    st = ctx.L.snowtri_blender_points(ctx.handle, n, kn, _lib.ptr(a), _lib.dtype_code(a.dtype), _lib.ptr(pts),
                                           _lib.ptr(val), _lib.HOST, None)
    if st == _lib.ERR_BAD_INDEX:
        raise IndexError(f"index 129 is out of bounds for axis 0 with size {kn}")      # blender.py:117
    _lib.check(st, "snowtri_blender_points")
    return pts.reshape(lead + (24, 4)), val.reshape(lead + (24,))


def blender_smooth_track(points, valid, blender_smooth_profile, delta_time=1 / 30):
    """Whole-track form of Human_Triangulation_Blender_Smooth: points[T, P, 24, 4], valid[T, P, 24], per-bone
    (f, z, r) from the smooth profile dict  ->  smoothed[T, P, 24, 4] (frame 0 as given)."""
    x = np.ascontiguousarray(points, dtype=np.float64)
    v = np.ascontiguousarray(valid, dtype=np.uint8)
    T, P = x.shape[0], x.shape[1]
    fzr = np.ascontiguousarray([blender_smooth_profile[n] for n in CONTROL_POINT_NAMES], dtype=np.float64)
    y = np.empty_like(x)
    ctx = _lib.scratch_context()
    _lib.check(ctx.L.snowtri_blender_smooth(ctx.handle, T, P, _lib.ptr(x), _lib.ptr(v), _lib.ptr(fzr),
                                                 float(delta_time), _lib.ptr(y), _lib.HOST, None),
               "snowtri_blender_smooth")
    return y


def save_blender_result(blender_result, file_path):
    with open(file_path, "w") as fh:
        fh.write(json.dumps(blender_result, indent=4))


def Human_Triangulation_Blender(result, blender_armature_profile):
    """Per person: {control point name: list} for every name in the armature profile, plus a 0/1 score
    (0 when the point is NaN, e.g. built from a zero-score joint at the origin) -- blender.py:98-143."""
    out = {"blender_armature_control_points": [], "blender_armature_control_points_scores": []}
    persons = result["hrnet_triangulate_points"]
    if len(persons) == 0:
        return out
    pts, val = blender_points_track(np.stack([np.asarray(p, dtype=np.float64) for p in persons]))
    if not val[:, _SLOT["root_rotation"]].all():
        raise np.linalg.LinAlgError("SVD did not converge")            # what SciPy raises on the NaN pelvis matrix
    for p in range(len(persons)):
        points, scores = copy.deepcopy(blender_armature_profile), copy.deepcopy(blender_armature_profile)
        for name in blender_armature_profile.keys():
            i = _SLOT[name]
            points[name] = pts[p, i, :_WIDTH[name]].tolist()
            scores[name] = int(val[p, i])
        out["blender_armature_control_points"].append(points)
        out["blender_armature_control_points_scores"].append(scores)
    return out


def Human_Triangulation_Blender_Smooth(current_blender_result, blender_armature_profile, blender_smooth_profile,
                                       previous_blender_result=None, delta_time=1 / 30):
    """Per control point SecondOrderDynamic with per-bone (f, z, r) (blender.py:145-178); an invalid point
    (score 0) feeds the filter its own previous input."""
    cur_pts = current_blender_result["blender_armature_control_points"]
    cur_sc = current_blender_result["blender_armature_control_points_scores"]
    out = {"blender_armature_control_points": [], "blender_armature_control_points_scores": cur_sc,
           "second_order_dynamics": []}
    if isinstance(previous_blender_result, dict):
        banks = previous_blender_result["second_order_dynamics"]
        for pts, bank, sc in zip(cur_pts, banks, cur_sc):
            smoothed = copy.deepcopy(blender_armature_profile)
            for name in blender_armature_profile.keys():
                flt = bank[name]
                x = np.array(pts[name]) if sc[name] else flt.xp
                smoothed[name] = flt.update(delta_time, x).tolist()
            out["blender_armature_control_points"].append(smoothed)
        out["second_order_dynamics"] = banks
    else:
        for pts, sc in zip(cur_pts, cur_sc):
            bank = copy.deepcopy(blender_armature_profile)
            for name in blender_armature_profile.keys():
                f, z, r = blender_smooth_profile[name]
                x0 = np.array(pts[name]) if sc[name] else np.zeros(len(pts[name]))
                bank[name] = SecondOrderDynamic(f, z, r, x0)
            out["second_order_dynamics"].append(bank)
        out["blender_armature_control_points"] = cur_pts
    return out


def Human_Triangulation_To_Blender_Result(result):
    """blender.py:180-187: {'armature': [per person dict], 'score': [per person dict]} for one frame."""
    return {"armature": list(result["blender_armature_control_points"]),
            "score": list(result["blender_armature_control_points_scores"])}

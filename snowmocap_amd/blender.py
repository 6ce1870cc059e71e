"""Row N2 (after the hot path): 133 whole-body joints -> the 24 Blender IK control points of
snowvision/blender.py, per-control-point temporal filtering, and the on-disk JSON track.

Host-side shim so the reference's caller sequence (main.py:80-87,104) runs on this package; same names,
signatures and dict schemas as the reference.  The geometry is written as one declarative table
(CONTROL_POINTS) instead of the reference's per-name helper calls + `eval` (blender.py:98-143); results are
pinned to the reference by tests/golden/g6_smooth_blender.npz.
"""
from __future__ import annotations

import copy
import json

import numpy as np
from scipy.spatial.transform import Rotation

from .triangulation import SecondOrderDynamic


def _unit(v):
    return v / np.linalg.norm(v)


def _mid(a, b):
    return (a + b) / 2


def _cross_pole(base, first, second):
    """base + unit(first x second)"""
    return base + _unit(np.cross(first, second))


def _joint_pole(joint, upper, lower):
    """Elbow / knee pole (blender.py:87-95): joint + unit((b x a) x c), a = upper-joint, b = lower-joint, c = upper-lower."""
    a, b, c = upper - joint, lower - joint, upper - lower
    return joint + _unit(np.cross(np.cross(b, a), c))


def _root_rotation(p5, p6, p11, p12):
    """Pelvis frame (blender.py:15-35): x = pelvis axis, y = spine, z = x x y (each normalised, NOT mutually
    orthogonal), turned into a quaternion by SciPy (which orthogonalises first) and reordered to (w, x, y, z)."""
    x = _unit(p11 - p12)
    y = _unit(_mid(p5, p6) - _mid(p11, p12))
    z = _unit(np.cross(x, y))
    q = Rotation.from_matrix(np.array([x, y, z]).T).as_quat()
    return np.array([q[3], q[0], q[1], q[2]])


def _head_ik(p3, p4, p5, p6):
    sh = _mid(p5, p6)
    return sh + _unit(_mid(p3, p4) - sh)


# name -> function of the person array P[J,3]; joint numbers are COCO-WholeBody indices (blender.py:105-130)
CONTROL_POINTS = {
    "root_position": lambda P: _mid(P[11], P[12]),
    "root_rotation": lambda P: _root_rotation(P[5], P[6], P[11], P[12]),
    "clavicle_r_ik": lambda P: P[6],
    "clavicle_l_ik": lambda P: P[5],
    "arm_r_ik": lambda P: P[10],
    "arm_r_pole": lambda P: _joint_pole(P[8], P[6], P[10]),
    "arm_l_ik": lambda P: P[9],
    "arm_l_pole": lambda P: _joint_pole(P[7], P[5], P[9]),
    "leg_r_ik": lambda P: P[16],
    "leg_r_pole": lambda P: _joint_pole(P[14], P[12], P[16]),
    "leg_l_ik": lambda P: P[15],
    "leg_l_pole": lambda P: _joint_pole(P[13], P[11], P[15]),
    "hand_r_ik": lambda P: P[121],
    "hand_r_pole": lambda P: _cross_pole(P[112], P[117] - P[112], P[129] - P[112]),
    "hand_l_ik": lambda P: P[100],
    "hand_l_pole": lambda P: _cross_pole(P[91], P[108] - P[91], P[96] - P[91]),
    "foot_r_ik": lambda P: _mid(P[20], P[21]),
    "foot_r_pole": lambda P: _cross_pole(P[22], P[20] - P[22], P[21] - P[22]),
    "foot_l_ik": lambda P: _mid(P[17], P[18]),
    "foot_l_pole": lambda P: _cross_pole(P[19], P[18] - P[19], P[17] - P[19]),
    "chest_ik": lambda P: _mid(P[5], P[6]),
    "chest_pole": lambda P: _cross_pole(_mid(P[5], P[6]), P[5] - P[6], _mid(P[5], P[6]) - _mid(P[11], P[12])),
    "head_ik": lambda P: _head_ik(P[3], P[4], P[5], P[6]),
    "head_pole": lambda P: _cross_pole(_mid(P[3], P[4]), P[3] - P[4], _mid(P[3], P[4]) - _mid(P[5], P[6])),
}


def save_blender_result(blender_result, file_path):
    with open(file_path, "w") as fh:
        fh.write(json.dumps(blender_result, indent=4))


def Human_Triangulation_Blender(result, blender_armature_profile):
    """Per person: {control point name: list} for every name in the armature profile, plus a 0/1 score
    (0 when the point is NaN, e.g. built from a zero-score joint at the origin) -- blender.py:98-143."""
    out = {"blender_armature_control_points": [], "blender_armature_control_points_scores": []}
    for person in result["hrnet_triangulate_points"]:
        P = np.asarray(person)
        points, scores = copy.deepcopy(blender_armature_profile), copy.deepcopy(blender_armature_profile)
        with np.errstate(all="ignore"):
            for name in blender_armature_profile.keys():
                value = CONTROL_POINTS[name](P)
                points[name] = value.tolist()
                scores[name] = 0 if np.isnan(value).any() else 1
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

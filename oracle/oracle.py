"""ctypes front-end of the CPU oracle (oracle/snowtri_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package `snowmocap_amd` never imports this module.
Parity status: pinned against the reference through tests/golden/*.npz (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsnowtri_oracle.so")

ORC_OK, ORC_SINGULAR, ORC_BAD_INDEX, ORC_OVERFLOW = 0, 1, 2, 3


class OrcParams(ct.Structure):
    _fields_ = [("keypoint_score_threshold", ct.c_double),
                ("average_score_threshold", ct.c_double),
                ("distance_threshold", ct.c_double),
                ("condense_distance_tol", ct.c_double),
                ("condense_person_num_tol", ct.c_double),
                ("condense_score_tol", ct.c_double),
                ("center_point_index", ct.c_int32),
                ("keypoint_num", ct.c_int32)]


def make_params(keypoint_score_threshold=0.5, average_score_threshold=0.0, distance_threshold=0.05,
                condense_distance_tol=0.1, condense_person_num_tol=0, condense_score_tol=0.0,
                center_point_index=18, keypoint_num=30, **_ignored):
    """Defaults are the reference's function-signature defaults (triangulation.py:50,95-100)."""
    return OrcParams(float(keypoint_score_threshold), float(average_score_threshold),
                     float(distance_threshold), float(condense_distance_tol),
                     float(condense_person_num_tol), float(condense_score_tol),
                     int(center_point_index), int(keypoint_num))


def build(force=False):
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "snowtri_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ct.CDLL(build())
        _lib.orc_triangulate_condense_batch.restype = ct.c_int
    return _lib


def _p(a, typ=ct.c_double):
    return a.ctypes.data_as(ct.POINTER(typ))


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def rays_from_pixels(K, R, uv):
    """A1 (camera.py:234-253): uv[J,2] -> rays[J,3]."""
    uv = _c64(uv)
    J = uv.shape[0]
    rays = np.empty((J, 3))
    rc = lib().orc_rays_from_pixels(_p(_c64(K)), _p(_c64(R)), _p(uv), J, _p(rays))
    if rc:
        raise np.linalg.LinAlgError("Singular matrix")
    return rays


def skew_ray_solver(hm, hs, tm, ts):
    """A2 (triangulation.py:24-31) on single (3,)/(3,1) vectors -> (dist, W[3])."""
    dist = ct.c_double()
    W = np.empty(3)
    rc = lib().orc_skew_ray_solver(_p(_c64(hm).ravel()), _p(_c64(hs).ravel()), _p(_c64(tm).ravel()),
                                   _p(_c64(ts).ravel()), ct.byref(dist), _p(W))
    if rc:
        raise np.linalg.LinAlgError("Singular matrix")
    return dist.value, W


def skew_ray_solver_batch(hm, hs, tm, ts):
    hm, hs, tm, ts = (_c64(x).reshape(-1, 3) for x in (hm, hs, tm, ts))
    n = hm.shape[0]
    dist = np.empty(n)
    W = np.empty((n, 3))
    sing = np.zeros(n, dtype=np.int32)
    lib().orc_skew_ray_solver_batch(ct.c_long(n), _p(hm), _p(hs), _p(tm), _p(ts), _p(dist), _p(W),
                                    _p(sing, ct.c_int32))
    return dist, W, sing


def human_triangulation_frame(K, R, t, kpts, n_persons, params, score_is_f32=None):
    """A1 + A3 for one frame.  kpts[C,Pmax,J,3] (u,v,score), n_persons[C].
    Returns the reference's result dict (lists of fp64 arrays) plus 'enum_index'."""
    kpts = np.asarray(kpts)
    if score_is_f32 is None:
        score_is_f32 = kpts.dtype == np.float32
    C, Pmax, J, _ = kpts.shape
    K, R, t = _c64(K), _c64(R), _c64(t).reshape(C, 3)
    n_persons = np.ascontiguousarray(n_persons, dtype=np.int32)
    rays = np.zeros((C, Pmax, J, 3))
    score = np.ascontiguousarray(kpts[..., 2], dtype=np.float64)
    for c in range(C):
        for p in range(int(n_persons[c])):
            rays[c, p] = rays_from_pixels(K[c], R[c], kpts[c, p, :, :2])
    maxc = max(1, sum(int(n_persons[a]) * int(n_persons[b]) for a in range(C) for b in range(a + 1, C)))
    cxyz = np.empty((maxc, J, 3))
    cks = np.empty((maxc, J))
    cps = np.empty(maxc)
    cidx = np.empty(maxc, dtype=np.int32)
    n = ct.c_int32()
    nsing = ct.c_int32()
    rc = lib().orc_human_triangulation(C, Pmax, J, _p(n_persons, ct.c_int32), _p(rays), _p(score),
                                       int(bool(score_is_f32)), _p(t), ct.byref(params), maxc,
                                       _p(cxyz), _p(cks), _p(cps), _p(cidx, ct.c_int32),
                                       ct.byref(n), ct.byref(nsing))
    if rc == ORC_SINGULAR:
        raise np.linalg.LinAlgError("Singular matrix")
    k = n.value
    return {"hrnet_triangulate_points": [cxyz[i].copy() for i in range(k)],
            "hrnet_triangulate_keypoint_scores": [cks[i].copy() for i in range(k)],
            "hrnet_triangulate_person_scores": [np.float64(cps[i]) for i in range(k)],
            "enum_index": cidx[:k].copy()}


def condense_frame(result, params, max_out=None):
    """A4 for one frame on a reference-shaped result dict."""
    pts = result["hrnet_triangulate_points"]
    n = len(pts)
    if n == 0:
        return {"hrnet_triangulate_points": [], "hrnet_triangulate_keypoint_scores": [],
                "hrnet_triangulate_person_scores": [], "member_count": []}
    cxyz = _c64(np.stack(pts))
    cks = _c64(np.stack(result["hrnet_triangulate_keypoint_scores"]))
    J = cxyz.shape[1]
    kn = params.keypoint_num
    max_out = max(1, n if max_out is None else max_out)
    oxyz = np.zeros((max_out, max(kn, 0), 3))
    oks = np.zeros((max_out, max(kn, 0)))
    ops = np.zeros(max_out)
    mem = np.zeros(max_out, dtype=np.int32)
    no = ct.c_int32()
    rc = lib().orc_condense(n, J, _p(cxyz), _p(cks), ct.byref(params), max_out, _p(oxyz), _p(oks),
                            _p(ops), _p(mem, ct.c_int32), ct.byref(no))
    if rc == ORC_BAD_INDEX:
        raise IndexError("center_point_index / keypoint_num out of range")
    k = min(no.value, max_out)
    return {"hrnet_triangulate_points": [oxyz[i].copy() for i in range(k)],
            "hrnet_triangulate_keypoint_scores": [oks[i].copy() for i in range(k)],
            "hrnet_triangulate_person_scores": [np.float64(ops[i]) for i in range(k)],
            "member_count": mem[:k].copy()}


def triangulate_condense_batch(K, R, t, kpts, n_persons, params, max_out, nthreads=0):
    """A1..A4 over a batch.  kpts[F,C,Pmax,J,3] float32 or float64.  Returns dict of arrays:
    xyz[F,max_out,kn,3], kscore[F,max_out,kn], pscore[F,max_out], count[F], status[F], threads."""
    kpts = np.ascontiguousarray(kpts)
    assert kpts.dtype in (np.float32, np.float64)
    F, C, Pmax, J, _ = kpts.shape
    K, R, t = _c64(K), _c64(R), _c64(t).reshape(C, 3)
    n_persons = np.ascontiguousarray(n_persons, dtype=np.int32)
    kn = params.keypoint_num
    oxyz = np.zeros((F, max_out, kn, 3))
    oks = np.zeros((F, max_out, kn))
    ops = np.zeros((F, max_out))
    cnt = np.zeros(F, dtype=np.int32)
    st = np.zeros(F, dtype=np.int32)
    used = lib().orc_triangulate_condense_batch(
        ct.c_long(F), C, Pmax, J, _p(K), _p(R), _p(t), kpts.ctypes.data_as(ct.c_void_p),
        int(kpts.dtype == np.float32), _p(n_persons, ct.c_int32), ct.byref(params), max_out,
        _p(oxyz), _p(oks), _p(ops), _p(cnt, ct.c_int32), _p(st, ct.c_int32), int(nthreads))
    return dict(xyz=oxyz, kscore=oks, pscore=ops, count=cnt, status=st, threads=used)


def second_order_track(x, f, z, r, dt):
    """N1: SecondOrderDynamic over a track x[T, ...] (triangulation.py:4-22)."""
    x = _c64(x)
    T = x.shape[0]
    n = int(np.prod(x.shape[1:])) if x.ndim > 1 else 1
    y = np.empty_like(x)
    lib().orc_second_order_track(ct.c_long(T), ct.c_long(n), _p(x), ct.c_double(f), ct.c_double(z),
                                 ct.c_double(r), ct.c_double(dt), _p(y))
    return y

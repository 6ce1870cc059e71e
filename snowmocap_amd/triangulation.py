"""The reference's triangulation API (snowvision/triangulation.py) on the MI355X kernels.

Same names, positional/keyword signatures, return schema and error behaviour as the reference:

  Skew_Ray_Solver(hm, hs, tm, ts)                      triangulation.py:24-31   -> snowtri_skew_ray_batch
  Human_Triangulation(camera_group, ...)               triangulation.py:50-93   -> snowtri_triangulate
  Human_Triangulation_Condense(result, ...)            triangulation.py:95-162  -> snowtri_condense
  Human_Triangulation_Smooth / SecondOrderDynamic      triangulation.py:4-22,164-186 (host shim, row N1)

Every ★ function runs on the GPU through the C ABI; there is no NumPy fallback.
"""
from __future__ import annotations

import ctypes as ct
import math

import numpy as np

from . import _lib

_KEYS = ("hrnet_triangulate_points", "hrnet_triangulate_keypoint_scores", "hrnet_triangulate_person_scores")


def Skew_Ray_Solver(hm, hs, tm, ts):
    """Closest approach of rays tm + a*hm and ts + b*hs -> (skew distance, midpoint[3])."""
    ctx = _lib.scratch_context()
    L = ctx.L
    arrs = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(1, 3)) for x in (hm, hs, tm, ts)]
    dist = np.empty(1)
    W = np.empty((1, 3))
    rc = L.snowtri_skew_ray_batch(ctx.handle, 1, *[_lib.ptr(a) for a in arrs], _lib.ptr(dist), _lib.ptr(W), None)
    if rc == _lib.ERR_SINGULAR:
        raise np.linalg.LinAlgError("Singular matrix")
    _lib.check(rc, "snowtri_skew_ray_batch")
    return np.float64(dist[0]), W[0].copy()


def skew_ray_solver_batch(hm, hs, tm, ts):
    """Additive batched form of Skew_Ray_Solver: [n,3] x4 -> dist[n], W[n,3], n_singular."""
    ctx = _lib.scratch_context()
    L = ctx.L
    arrs = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, 3)) for x in (hm, hs, tm, ts)]
    n = arrs[0].shape[0]
    dist = np.empty(n)
    W = np.empty((n, 3))
    import ctypes as ct
    ns = ct.c_int64(0)
    rc = L.snowtri_skew_ray_batch(ctx.handle, n, *[_lib.ptr(a) for a in arrs], _lib.ptr(dist), _lib.ptr(W),
                                  ct.byref(ns))
    if rc not in (_lib.OK, _lib.ERR_SINGULAR):
        _lib.check(rc, "snowtri_skew_ray_batch")
    return dist, W, int(ns.value)


def Human_Triangulation(camera_group, keypoint_score_threshold=0.5, average_score_threshold=0.0,
                        distance_threshold=0.05):
    """All camera-pair x person-pair candidate skeletons of the current frame, scored and filtered
    by their mean score; list order = the reference's loop order (triangulation.py:56-65)."""
    ctx = camera_group.native_context()
    L = ctx.L
    kpts, n_persons = camera_group.pack_frame()
    C, Pmax, J, _ = kpts.shape
    Kc = int(L.snowtri_num_candidate_slots(C, Pmax))
    result = {k: [] for k in _KEYS}
    if Kc == 0 or not n_persons.any():
        return result
    prm = _lib.make_params(keypoint_score_threshold=keypoint_score_threshold,
                           average_score_threshold=average_score_threshold,
                           distance_threshold=distance_threshold)
    # the four outputs in ONE block (a pointer costs more than the arrays: ~1.3 us per ndarray.ctypes.data)
    n3, n4 = Kc * J * 3, Kc * J * 4
    blk = np.empty(n4 + Kc + (Kc + 7) // 8)
    xyz, ks, ps = blk[:n3].reshape(Kc, J, 3), blk[n3:n4].reshape(Kc, J), blk[n4:n4 + Kc]
    keep = blk[n4 + Kc:].view(np.uint8)[:Kc]
    base = blk.ctypes.data
    vp = ct.c_void_p
    rc = L.snowtri_triangulate(ctx.handle, 1, Pmax, J, _lib.ptr(kpts), _lib.dtype_code(kpts.dtype),
                               _lib.ptr(n_persons), prm, vp(base), vp(base + 8 * n3), vp(base + 8 * n4),
                               vp(base + 8 * (n4 + Kc)), _lib.HOST, None)
    if rc == _lib.ERR_SINGULAR:
        raise np.linalg.LinAlgError("Singular matrix")      # np.linalg.inv at triangulation.py:26
    _lib.check(rc, "snowtri_triangulate")
    kept = np.nonzero(keep)[0]
    # the candidates are rows of the block the library filled (no copy per candidate); the block and the device-side twin
    # the call left behind are remembered so that Human_Triangulation_Condense can skip the upload (_Resident)
    pts = [xyz[k] for k in kept]
    scs = [ks[k] for k in kept]
    result[_KEYS[0]] = pts
    result[_KEYS[1]] = scs
    result[_KEYS[2]] = [np.float64(ps[k]) for k in kept]
    _Resident.remember(ctx, int(L.snowtri_candidates_token(ctx.handle)), pts, scs, blk, n4, J)
    return result


class _Resident:
    """The device-resident twin of the last Human_Triangulation result (main.py:62-71 calls Condense right after it).

    Human_Triangulation leaves its candidates in the context's device scratch (snowtri_candidates_token) and hands the caller
    VIEWS of the two host blocks it downloaded.  Human_Triangulation_Condense may use the device twin instead of uploading
    the candidates again only if the dict it receives still IS that result: the same view objects in the same order, and
    the blocks' contents unchanged (a user may edit candidates in place -- then, or after any other change, the ordinary
    upload path runs on the arrays as they are)."""
    last = None
    used = 0       # Condense calls that took the device twin (tests / bench)

    @classmethod
    def remember(cls, ctx, token, pts, scs, blk, n4, J):
        # (the bytes of the candidate block as downloaded: an in-place edit of any candidate shows as a difference)
        cls.last = (ctx, token, pts, scs, blk, n4, blk[:n4].tobytes(), J) if token else None

    @classmethod
    def match(cls, points, scores):
        r = cls.last
        if r is None:
            return None
        ctx, token, pts, scs, blk, n4, pristine, J = r
        if len(points) != len(pts) or len(scores) != len(scs) or not ctx.handle:
            return None
        for a, b in zip(points, pts):
            if a is not b:
                return None
        for a, b in zip(scores, scs):
            if a is not b:
                return None
        if blk[:n4].tobytes() != pristine:
            return None
        return ctx, token, J


def Human_Triangulation_Condense(result, condense_distance_tol=0.1, condense_person_num_tol=0,
                                 condense_score_tol=0.0, center_point_index=18, keypoint_num=30):
    """Greedy centre-joint clustering of the candidates + score-weighted fusion per joint."""
    points = result[_KEYS[0]]
    scores = result[_KEYS[1]]
    out = {k: [] for k in _KEYS}
    n = len(points)
    if n - 1 <= 0:                 # range(person_num - 1) is empty: nothing is ever emitted
        return out
    prm = _lib.make_params(condense_distance_tol=condense_distance_tol,
                           condense_person_num_tol=condense_person_num_tol,
                           condense_score_tol=condense_score_tol,
                           center_point_index=center_point_index, keypoint_num=keypoint_num)
    kn = int(keypoint_num)
    pout = n                        # at most n - 1 clusters
    knn = max(kn, 0)
    oblk = np.empty(pout * knn * 4 + pout)       # the three float64 outputs in one block (one pointer)
    oxyz, oks, ops = oblk[:pout * knn * 3].reshape(pout, knn, 3), oblk[pout * knn * 3:pout * knn * 4].reshape(pout, knn), oblk[pout * knn * 4:]
    cnt = np.zeros(1, dtype=np.int32)
    obase = oblk.ctypes.data
    p_xyz, p_ks, p_ps, p_cnt = ct.c_void_p(obase), ct.c_void_p(obase + 8 * pout * knn * 3), ct.c_void_p(obase + 8 * pout * knn * 4), _lib.ptr(cnt)
    rc = _lib.ERR_BAD_ARG
    resident = _Resident.match(points, scores)
    if resident is not None:        # the unmodified result of Human_Triangulation: its candidates are still on the device
        ctx, token, J = resident
        rc = ctx.L.snowtri_condense_resident(ctx.handle, token, prm, pout, p_xyz, p_ks, p_ps, p_cnt, None)
        _Resident.used += rc != _lib.ERR_BAD_ARG
    if resident is None or rc == _lib.ERR_BAD_ARG:
        ctx = _lib.scratch_context()
        cxyz = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.float64) for p in points]))
        cks = np.ascontiguousarray(np.stack([np.asarray(s, dtype=np.float64) for s in scores]))
        J = cxyz.shape[1]
        rc = ctx.L.snowtri_condense(ctx.handle, 1, n, J, _lib.ptr(cxyz), _lib.ptr(cks), None, prm, pout,
                                p_xyz, p_ks, p_ps, p_cnt, None, _lib.HOST, None)
    if rc == _lib.ERR_BAD_INDEX:
        raise IndexError("index out of bounds (center_point_index / keypoint_num vs. joints per candidate)")
    _lib.check(rc, "snowtri_condense")
    for i in range(int(cnt[0])):
        out[_KEYS[0]].append(oxyz[i].copy())
        out[_KEYS[1]].append(oks[i].copy())
        out[_KEYS[2]].append(np.float64(ops[i]))
    return out


class SecondOrderDynamic:
    """Second-order low-pass (semi-implicit Euler), triangulation.py:4-22: state (xp, y, yd),
    gains k1 = z/(pi f), k2 = 1/(2 pi f)^2, k3 = r z/(2 pi f)."""

    def __init__(self, f, z, r, x0):
        w = 2.0 * math.pi * f
        self.k1 = z / (math.pi * f)
        self.k2 = 1.0 / (w * w)
        self.k3 = r * z / w
        self.xp = x0
        self.y = x0
        self.yd = 0

    def update(self, T, x, xd=None):
        if xd is None:
            xd = (x - self.xp) / T
            self.xp = x
        self.y = self.y + T * self.yd
        self.yd = self.yd + T * (x + self.k3 * xd - self.y - self.k1 * self.yd) / self.k2
        return self.y


def Human_Triangulation_Smooth(result, previous_result=None, f=2, z=0.75, r=0, delta_time=1 / 30):
    """Per-joint temporal filter carried from frame to frame inside the result dict
    (triangulation.py:164-186).  Persons are matched by list index; the first frame passes through
    and seeds one SecondOrderDynamic per joint."""
    out = {_KEYS[1]: result[_KEYS[1]], _KEYS[2]: result[_KEYS[2]]}
    if isinstance(previous_result, dict):
        filters = previous_result["second_order_dynamics"]
        out[_KEYS[0]] = [[flt.update(delta_time, joint) for joint, flt in zip(person, bank)]
                         for person, bank in zip(result[_KEYS[0]], filters)]
        out["second_order_dynamics"] = filters
    else:
        out[_KEYS[0]] = result[_KEYS[0]]
        out["second_order_dynamics"] = [[SecondOrderDynamic(f, z, r, joint) for joint in person]
                                        for person in result[_KEYS[0]]]
    return out


def smooth_track(track, f=2, z=0.75, r=0, delta_time=1 / 30):
    """Additive batched form of Human_Triangulation_Smooth (row N1): track[T, ...] (e.g. [T, P, J, 3]) ->
    filtered track of the same shape, every trailing element an independent SecondOrderDynamic lane,
    frame 0 passing through.  Runs on the GPU as a chunked linear scan (snowtri_smooth_track)."""
    x = np.ascontiguousarray(track, dtype=np.float64)
    T = x.shape[0]
    n = int(np.prod(x.shape[1:])) if x.ndim > 1 else 1
    y = np.empty_like(x)
    ctx = _lib.scratch_context()
    _lib.check(ctx.L.snowtri_smooth_track(ctx.handle, T, n, _lib.ptr(x), float(f), float(z), float(r),
                                               float(delta_time), _lib.ptr(y), _lib.HOST, None),
               "snowtri_smooth_track")
    return y

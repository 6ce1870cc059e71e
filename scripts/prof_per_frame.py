#!/usr/bin/env python3
"""GPU box: where a frame of the per-frame API goes (floor rig, 4 x 1 x 133): Python packing, the two C calls, list building."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import snowmocap_amd as sm
from snowmocap_amd import synth, _lib
wl = synth.config_workload(1, 220)
kp = wl["kpts"]
C = kp.shape[1]
cg = sm.CameraGroup(camera_group_info_path=synth.FLOOR_RIG_PATH)
L = _lib.lib()
T = {k: [] for k in ("add", "pack", "c_triangulate", "lists", "c_condense_resident", "c_condense_upload", "whole_tri", "whole_con")}
prm_t = _lib.make_params(keypoint_score_threshold=3.0, average_score_threshold=0.0, distance_threshold=0.05)
prm_c = _lib.make_params(condense_distance_tol=10, condense_person_num_tol=0, condense_score_tol=0.0, center_point_index=0, keypoint_num=133)
for f in range(220):
    t0 = time.perf_counter()
    for c in range(C):
        cg.add_human_2D_points(kp[f, c, 0, :, :2], kp[f, c, 0, :, 2], c)
    t1 = time.perf_counter()
    kpts, npers = cg.pack_frame()
    t2 = time.perf_counter()
    ctx = cg.native_context()
    Kc, J = 6, 133
    xyz = np.empty((Kc, J, 3)); ks = np.empty((Kc, J)); ps = np.empty(Kc); keep = np.empty(Kc, dtype=np.uint8)
    t3 = time.perf_counter()
    rc = L.snowtri_triangulate(ctx.handle, 1, 1, J, _lib.ptr(kpts), _lib.dtype_code(kpts.dtype), _lib.ptr(npers), prm_t, _lib.ptr(xyz), _lib.ptr(ks), _lib.ptr(ps), _lib.ptr(keep), _lib.HOST, None)
    t4 = time.perf_counter()
    kept = np.nonzero(keep)[0]
    pts = [xyz[k] for k in kept]; scs = [ks[k] for k in kept]; pss = [np.float64(ps[k]) for k in kept]
    xyz0 = xyz.copy(); ks0 = ks.copy()
    t5 = time.perf_counter()
    tok = L.snowtri_candidates_token(ctx.handle)
    oxyz = np.empty((6, 133, 3)); oks = np.empty((6, 133)); ops = np.empty(6); cnt = np.zeros(1, dtype=np.int32)
    t6 = time.perf_counter()
    rc = L.snowtri_condense_resident(ctx.handle, tok, prm_c, 6, _lib.ptr(oxyz), _lib.ptr(oks), _lib.ptr(ops), _lib.ptr(cnt), None)
    t7 = time.perf_counter()
    sctx = _lib.scratch_context()
    rc = L.snowtri_condense(sctx.handle, 1, 6, J, _lib.ptr(xyz), _lib.ptr(ks), None, prm_c, 6, _lib.ptr(oxyz), _lib.ptr(oks), _lib.ptr(ops), _lib.ptr(cnt), None, _lib.HOST, None)
    t8 = time.perf_counter()
    tri = sm.Human_Triangulation(cg, keypoint_score_threshold=3.0, average_score_threshold=0.0, distance_threshold=0.05)
    t9 = time.perf_counter()
    con = sm.Human_Triangulation_Condense(tri, condense_distance_tol=10, condense_person_num_tol=0, condense_score_tol=0.0, center_point_index=0, keypoint_num=133)
    t10 = time.perf_counter()
    cg.clear_2D_points()
    if f >= 20:
        for k, v in (("add", t1 - t0), ("pack", t2 - t1), ("c_triangulate", t4 - t3), ("lists", t5 - t4), ("c_condense_resident", t7 - t6),
                     ("c_condense_upload", t8 - t7), ("whole_tri", t9 - t8), ("whole_con", t10 - t9)):
            T[k].append(v)
print({k: round(float(np.median(v)) * 1e6, 1) for k, v in T.items()})

#!/usr/bin/env python3
"""Latency of the path main.py really calls (reference main.py:50-71,106), per frame, on the floor rig (BASELINE
configs[0] shape: 4 cameras x 1 person x 133 joints, 300 frames):

    add_human_2D_points x C -> Human_Triangulation -> Human_Triangulation_Condense -> clear_2D_points

through the reference-named Python API, and the same frame through ONE host-memory call of the C ABI
(snowtri_triangulate_condense, F = 1).  Importable: bench.py calls per_frame_api()."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_frame_api(frames=300, warm=20):
    import snowmocap_amd as sm
    from snowmocap_amd import synth, _lib
    from snowmocap_amd.batch import BatchTriangulator
    wl = synth.config_workload(1, frames + warm)           # cfg1: floor rig, 4 cameras x 1 person, default thresholds
    K, R, t = wl["rig"]
    kp = wl["kpts"]                                        # [F, C, 1, J, 3] float32
    C = kp.shape[1]
    cg = sm.CameraGroup(camera_group_info_path=synth.FLOOR_RIG_PATH)
    tri_kw = dict(keypoint_score_threshold=3.0, average_score_threshold=0.0, distance_threshold=0.05)
    con_kw = dict(condense_distance_tol=10, condense_person_num_tol=0, condense_score_tol=0.0, center_point_index=0, keypoint_num=133)
    people = [[(np.ascontiguousarray(kp[f, c, 0, :, :2]), np.ascontiguousarray(kp[f, c, 0, :, 2])) for c in range(C)]
              for f in range(frames + warm)]
    t_all, t_tri, t_con = [], [], []
    last = None
    from snowmocap_amd.triangulation import _Resident
    resident0 = _Resident.used
    for f in range(frames + warm):
        a = time.perf_counter()
        for c in range(C):
            cg.add_human_2D_points(people[f][c][0], people[f][c][1], c)
        b = time.perf_counter()
        tri = sm.Human_Triangulation(cg, **tri_kw)
        c_ = time.perf_counter()
        con = sm.Human_Triangulation_Condense(tri, **con_kw)
        d = time.perf_counter()
        cg.clear_2D_points()
        e = time.perf_counter()
        if f >= warm:
            t_all.append(e - a)
            t_tri.append(c_ - b)
            t_con.append(d - c_)
        last = con
    assert len(last["hrnet_triangulate_points"]) == 1
    # the same frame through one fused host call of the C ABI
    bt = BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float64)
    t_fused = []
    for f in range(frames + warm):
        a = time.perf_counter()
        out = bt.run_host(kp[f:f + 1], wl["n_persons"][f:f + 1])
        b = time.perf_counter()
        if f >= warm:
            t_fused.append(b - a)
    err = float(np.abs(out["xyzs"][0, 0, :, :3] - np.asarray(last["hrnet_triangulate_points"][0])).max())
    bt.close()
    us = lambda v: float(np.median(v) * 1e6)
    return {"workload": "BASELINE configs[0] shape: floor rig, 4 cameras x 1 person x 133 joints, %d frames one by one" % frames,
            "api_sequence_us_median": us(t_all), "api_sequence_us_p90": float(np.percentile(t_all, 90) * 1e6),
            "human_triangulation_us": us(t_tri), "human_triangulation_condense_us": us(t_con),
            "fused_host_call_us_median": us(t_fused),
            "condense_calls_on_device_resident_candidates": _Resident.used - resident0,
            "reference_ms_per_frame": 24.9,
            "what": "add_human_2D_points x %d -> Human_Triangulation -> Human_Triangulation_Condense -> clear_2D_points "
                    "(reference main.py:50-71,106), wall time per frame incl. Python, PCIe and synchronisation; fused = one "
                    "snowtri_triangulate_condense(F = 1) host call; reference: BASELINE.md 2 (NumPy, this container's CPU)" % C,
            "fused_vs_api_max_abs_m": err}


if __name__ == "__main__":
    import json
    print(json.dumps(per_frame_api()))

#!/usr/bin/env python3
"""Dev aid (GPU box): exact ray intersections (dist == 0 -> inf score) combined with a gated (below-threshold)
confidence, through the multi-person kernel vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import snowmocap_amd as api
from oracle import oracle as orc
C, P, J = 3, 2, 4
K = np.tile(np.eye(3), (C, 1, 1)); R = np.tile(np.eye(3), (C, 1, 1))
t = np.array([[0.0, 0, 0], [2.0, 0, 0], [0, 2.0, 0]])
X = np.array([[[1.0, 0.0, 4.0], [0.5, 0.5, 2.0], [1.0, 1.0, 4.0], [0.0, 1.0, 2.0]],
              [[3.0, 2.0, 8.0], [2.5, 2.5, 4.0], [3.0, 3.0, 8.0], [2.0, 3.0, 4.0]]])   # [P,J,3], exactly representable
kp = np.zeros((1, C, P, J, 3))
for c in range(C):
    for p in range(P):
        kp[0, c, p, :, 0] = (X[p, :, 0] - t[c, 0]) / X[p, :, 2]
        kp[0, c, p, :, 1] = (X[p, :, 1] - t[c, 1]) / X[p, :, 2]
        kp[0, c, p, :, 2] = 5.0
kp[0, 0, 0, 1, 2] = 1.0      # one gated confidence on an exactly intersecting joint
kp[0, 1, 1, 2, 2] = 1.0
npers = np.full((1, C), P, np.int32)
prm = dict(keypoint_score_threshold=3.0, average_score_threshold=0.0, distance_threshold=0.05, condense_distance_tol=0.5,
           condense_person_num_tol=0, condense_score_tol=0.0, center_point_index=0, keypoint_num=J)
ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 8)
bt = api.BatchTriangulator(K, R, t, prm, pout_max=8, out_dtype=np.float64)
out = bt.run_host(kp, npers)
print("count", out["count"], ref["count"], "status", out["status"], ref["status"])
m = int(ref["count"][0])
np.set_printoptions(linewidth=200, precision=4)
g, o = out["xyzs"][0, :m, :, 3], ref["kscore"][0, :m]
print("kscore NaN/inf pattern equal:", np.array_equal(np.isnan(g), np.isnan(o)), np.array_equal(np.isinf(g), np.isinf(o)))
print("oracle kscore", o.tolist())
print("gpu    kscore", g.tolist())
gx, ox = out["xyzs"][0, :m, :, :3], ref["xyz"][0, :m]
print("xyz NaN pattern equal:", np.array_equal(np.isnan(gx), np.isnan(ox)))
print("persons with NaN: gpu", np.isnan(gx).any(axis=(1, 2)).tolist(), "oracle", np.isnan(ox).any(axis=(1, 2)).tolist())

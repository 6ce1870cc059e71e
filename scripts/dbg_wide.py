"""Dev aid (GPU box): which (cameras, persons) shapes the streaming association hands over, and how many frames it leaves behind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import snowmocap_amd as sm
from snowmocap_amd import synth
PRM = dict(keypoint_score_threshold=3.0, average_score_threshold=0.3, distance_threshold=0.05, condense_distance_tol=0.3,
           condense_person_num_tol=10, condense_score_tol=0.0, center_point_index=0)
shapes = [(16, 3), (16, 2), (16, 4), (16, 8), (15, 3), (14, 3), (12, 3), (10, 3), (9, 3)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
for C, P in shapes:
    rng = np.random.default_rng(1600 + 10 * C + P)
    F, J = 6, 133
    K, R, t = synth.ring_rig(C, radius=5.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
    mod = os.environ.get("DBG_MOD", "")
    npers = npers.copy()
    if "z" in mod:
        kp[1, 2, 0, :, 2] = 0.0
    if "r" in mod:
        npers[2, C - 1] = P - 1
    bt = sm.BatchTriangulator(K, R, t, dict(PRM, keypoint_num=J), pout_max=P + 2, out_dtype=np.float32)
    out = bt.run_host(kp, npers)
    print(f"mod={mod} C={C} P={P} flags={out['flags'].tolist()} handed={bt.ctx.last_handover_persons()} count={out['count'].tolist()} kernels={bt.ctx.last_kernel_names()[:60]}", flush=True)
    bt.close()

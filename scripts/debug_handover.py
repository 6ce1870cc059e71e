import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import snowmocap_amd as api
from snowmocap_amd import synth
from oracle import oracle as orc
rng = np.random.default_rng(1)
C, P, F, J = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 2, 3, 133
K, R, t = synth.ring_rig(C, radius=4.5)
X = synth.make_people(rng, F, P, J=J)
kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
prm = dict(keypoint_score_threshold=3.0, average_score_threshold=0.3, distance_threshold=0.05, condense_distance_tol=0.3,
           condense_person_num_tol=2, condense_score_tol=0.0, center_point_index=0, keypoint_num=J)
ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
for hm in ("1", "0"):
    os.environ["SNOWTRI_HANDOVER_MODE"] = hm
    bt = api.BatchTriangulator(K, R, t, prm, pout_max=3, out_dtype=np.float32)
    out = bt.run_host(kp, npers)
    print("mode", hm, "handed", bt.ctx.last_handover_persons(), "count", out["count"], "ref", ref["count"], "status", out["status"])
    print(" ps", out["pscore"][0], ref["pscore"][0][:3])
    print(" xyzs[0,0,:3]", out["xyzs"][0, 0, :3], "\n ref", ref["xyz"][0, 0, :3], ref["kscore"][0, 0, :3])
    print(" xyzs[2,1,130:]", out["xyzs"][2, 1, 130:], "\n ref", ref["xyz"][2, 1, 130:], ref["kscore"][2, 1, 130:])
    bt.close()
wl = synth.config_workload(3, 4)
print("cfg3 params", wl["params"], wl["kpts"].shape)
print("kp[0,:,:,0]", kp[0, :, :, 0], "npers", npers[0])
print("M0", (R[0] @ np.linalg.inv(K[0]))[0], "t", t.reshape(C, 3)[:2])

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import snowmocap_amd as api
from snowmocap_amd import synth
from oracle import oracle as orc
J = 133
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 7117)
for trial in range(36):
    C = int(rng.choice([5, 6, 7, 8, 9, 12, 16])); F = int(rng.choice([3, 40, 150]))
    out_dtype = np.float64 if rng.uniform() < 0.5 else np.float32
    kn = J if rng.uniform() < 0.6 else int(rng.integers(1, J)); pout = int(rng.choice([1, 1, 2, 3]))
    K, R, t = synth.ring_rig(C, radius=float(rng.uniform(3.5, 6)))
    X = synth.make_people(rng, F, 1)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=float(rng.choice([0.3, 1.0, 3.0])), score_range=(2.0, 8.0), dtype=np.float64 if trial % 5 == 0 else np.float32)
    kp, npers = kp.copy(), npers.copy()
    for _ in range(int(rng.integers(0, 3))): npers[rng.integers(0, F), rng.integers(0, C)] = 0
    nanpos = None
    if rng.uniform() < 0.3:
        nanpos = (int(rng.integers(0, F)), int(rng.integers(0, C)), int(rng.integers(0, J))); kp[nanpos[0], nanpos[1], 0, nanpos[2], 0] = np.nan
    if rng.uniform() < 0.3: kp[rng.integers(0, F), :, 0, :, 2] = 0.0
    prm = dict(keypoint_score_threshold=float(rng.choice([0.0, 3.0, 5.0])), average_score_threshold=float(rng.choice([0.0, 0.0, 0.5])),
               distance_threshold=float(rng.choice([0.02, 0.05, 1.0])), condense_distance_tol=float(rng.choice([0.05, 0.5, 10.0])),
               condense_person_num_tol=int(rng.choice([0, 2, C * (C - 1) // 2])), condense_score_tol=float(rng.choice([0.0, 0.0, 0.6])),
               center_point_index=int(rng.integers(0, kn)), keypoint_num=kn)
    if trial != int(sys.argv[1]): continue
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), max(pout, 4))
    print("trial", trial, "C", C, "F", F, out_dtype, "kn", kn, "pout", pout, "nan at", nanpos, prm)
    for mode in (None, "2", "1"):
        if mode: os.environ["SNOWTRI_GENERAL_MODE"] = mode
        bt = api.BatchTriangulator(K, R, t, prm, pout_max=pout, out_dtype=out_dtype)
        out = bt.run_host(kp, npers); names = bt.ctx.last_kernel_names(); bt.close()
        os.environ.pop("SNOWTRI_GENERAL_MODE", None)
        f = int(sys.argv[2])
        m = min(int(ref["count"][f]), pout)
        g, w = out["xyzs"][f, :m, :, 3], ref["kscore"][f, :m]
        print(mode, names[:70], "count", out["count"][f], ref["count"][f], "nan got", np.argwhere(np.isnan(g)).tolist(), "nan ref", np.argwhere(np.isnan(w)).tolist(), "flags", out["flags"][f], "pscore", out["pscore"][f, :m], ref["pscore"][f, :m])
        rel = np.abs(g - w) / np.maximum(np.abs(w), 1e-300)
        j = int(np.nanargmax(rel)); print("   worst joint", j % g.shape[1], "got %.17g ref %.17g rel %.3g" % (g.flat[j], w.flat[j], rel.flat[j]), "xyz err", np.abs(out["xyzs"][f, :m, :, :3] - ref["xyz"][f, :m]).max())
        print("   npers", npers[f].tolist())

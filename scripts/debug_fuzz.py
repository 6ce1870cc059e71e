#!/usr/bin/env python3
"""Dev aid (GPU box): replay the randomised rigs of tests/test_gpu_parity.py::test_random_small_rigs_against_oracle
and print the first trials where the fused entry and the oracle disagree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import snowmocap_amd as api
from snowmocap_amd import synth
from oracle import oracle as orc
rng = np.random.default_rng(2024)
bad = 0
for trial in range(60):
    C = int(rng.integers(2, 7)); P = int(rng.integers(1, 4)); J = int(rng.choice([3, 5, 20, 33, 40])); F = int(rng.integers(1, 6))
    K, R, t = synth.ring_rig(C, radius=float(rng.uniform(3, 6)))
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=float(rng.choice([0.3, 1.0, 3.0])), score_range=(2.0, 8.0),
                                     permute_persons=True, dtype=np.float64 if trial % 2 else np.float32)
    npers = npers.copy()
    for _ in range(int(rng.integers(0, 4))):
        npers[rng.integers(0, F), rng.integers(0, C)] = rng.integers(0, P + 1)
    kn = int(rng.integers(1, J + 1))
    prm = dict(keypoint_score_threshold=float(rng.choice([0.0, 3.0, 3.0, 5.0])), average_score_threshold=float(rng.choice([0.0, 0.0, 0.3, 1.5])),
               distance_threshold=float(rng.choice([0.02, 0.05, 1.0])), condense_distance_tol=float(rng.choice([0.05, 0.3, 10.0])),
               condense_person_num_tol=int(rng.choice([0, 0, 1, 2])), condense_score_tol=float(rng.choice([0.0, 0.0, 0.3, 2.0])),
               center_point_index=int(rng.integers(0, J)), keypoint_num=kn)
    pout = int(rng.choice([1, 4, 16]))
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    bt = api.BatchTriangulator(K, R, t, prm, pout_max=pout, out_dtype=np.float64)
    out = bt.run_host(kp, npers)
    bt.close()
    ok = np.array_equal(out["count"], ref["count"])
    worst = 0.0
    for f in range(F):
        m = min(int(ref["count"][f]), pout)
        if m and ok:
            ds = np.abs(out["xyzs"][f, :m, :, 3] - ref["kscore"][f, :m]) / (np.abs(ref["kscore"][f, :m]) + 1e-300)
            dx = np.abs(out["xyzs"][f, :m, :, :3] - ref["xyz"][f, :m])
            worst = max(worst, float(np.nanmax(ds)), float(np.nanmax(dx)))
    if not ok or worst > 1e-6:
        bad += 1
        print(f"trial {trial}: C={C} P={P} J={J} kn={kn} F={F} pout={pout} dtype={kp.dtype} prm={prm}")
        print("  npers", npers.tolist(), "count gpu", out["count"].tolist(), "oracle", ref["count"].tolist(), "flags", out["flags"].tolist(), "worst", worst)
        for f in range(F):
            m = min(int(ref["count"][f]), pout)
            if m:
                print("  f", f, "gpu score[0,:4]", out["xyzs"][f, 0, :4, 3], "oracle", ref["kscore"][f, 0, :4])
                print("       gpu xyz[0,0]", out["xyzs"][f, 0, 0, :3], "oracle", ref["xyz"][f, 0, 0])
                break
        if bad >= 4:
            break
print("bad trials:", bad)

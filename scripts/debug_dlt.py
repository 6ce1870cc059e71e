import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snowmocap_amd import synth, _lib
from snowmocap_amd.batch import BatchTriangulator
from oracle import dlt
wl = synth.config_workload(2, 40, seed=9)
kp = wl["kpts"]
K, R, t = wl["rig"]
prm = dict(wl["params"])
for variant in ("plain", "gated"):
    if variant == "gated":
        kp[:, :, :, :, 2] = np.random.default_rng(3).uniform(2.0, 8.0, size=kp.shape[:-1]).astype(np.float32)
    want, wps, wcnt = dlt.dlt_batch(K, R, t, kp, prm["keypoint_score_threshold"], prm["keypoint_num"])
    bt = BatchTriangulator(K, R, t, prm, pout_max=1, out_dtype=np.float64, method=_lib.DLT)
    out = bt.run_host(kp, wl["n_persons"])
    bt.close()
    x = out["xyzs"][:, 0, :, :3]
    nanmask = ~np.isfinite(x).all(axis=-1)
    print(variant, "nan joints:", int(nanmask.sum()), "of", nanmask.size)
    fj = np.argwhere(nanmask)[:5]
    for f, j in fj:
        print("  f,j", f, j, "scores", kp[f, :, 0, j, 2], "got", out["xyzs"][f, 0, j], "want", want[f, 0, j])
    ok = ~nanmask
    print("  max err on finite:", np.abs(x[ok] - want[:, 0, :, :3][ok]).max())

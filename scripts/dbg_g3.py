import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_scenarios, ALL_SCENARIO_FILES
import snowmocap_amd as sm
scs = {}
for fn in ALL_SCENARIO_FILES:
    for s, sc in load_scenarios(fn).items():
        scs[f"{fn[:2]}-{s}"] = sc
sc = scs["g3-ring8x4"]
print("kpts", sc["kpts"].shape, "np", sc["n_persons"], "params", sc["params"], "cond_n", sc["cond_n"])
res = {}
for hm in ("1", "2"):
    os.environ["SNOWTRI_HANDOVER_MODE"] = hm
    pout = max(1, sc["cond_xyz"].shape[1])
    bt = sm.BatchTriangulator(sc["K"], sc["R"], sc["t"], sc["params"], pout_max=pout, out_dtype=np.float32)
    out = bt.run_host(sc["kpts"], sc["n_persons"])
    print("mode", hm, "handed", bt.ctx.last_handover_persons(), "count", out["count"], "flags", out["flags"])
    print(" pscore", out["pscore"])
    bt.close()
    res[hm] = out
print("ref pscore", sc["cond_pscore"])
d = np.abs(res["1"]["xyzs"].astype(np.float64) - res["2"]["xyzs"]).max(axis=(2, 3))
print("max |xyzs mode1 - mode2| per (frame, slot)", d)

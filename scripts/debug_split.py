#!/usr/bin/env python3
"""Dev aid (GPU box): where does a 125 000-frame launch of the fast path differ from the same frames in other launch shapes?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import snowmocap_amd as api
from test_gpu_shard_sizes import _tiled_shard

F = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
wl, kp, npers = _tiled_shard(2, F, 25000, seed=31)
K, R, t = wl["rig"]
bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float32)

def full(env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    o = bt.run_torch(kp); torch.cuda.synchronize()
    for k in (env or {}):
        os.environ.pop(k)
    return o["xyzs"].cpu().numpy().copy()

def pieces(cuts):
    xs = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        o = bt.run_torch(kp[lo:hi].contiguous()); torch.cuda.synchronize()
        xs.append(o["xyzs"].cpu().numpy())
    return np.concatenate(xs)

def diff(name, a, b):
    d = np.nonzero(np.any(a.reshape(F, -1) != b.reshape(F, -1), axis=1))[0]
    print(f"{name}: {len(d)} frames differ {d[:8].tolist()}")

a = full()
diff("full vs full again", a, full())
c1 = [0, 1, F // 3, F // 3 + 1, (4 * F) // 5, F]
p1 = pieces(c1)
diff("full vs pieces", a, p1)
diff("pieces vs pieces again", p1, pieces(c1))
for tpw in ("1", "2", "8", "16"):
    diff(f"full vs full tiles_per_wave={tpw}", a, full({"SNOWTRI_LEAN_TILES_PER_WAVE": tpw}))
half = pieces([0, F // 2, F])
diff("full vs halves", a, half)
diff("pieces vs halves", p1, half)

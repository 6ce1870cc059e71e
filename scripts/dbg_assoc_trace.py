"""Dev aid (GPU box, SNOWTRI_LIB = a -DSNOWTRI_ASSOC_TRACE build): where k_associate spends a frame -- wall-clock stamps
(100 MHz) of every workgroup's second frame at the phase boundaries."""
import ctypes as ct, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snowmocap_amd import synth, _lib
from snowmocap_amd.batch import BatchTriangulator
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
F, gen, pout = (10000, 1000, 16) if cfg == 3 else (12000, 250, 32)
os.environ["SNOWTRI_SPLIT_SEGMENTS"] = "1"   # whole segments: workgroups run a second frame (the traced one)
wl = synth.config_workload(cfg, gen)
K, R, t = wl["rig"]
dev = torch.device("cuda", 0)
kp = torch.from_numpy(wl["kpts"]).to(dev).repeat(F // gen, 1, 1, 1, 1).contiguous()
npers = torch.from_numpy(wl["n_persons"]).to(dev).repeat(F // gen, 1).contiguous()
bt = BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=np.float32)
out = bt.run_torch(kp, npers)
for _ in range(3):
    bt.run_torch(kp, npers, out=out)
torch.cuda.synchronize()
L = ct.CDLL(_lib.LIB_PATH)
raw = np.zeros(4096 * 16, dtype=np.uint64)
assert L.snowtri_debug_read_assoc_trace(raw.ctypes.data_as(ct.c_void_p)) == 0
st = raw.reshape(4096, 16)[:, :9].astype(np.int64)
st = st[(st > 0).all(axis=1)]
if len(st) == 0:
    sys.exit("no workgroup ran a second frame (too few frames for the grid): nothing traced")
us = (st - st[:, :1]) / 100.0
names = ["frame start", "ragged check done", "kept list built", "centres solved", "clustered + grouped", "filters done", "list room reserved (atomics)", "descriptors written", "zero-fill + count written"]
print("workgroups traced:", len(st))
prev = np.zeros(len(st))
for i, n in enumerate(names):
    v = us[:, i]
    print("%-32s at median %7.2f us  (+%6.2f)   p95 %7.2f" % (n, np.median(v), np.median(v - prev), np.percentile(v, 95)))
    prev = v

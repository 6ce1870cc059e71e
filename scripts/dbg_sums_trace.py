"""Dev aid (GPU box, SNOWTRI_LIB = a -DSNOWTRI_SUMS_TRACE build): where k_candidate_sums spends its time -- wall-clock stamps
(100 MHz) of the first four waves of every workgroup: kernel entry / exit, and the phase boundaries of the workgroup's second
frame; wave placement from HW_ID.   usage: dbg_sums_trace.py [cfg 3|5]"""
import ctypes as ct, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snowmocap_amd import synth, _lib
from snowmocap_amd.batch import BatchTriangulator
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
F, gen, pout = (10000, 1000, 16) if cfg == 3 else (4000, 250, 32)
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
raw = np.zeros(4096 * 4 * 16, dtype=np.uint64)
assert L.snowtri_debug_read_sums_trace(raw.ctypes.data_as(ct.c_void_p)) == 0
st = raw.reshape(4096 * 4, 16).astype(np.int64)
live = st[:, 12] > 0
st = st[live]
print("waves traced:", len(st))
t0 = st[:, 12].min()
entry, exit_ = (st[:, 12] - t0) / 100.0, (st[:, 13] - t0) / 100.0
print("kernel: first entry 0, last exit %.1f us; entry median %.1f p95 %.1f max %.1f; exit min %.1f p5 %.1f median %.1f" % (
    exit_.max(), np.median(entry), np.percentile(entry, 95), entry.max(), exit_.min(), np.percentile(exit_, 5), np.median(exit_)))
print("wave lifetime / kernel span: mean %.3f" % ((exit_ - entry).mean() / exit_.max()))
hw = st[:, 14]
cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)   # xcc, se, sh, cu
u, n = np.unique(cu, return_counts=True)
print("distinct CUs %d; waves per CU: min %d median %d max %d" % (len(u), n.min(), np.median(n), n.max()))
idx = [0, 1, 10, 11]
fr = st[(st[:, idx] > 0).all(axis=1)]
if len(fr) == 0:
    sys.exit("no workgroup ran a second frame")
us = (fr[:, idx] - fr[:, :1]) / 100.0
names = ["frame start", "first chunk written (+ barrier where the frame has them)", "last chunk solved", "frame end"]
prev = np.zeros(len(fr))
for i, nme in enumerate(names):
    v = us[:, i]
    print("%-58s at median %7.2f us  (+%6.2f, p95 +%6.2f)" % (nme, np.median(v), np.median(v - prev), np.percentile(v - prev, 95)))
    prev = v

# the third joint chunk of that frame, wave by wave (its keypoint requests are in front of stamp 2): solve | write the next chunk's records | barrier
idx = [2, 3, 4, 6]
ch = st[(st[:, idx] > 0).all(axis=1)]
if len(ch):
    d = np.diff(ch[:, idx], axis=1) / 100.0
    for i, nme in enumerate(["solve", "commit (records of the next chunk)", "barrier wait"]):
        print("chunk 2: %-40s median %6.2f us  p5 %6.2f  p95 %6.2f" % (nme, np.median(d[:, i]), np.percentile(d[:, i], 5), np.percentile(d[:, i], 95)))
    tot = (ch[:, 6] - ch[:, 2]) / 100.0
    print("chunk 2: whole                                    median %6.2f us" % np.median(tot))
    wv = np.arange(len(st))[(st[:, idx] > 0).all(axis=1)] % 4
    for w in range(4):
        print("  wave %d of its workgroup: solve median %6.2f us, barrier wait %6.2f us" % (w, np.median(d[wv == w, 0]), np.median(d[wv == w, 2])))

if os.environ.get("TRACE_MODE") == "2":   # a -DSNOWTRI_SUMS_TRACE=2 build: slots 2 .. 9 = the end of chunk 0 .. 7 of that frame
    idx = [1] + list(range(2, 10))
    ok = st[(st[:, [0, 1, 2]] > 0).all(axis=1)]
    d = np.diff(ok[:, idx], axis=1) / 100.0
    for c in range(8):
        v = d[:, c][ok[:, 2 + c] > 0]
        if len(v):
            print("chunk %d: median %6.2f us  p5 %6.2f  p95 %6.2f" % (c, np.median(v), np.percentile(v, 5), np.percentile(v, 95)))

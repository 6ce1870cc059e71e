"""Dev aid (GPU box, SNOWTRI_LIB = a -DSNOWTRI_LEAN_TRACE build): where a 10 000-frame launch of k_fused_lean_coop spends
its time -- wall-clock stamps (100 MHz) of every wave at the phase boundaries, relative to the launch's first stamp."""
import ctypes as ct, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snowmocap_amd import synth, _lib
from snowmocap_amd.batch import BatchTriangulator
F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
wl = synth.config_workload(2, F, seed=1)
K, R, t = wl["rig"]
dev = torch.device("cuda", 0)
bt = BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float32)
pool = [torch.from_numpy(wl["kpts"]).to(dev) + 0.01 * i for i in range(12)]
out = bt.alloc_outputs(F, dev)
for i in range(40):
    bt.run_torch(pool[i % 12], None, out=out)
torch.cuda.synchronize()
L = ct.CDLL(_lib.LIB_PATH)
per_block = 6 * 133 * 8 * 4 + 1024            # >= general_scratch_bytes: read a generous prefix per block
grid = min(F, 512)
# slab stride: ask the kernel's own layout -- stamps sit at the start of each slab; find the stride by scanning
raw = np.zeros(64 << 20, dtype=np.uint8)
L.snowtri_debug_read_scratch.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_size_t]
rc = L.snowtri_debug_read_scratch(bt.ctx.handle, raw.ctypes.data_as(ct.c_void_p), grid * 25600)
assert rc == 0, rc
stride = 25600        # general_scratch_bytes(6 pairs, 133 joints)
st = np.stack([raw[b * stride: b * stride + 4 * 8 * 8].view(np.uint64).reshape(4, 8) for b in range(grid)]).astype(np.int64)   # [block, wave, stamp]
hw = st[:, :, 7]
st = st[:, :, :7]
t0 = st[:, :, 0].min()
us = (st - t0) / 100.0
names = ["entry", "constants + table written", "after barrier 1", "items done", "after barrier 2", "epilogue done", "after barrier 3 (outputs written)"]
for i, n in enumerate(names):
    v = us[:, :, i]
    print("%-36s min %6.2f  median %6.2f  p95 %6.2f  max %6.2f us" % (n, v.min(), np.median(v), np.percentile(v, 95), v.max()))
print("items phase per wave: median %.2f us, min %.2f, max %.2f" % (np.median(us[:, :, 3] - us[:, :, 2]), (us[:, :, 3] - us[:, :, 2]).min(), (us[:, :, 3] - us[:, :, 2]).max()))

# placement: HW_ID bits: wave_id 3:0, simd_id 5:4, pipe 7:6, cu_id 11:8, sh_id 12, se_id 15:13 (gfx9); XCC_ID 3:0
xcc = (hw >> 32) & 15
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
import collections
per_cu = collections.Counter(cuid[:, 0].tolist())
print("workgroups per CU:", sorted(collections.Counter(per_cu.values()).items()), "distinct CUs", len(per_cu))
simdkey = cuid * 4 + simd
per_simd = collections.Counter(simdkey.flatten().tolist())
print("waves per SIMD:", sorted(collections.Counter(per_simd.values()).items()))
dur = us[:, :, 3] - us[:, :, 2]
for n in sorted(set(per_simd.values())):
    sel = np.array([per_simd[k] == n for k in simdkey.flatten()]).reshape(simdkey.shape)
    print("  SIMDs with %d wave(s): items phase median %.2f us, end of items median %.2f max %.2f" % (n, np.median(dur[sel]), np.median(us[:, :, 3][sel]), us[:, :, 3][sel].max()))
for x in range(8):
    sel = xcc == x
    if sel.any(): print("  XCC %d: waves %d, items phase median %.2f, end median %.2f max %.2f" % (x, sel.sum(), np.median(dur[sel]), np.median(us[:, :, 3][sel]), us[:, :, 3][sel].max()))

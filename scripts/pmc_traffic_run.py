#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (scripts/profile.sh): the bench's 10 000-frame fused launch, and beside it two
calibration streams of KNOWN size with the same access shapes (snowtri_calib_stream: 12-byte records read per lane,
16-byte records written per lane) so that FETCH_SIZE / WRITE_SIZE are scaled on kernels other than the one measured.
Every launch works on its own HBM-resident buffers (a pool larger than the 256 MB Infinity Cache)."""
import ctypes as ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from snowmocap_amd import synth, _lib
from snowmocap_amd.batch import BatchTriangulator

F, POOL, REPS = 10000, 16, 3
wl = synth.config_workload(2, F, seed=1000)
K, R, t = wl["rig"]
dev = torch.device("cuda", 0)
bt = BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float32)
base = torch.from_numpy(wl["kpts"]).to(dev)
pool = [(base + 0.01 * i).contiguous() for i in range(POOL)]
outs = [bt.alloc_outputs(F, dev) for _ in range(POOL)]
L = _lib.lib()
st = ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
rb, wb = pool[0].numel() * 4, outs[0]["xyzs"].numel() * 4
for rep in range(REPS):
    for i in range(POOL):
        _lib.check(L.snowtri_calib_stream(bt.ctx.handle, ct.c_void_p(pool[i].data_ptr()), rb, ct.c_void_p(outs[i]["xyzs"].data_ptr()), wb, st),
                   "snowtri_calib_stream")
    for i in range(POOL):
        bt.run_torch(pool[i], None, out=outs[i])
torch.cuda.synchronize()
print("known_bytes", rb, wb, "frames", F)

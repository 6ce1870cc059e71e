#!/usr/bin/env python3
"""Dev aid (GPU box): run the -DSNOWTRI_ITEM_TIMERS build (SNOWTRI_LIB=...) and print the per-stage cycle counts of
the fast kernel's item (s_memtime ticks, 100 MHz constant clock on gfx9: x (core clock / 100 MHz) core cycles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snowmocap_amd import synth
from snowmocap_amd.batch import BatchTriangulator
for F in (20, 5120, 10240, 40960):
    wl = synth.config_workload(2, F)
    K, R, t = wl["rig"]
    os.environ["SNOWTRI_TILE_FRAMES"] = "20"
    bt = BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float64)
    kp = torch.from_numpy(wl["kpts"]).cuda()
    out = bt.run_torch(kp); torch.cuda.synchronize()
    out = bt.run_torch(kp); torch.cuda.synchronize()
    x = out["xyzs"].cpu().numpy()[:, 0]            # [F,133,4]
    rays, pairs, tail, start = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    entry = out["count"].cpu().numpy().astype(np.int64)
    fin = out["pscore"].cpu().numpy()[:, 0]
    start = np.mod(start, 2.0 ** 31)
    # per-tile span: frames f0..f0+19 share a workgroup; items of one lane are 256 apart
    span, pro, epi, tot = [], [], [], []
    for f0 in range(0, F, 20):
        st = start[f0:f0 + 20].ravel()
        span.append((st.max() - st.min()))
        pro.append(st.min() - entry[f0])
        epi.append(fin[f0:f0 + 20].max() - st.max())
        tot.append(fin[f0:f0 + 20].max() - entry[f0])
    print(f"     tile: entry->first item {np.median(pro):.0f}  first->last item start {np.median(span):.0f}  last item start->epilogue end {np.median(epi):.0f}  entry->end {np.median(tot):.0f} ticks")
    print(f"F={F}: ticks per item  const wait {rays.mean():.1f}  rays+keypoint wait {pairs.mean():.1f}  pairs+tail {tail.mean():.1f}  sum {(rays+pairs+tail).mean():.1f} | "
          f"first-to-last item start within a tile: mean {np.mean(span):.0f} ticks (11 rounds)")
    bt.close()

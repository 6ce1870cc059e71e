#!/usr/bin/env python3
"""GPU box: what a K-step timed region of the headline loop costs beyond K steps.  Regions of K = 5 ... 200 calls of the
configs[1] batch (10 000 frames) between two device synchronisations, one stream and the overlap mode; least-squares line
region = slope x K + intercept.  The intercept (first launch latency + pipeline fill and drain + the host's wake-up from
the synchronize) was 19 us on one stream and 33-75 us in overlap mode depending on the box: 8-17 % of the driver's
`--steps 20` region, 1-2 % of the default `--steps 200`.  usage: python scripts/region_fit.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from snowmocap_amd import synth
from snowmocap_amd.batch import BatchTriangulator
dev = torch.device("cuda", 0)
wl = synth.config_workload(2, 10000)
K_, R_, t_ = wl["rig"]
F = 10000
base = torch.from_numpy(wl["kpts"]).to(dev)
pool = [base] + [(base + 0.0).contiguous() for _ in range(15)]
def run(nstreams, Ks):
    bt = BatchTriangulator(K_, R_, t_, wl["params"], pout_max=1, out_dtype=np.float32, streams=nstreams)
    outs = [bt.alloc_outputs(F, dev) for _ in pool]
    def region(K):
        for i in range(5):
            bt.run_torch(pool[i % 16], None, out=outs[i % 16])
        bt.join(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(K):
            bt.run_torch(pool[i % 16], None, out=outs[i % 16])
        t1 = time.perf_counter()
        bt.join()
        t2 = time.perf_counter()
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        return (t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.15:
        region(50)
    res = {}
    for K in Ks:
        r = np.array([region(K) for _ in range(15)])
        res[K] = np.median(r, axis=0)
        print(f"streams {nstreams} K {K:4d}: region {res[K][0]:8.1f} us = {res[K][0]/K:6.2f} us/step | issue {res[K][1]:7.1f} join {res[K][2]:5.1f} sync {res[K][3]:7.1f}", flush=True)
    ks = np.array(list(res)); ts = np.array([res[k][0] for k in res])
    A = np.vstack([ks, np.ones_like(ks)]).T
    slope, icpt = np.linalg.lstsq(A, ts, rcond=None)[0]
    print(f"streams {nstreams}: slope {slope:.2f} us/step, intercept {icpt:.1f} us")
    bt.close()
for ns in (1, 2):
    run(ns, (5, 10, 20, 40, 80, 200))

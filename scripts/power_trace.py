#!/usr/bin/env python3
"""GPU box: socket power and shader clock WHILE the kernels run (missing #5: the 1.5 GHz sustained clock behind
DESIGN.md's "power-limited fp64 issue roof" was inferred from GRBM_GUI_ACTIVE / duration; this is the direct record).

A sampler thread polls amdsmi (gpu_metrics: current_gfxclk(s), socket power, throttle status; power_info / clock_info as
a fall-back) every few milliseconds while the main thread keeps ONE workload running for a few seconds:

    idle | fp64 fma loop | the pair solve's instruction mix | fma loop + 4 TB/s of streaming reads
    | one 2 000 000-frame launch of the fast kernel, back to back | the bench's 10 000-frame launches on two streams
    | the 8 x 4 and 16 x 8 multi-person calls

and writes gpurun_out/power/power_trace.json (per workload: median / p10 / p90 of clock and power over the steady part,
the achieved rate) + samples.csv.   usage: gpurun -- python scripts/power_trace.py [seconds per workload]
"""
import ctypes as ct
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

OUT = os.path.join(ROOT, "gpurun_out", "power")
os.makedirs(OUT, exist_ok=True)
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


class Sampler(threading.Thread):
    def __init__(self, period=0.004):
        super().__init__(daemon=True)
        self.period = period
        self.samples = []      # (t, label, gfxclk MHz, power W, extra)
        self.label = "idle"
        self.stop = False
        self.src = None
        self.err = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.smi = amdsmi
            self.h = amdsmi.amdsmi_get_processor_handles()[0]
            self.src = "amdsmi"
        except Exception as e:   # noqa: BLE001
            self.err = repr(e)
            self.smi = None

    def read(self):
        smi = self.smi
        clk = pw = None
        extra = {}
        try:
            m = smi.amdsmi_get_gpu_metrics_info(self.h)
            g = m.get("current_gfxclks") or m.get("current_gfxclk")
            if isinstance(g, (list, tuple)):
                vals = [float(x) for x in g if isinstance(x, (int, float)) and 0 < x < 60000]
                if vals:
                    clk = float(np.mean(vals))
                    extra["gfxclk_min"] = min(vals)
                    extra["gfxclk_max"] = max(vals)
            elif isinstance(g, (int, float)) and 0 < g < 60000:
                clk = float(g)
            for key in ("current_socket_power", "average_socket_power"):
                v = m.get(key)
                if isinstance(v, (int, float)) and 0 < v < 5000:
                    pw = float(v)
                    extra["power_key"] = key
                    break
            for key in ("throttle_status", "indep_throttle_status", "temperature_hotspot", "average_gfx_activity", "average_umc_activity",
                        "current_uclk", "accumulation_counter"):
                v = m.get(key)
                if isinstance(v, (int, float)):
                    extra[key] = v
        except Exception as e:   # noqa: BLE001
            extra["metrics_err"] = repr(e)[:80]
        if pw is None:
            try:
                p = smi.amdsmi_get_power_info(self.h)
                for key in ("current_socket_power", "average_socket_power", "socket_power"):
                    v = p.get(key)
                    if isinstance(v, (int, float)) and 0 < v < 5000:
                        pw = float(v)
                        extra["power_key"] = "power_info." + key
                        break
            except Exception as e:   # noqa: BLE001
                extra["power_err"] = repr(e)[:80]
        if clk is None:
            try:
                c = smi.amdsmi_get_clock_info(self.h, smi.AmdSmiClkType.SYS)
                v = c.get("clk") or c.get("cur_clk")
                if isinstance(v, (int, float)) and v > 0:
                    clk = float(v)
            except Exception as e:   # noqa: BLE001
                extra["clock_err"] = repr(e)[:80]
        return clk, pw, extra

    def run(self):
        if self.smi is None:
            return
        while not self.stop:
            t = time.perf_counter()
            clk, pw, extra = self.read()
            self.samples.append((t, self.label, clk, pw, extra))
            dt = self.period - (time.perf_counter() - t)
            if dt > 0:
                time.sleep(dt)


def stats(xs):
    xs = [x for x in xs if x is not None]
    if not xs:
        return None
    a = np.asarray(xs, dtype=np.float64)
    return {"median": float(np.median(a)), "p10": float(np.percentile(a, 10)), "p90": float(np.percentile(a, 90)),
            "mean": float(a.mean()), "n": int(a.size)}


def main():
    import torch
    from snowmocap_amd import synth
    from snowmocap_amd.batch import BatchTriangulator
    dev = torch.device("cuda", 0)
    sampler = Sampler()
    sampler.start()
    results = []
    t_base = time.perf_counter()

    def run(label, fn, what):
        """fn(seconds) -> dict of achieved rates; the sampler tags its samples with `label` meanwhile."""
        torch.cuda.synchronize(dev)
        sampler.label = label
        t0 = time.perf_counter()
        r = fn(SECONDS)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        sampler.label = "gap"
        # steady part: the second half of the interval
        mid = 0.5 * (t0 + t1)
        ss = [s for s in sampler.samples if s[1] == label and s[0] >= mid]
        res = {"workload": label, "what": what, "seconds": t1 - t0, "gfxclk_MHz": stats([s[2] for s in ss]),
               "socket_power_W": stats([s[3] for s in ss]), "achieved": r}
        if ss:
            res["last_sample_extra"] = ss[-1][4]
        results.append(res)
        print(json.dumps(res), flush=True)
        time.sleep(0.5)

    run("idle", lambda s: (time.sleep(min(s, 1.0)), {})[1], "no work")

    # ---- fp64 issue loops (scripts/ubench/fp64_burn.hip)
    so = "/tmp/libfp64_burn.so"
    burn = None
    if subprocess.call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-w", "-o", so,
                        os.path.join(ROOT, "scripts", "ubench", "fp64_burn.hip")]) == 0:
        burn = ct.CDLL(so)
        burn.fp64_burn.restype = ct.c_double
        burn.fp64_burn.argtypes = [ct.c_int, ct.c_double, ct.c_int, ct.c_int64]
    if burn is not None:
        def mk(mode, wg, sb):
            def f(s):
                v = burn.fp64_burn(mode, s, wg, sb)
                return {"valu_wave_insts_per_s_per_simd": v, "issue_clock_GHz_if_4_cycles_each": v * 4 / 1e9}
            return f
        run("fp64_fma", mk(0, 2, 0), "v_fma_f64 only, 8 chains per lane, 2 waves per SIMD")
        run("fp64_fma_3waves", mk(0, 3, 0), "v_fma_f64 only, 3 waves per SIMD")
        run("fp64_solve_mix", mk(1, 2, 0), "40 v_fma_f64 per v_rsq_f64 + v_rcp_f64 (transcendentals counted as one instruction each)")
        run("fp64_fma_plus_stream", mk(2, 2, 4 << 30), "v_fma_f64 (128 per 16 B read per lane) beside a streaming read of 4 GiB per launch")

    # ---- the fast kernel
    J = 133
    wl = synth.config_workload(2, 10000, seed=1000)
    K, R, t = wl["rig"]
    bt = [BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float32) for _ in range(2)]
    base = torch.from_numpy(wl["kpts"]).to(dev)
    pool = [base] + [(base + torch.randn_like(base) * torch.tensor([0.25, 0.25, 0.0], device=dev)).contiguous() for _ in range(15)]
    outs = [bt[0].alloc_outputs(10000, dev) for _ in pool]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def bench_small(s):
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < s:
            for i in range(400):
                bt[i & 1].run_torch(pool[i % 16], None, out=outs[i % 16], stream=streams[i & 1].cuda_stream)
            torch.cuda.synchronize(dev)
            n += 400
        dt = time.perf_counter() - t0
        return {"joints_per_s": n * 10000 * J / dt, "us_per_launch": dt / n * 1e6, "frac_of_8TBs": n * 10000 * 8512 / dt / 8e12}
    run("lean_10k_two_streams", bench_small, "k_fused_lean_coop, 10 000-frame launches alternating on two streams (the bench's `value`)")

    FL = 2000000
    big = torch.cat([pool[i % 16] for i in range(FL // 10000)], dim=0).contiguous()
    bout = bt[0].alloc_outputs(FL, dev)

    def bench_large(s):
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < s:
            for _ in range(8):
                bt[0].run_torch(big, None, out=bout)
            torch.cuda.synchronize(dev)
            n += 8
        dt = time.perf_counter() - t0
        return {"joints_per_s": n * FL * J / dt, "ms_per_launch": dt / n * 1e3, "frac_of_8TBs": n * FL * 8512 / dt / 8e12}
    run("lean_2M", bench_large, "k_fused_lean, 2 000 000-frame launches back to back on one stream")
    del big, bout
    for b in bt:
        b.close()

    # ---- the multi-person path
    for cfg, F, gen, pout, label in ((3, 10000, 1000, 16, "multi_8x4"), (5, 12500, 250, 32, "multi_16x8")):
        wl = synth.config_workload(cfg, gen)
        K, R, t = wl["rig"]
        kp = torch.from_numpy(wl["kpts"]).to(dev).repeat(F // gen, 1, 1, 1, 1).contiguous()
        npers = torch.from_numpy(wl["n_persons"]).to(dev).repeat(F // gen, 1).contiguous()
        b = BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=np.float32)
        out = b.run_torch(kp, npers)
        torch.cuda.synchronize(dev)

        def bench_multi(s, b=b, kp=kp, npers=npers, out=out, F=F):
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < s:
                for _ in range(4):
                    b.run_torch(kp, npers, out=out)
                torch.cuda.synchronize(dev)
                n += 4
            dt = time.perf_counter() - t0
            return {"frames_per_s": n * F / dt, "ms_per_call": dt / n * 1e3}
        run(label, bench_multi, f"config {cfg}: {F} frames per call, calls back to back on one stream")
        b.close()
        del kp, npers, out

    # ---- one detection per camera on 8 cameras (the lean kernel on the complete-graph item: 28 pair solves per joint)
    rng = np.random.default_rng(8)
    K, R, t = synth.ring_rig(8)
    X = synth.make_people(rng, 500, 1)
    kp8, np8 = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(3.5, 8.0))
    F8 = 200000
    kp8 = torch.from_numpy(kp8).to(dev).repeat(F8 // 500, 1, 1, 1, 1).contiguous()
    np8 = torch.from_numpy(np8).to(dev).repeat(F8 // 500, 1).contiguous()
    b8 = BatchTriangulator(K, R, t, synth.default_thresholds(), pout_max=1, out_dtype=np.float32)
    out8 = b8.run_torch(kp8, np8)
    torch.cuda.synchronize(dev)

    def bench_single8(s):
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < s:
            for _ in range(8):
                b8.run_torch(kp8, np8, out=out8)
            torch.cuda.synchronize(dev)
            n += 8
        dt = time.perf_counter() - t0
        return {"joints_per_s": n * F8 * J / dt, "ms_per_call": dt / n * 1e3, "frac_of_fp64_peak": n * F8 * J * (28 * 90 + 8 * 15) / dt / 78.6e12}
    run("single_8x1_200k", bench_single8, "k_fused_lean<8,float,133>: 200 000-frame launches back to back on one stream")
    b8.close()

    sampler.stop = True
    sampler.join(timeout=2.0)
    with open(os.path.join(OUT, "samples.csv"), "w") as fh:
        fh.write("t_s,workload,gfxclk_MHz,socket_power_W\n")
        for s in sampler.samples:
            fh.write(f"{s[0] - t_base:.4f},{s[1]},{'' if s[2] is None else round(s[2], 1)},{'' if s[3] is None else round(s[3], 1)}\n")
    summary = {"sampler": {"source": sampler.src, "error": sampler.err, "period_s": sampler.period, "samples": len(sampler.samples)},
               "seconds_per_workload": SECONDS, "device": torch.cuda.get_device_name(dev), "workloads": results}
    try:
        summary["power_cap"] = {k: v for k, v in sampler.smi.amdsmi_get_power_cap_info(sampler.h).items() if isinstance(v, (int, float))}
    except Exception as e:   # noqa: BLE001
        summary["power_cap"] = repr(e)[:120]
    json.dump(summary, open(os.path.join(OUT, "power_trace.json"), "w"), indent=1)
    print("wrote", os.path.join(OUT, "power_trace.json"))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- joint-triangulations/s + HBM roofline of the fused hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames 10000] [--pool 32]

A "step" is one pass of the hot path (snowtri_triangulate_condense: pixel->ray, pairwise
triangulation + scoring, association/fusion) over one batch of synthetic input per GPU:
BASELINE.json configs[1] -- 4 cameras x 1 person x 133 joints x 10 000 frames -- already resident
in HBM.  Successive steps walk a POOL of distinct resident batches (default 32 x 64 MB in,
far larger than the 256 MB Infinity Cache) so every launch streams its input from HBM instead
of re-reading a cache-resident 85 MB working set (SURVEY.md §7 hard part 6).

Timed region: barrier + synchronize, K steps, synchronize + barrier; MAX over ranks.
N > 1: one rank per GPU (torch.distributed, backend nccl = RCCL); frames are sharded across ranks
(weak scaling: every rank processes --frames per step) and the path itself needs NO collective (frames are
independent), so `value` is measured without one.  north_star also names a RCCL all-gather that reassembles
the 3D track on every GPU: the same K steps are then timed a second time with that all-gather per step
(double-buffered on a side stream, overlapping the next kernels) and reported as `with_track_allgather`.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel (k_fused_single): algorithmic bytes per launch / mean launch duration
                measured with HIP events on the launch stream; peak = 8 TB/s HBM3E
  cpu_baseline  the oracle (oracle/snowtri_oracle.c, OpenMP over frames) on the host cores, on a
                bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
J = 133


def algorithmic_bytes_per_frame(C, P, Pout, in_bytes=4, out_bytes=4):
    """3*in*C*P*J in + 4*out*Pout*J out (fp32/fp32: 12 C P J + 16 Pout J)."""
    return 3 * in_bytes * C * P * J + 4 * out_bytes * Pout * J


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=10000, help="frames per step per GPU (configs[1] = 10000)")
    ap.add_argument("--pool", type=int, default=32, help="distinct HBM-resident input batches cycled through")
    ap.add_argument("--large-frames", type=int, default=2000000,
                    help="extra single-launch roofline run (0 = skip); SURVEY §8d asks for >= 2e6 frames")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are issued on round-robin (one context each); 2 lets the ramp-up / "
                         "tail of consecutive 10 000-frame launches overlap")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the RCCL all-gather of the track")
    ap.add_argument("--method", choices=["pairwise", "dlt"], default="pairwise",
                    help="pairwise = the reference's algorithm (the metric); dlt = N-view DLT (row N3), for comparison only")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing aid: run the torch.distributed / all-gather code path even with --gpus 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    import torch
    from snowmocap_amd import synth, _lib
    from snowmocap_amd.batch import BatchTriangulator

    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("NCCL_DEBUG", "NONE")      # no RCCL version banner on stdout
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=int(os.environ.get("WORLD_SIZE", world)),
                                device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    F, K_steps, W_steps = args.frames, args.steps, args.warmup
    wl = synth.config_workload(2, F, seed=1000 + rank)           # cfg2 shape, per-rank shard
    Kc, Rc, tc = wl["rig"]
    C, P, Pout = Kc.shape[0], 1, 1
    params = wl["params"]
    method = _lib.DLT if args.method == "dlt" else _lib.PAIRWISE
    bt = BatchTriangulator(Kc, Rc, tc, params, pout_max=Pout, out_dtype=np.float32, device=local_rank, method=method)
    nstreams = max(1, args.streams)
    bts = [bt] + [BatchTriangulator(Kc, Rc, tc, params, pout_max=Pout, out_dtype=np.float32, device=local_rank,
                                    method=method) for _ in range(nstreams - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]

    # pool of distinct resident batches: exact projections + N(0, 1 px) noise, scores U(3.5, 8)
    base = torch.from_numpy(wl["kpts"]).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    pool = []
    for b in range(max(1, args.pool)):
        if b == 0:
            pool.append(base)
        else:
            jitter = torch.zeros_like(base)
            jitter[..., :2] = torch.randn(base.shape[:-1] + (2,), generator=gen, device=dev) * 0.25
            pool.append((base + jitter).contiguous())
    outs = [bt.alloc_outputs(F, dev) for _ in range(len(pool))]
    can_gather = dist is not None and not args.no_gather
    if can_gather:
        gbuf = [torch.empty((world * F, Pout, J, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        side = torch.cuda.Stream(device=dev)

    def timed_region(gather_on):
        """W warm-up + K timed steps; returns seconds (MAX over ranks).  gather_on adds, per step, the RCCL
        all-gather of this step's track shard, issued on a side stream so it overlaps the next kernels."""
        gathered = [None] * len(pool)      # event: the all-gather that last read outs[b] has finished

        def step(i):
            b = i % len(pool)
            k = i % nstreams
            if gather_on and gathered[b] is not None:
                streams[k].wait_event(gathered[b])      # do not overwrite a shard that is still being gathered
            bts[k].run_torch(pool[b], None, out=outs[b], stream=streams[k].cuda_stream)
            if gather_on:
                ready = torch.cuda.Event()
                ready.record(streams[k])
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    dist.all_gather_into_tensor(gbuf[i & 1], outs[b]["xyzs"])
                    done = torch.cuda.Event()
                    done.record(side)
                gathered[b] = done

        def fence():
            torch.cuda.synchronize(dev)        # every stream of this device, side stream included
            if dist is not None:
                dist.barrier()

        for i in range(W_steps):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(K_steps):
            step(W_steps + i)
        fence()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # `value`: frames sharded across ranks, no data-path collective (frames are independent: SURVEY 8e).
    elapsed = timed_region(False)
    joints_per_step = F * Pout * J * world
    value = joints_per_step * K_steps / elapsed
    ms_per_step = elapsed / K_steps * 1e3
    # N > 1: the same K steps again WITH the all-gather of the 3D track that north_star names; reported
    # beside `value` (it is xGMI-bandwidth-bound: 16 B/joint over the links vs 64 B/joint over HBM).
    with_gather = None
    if can_gather:
        e2 = timed_region(True)
        with_gather = {"value": joints_per_step * K_steps / e2, "ms_per_step": e2 / K_steps * 1e3,
                       "what": "same steps + one RCCL all_gather_into_tensor of the step's track shard "
                               f"({F * Pout * J * 16 / 1e6:.1f} MB per rank) per step, overlapped on a side stream"}

    # dominant-kernel duration: HIP events bracketing each launch on the launch stream
    # (snowtri_set_timing records them inside the C ABI around k_fused_single only).
    # (launches are queued back to back on ONE stream and their event pairs read afterwards: a synchronize
    # between launches would let the GPU idle and clock down, and stretch every launch by ~10 %)
    bt.ctx.set_timing(True)
    for i in range(min(K_steps, 1000)):
        bt.run_torch(pool[i % len(pool)], None, out=outs[i % len(pool)])
    kms = bt.ctx.timing_collect()
    bt.ctx.set_timing(False)
    torch.cuda.synchronize(dev)
    kernel_ms = float(np.mean(kms))
    kernel_ms_min = float(np.min(kms))
    bpf = algorithmic_bytes_per_frame(C, P, Pout)
    ach = bpf * F / (kernel_ms * 1e-3) / 1e9

    # correctness guard inside the bench: every frame resolved (count == 1) on the fast path
    cnt = outs[0]["count"].cpu().numpy()
    flg = outs[0]["flags"].cpu().numpy()
    if os.environ.get("SNOWTRI_BENCH_NOCHECK") != "1":       # (timing-only development builds write wrong outputs)
        assert (cnt == 1).all() and ((flg & _lib.FLAG_FASTPATH) != 0).all(), "bench output is not the expected fast path"

    large = None
    if args.large_frames and rank == 0 and world == 1:
        FL = args.large_frames
        reps = (FL + F - 1) // F
        big = torch.cat([pool[i % len(pool)] for i in range(reps)], dim=0)[:FL].contiguous()
        bout = bt.alloc_outputs(FL, dev)
        bt.ctx.set_timing(True)
        lms = []
        for _ in range(5):
            bt.run_torch(big, None, out=bout)
            lms.append(bt.ctx.last_kernel_ms()[0])
        bt.ctx.set_timing(False)
        lm = float(np.median(lms[1:]))
        large = {"frames": FL, "kernel_ms": lm, "joints_per_s": FL * J / (lm * 1e-3),
                 "achieved_GBs": bpf * FL / (lm * 1e-3) / 1e9, "frac": bpf * FL / (lm * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del big, bout

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        prm = orc.make_params(**params)
        ncores = usable_cores()
        sample = wl["kpts"][: min(F, 10000)]
        nps = wl["n_persons"][: sample.shape[0]]
        orc.triangulate_condense_batch(Kc, Rc, tc, sample[:256], nps[:256], prm, Pout, nthreads=1)   # warm
        # Containers often expose more logical CPUs than they may use: try a few team sizes on the
        # bounded sample and report the best one together with the thread count that achieved it.
        trials = sorted({1, min(8, ncores), min(32, ncores), ncores})
        budget = args.cpu_seconds / len(trials)
        best = None
        for nt in trials:
            reps, t_cpu = 0, 0.0
            c0 = time.perf_counter()
            while t_cpu < budget:
                r = orc.triangulate_condense_batch(Kc, Rc, tc, sample, nps, prm, Pout, nthreads=nt)
                reps += 1
                t_cpu = time.perf_counter() - c0
            rate = sample.shape[0] * reps * J / t_cpu
            if best is None or rate > best[0]:
                best = (rate, int(r["threads"]), reps, t_cpu)
        cpu = {"value": best[0], "unit": "joints/s", "cores": best[1], "kind": "port",
               "sample": f"cfg2 batch of {sample.shape[0]} frames x {best[2]} repeats ({best[3]:.1f} s) at the best of "
                         f"{trials} OpenMP threads ({ncores} logical CPUs visible); oracle/snowtri_oracle.c fp64"}

    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path) and args.method != "dlt":     # the PMC pass was made on the pairwise kernel
        try:
            traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    line = None
    if rank == 0:
        line = {
            "metric": "joint-triangulations/sec", "value": value, "unit": "joints/s", "n_gpus": world,
            "steps": K_steps, "warmup": W_steps, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 4 cameras x 1 person x 133 joints x "
                                   f"{F} frames per step per GPU, floor rig, default thresholds; "
                                   f"steps cycle a pool of {len(pool)} distinct HBM-resident batches",
                       "frames_per_step_per_gpu": F, "cameras": C, "persons": P, "joints": J,
                       "method": "pairwise (reference-exact)" if args.method == "pairwise" else "dlt (N-view, NOT the reference's algorithm)",
                       "io": "fp32 in / fp32 out, fp64 math",
                       "streams": nstreams,
                       "parallelism": f"frames sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_fused_single<4,%d,float,float>" % (1 if args.method == "dlt" else 0),
                         "kernel_ms_mean": kernel_ms, "kernel_ms_min": kernel_ms_min,
                         "algorithmic_bytes_per_launch": bpf * F, "bytes_per_joint": bpf / (Pout * J)},
            # the roof that actually binds this kernel: fp64 VALU issue.  422 VALU wave-instructions
            # per 64 joints (rocprofv3 SQ_INSTS_VALU, profiles/), 4 issue cycles each, 1024 SIMDs at the 2.4 GHz peak clock.
            "fp64_valu_issue": None if args.method != "pairwise" else {
                "valu_insts_per_64_joints": 422, "simds": 1024, "peak_clock_GHz": 2.4,
                "frac_this_launch": (F * Pout * J / 64.0 * 422 * 4) / (kernel_ms * 1e-3) / (1024 * 2.4e9),
                "frac_large_batch": None if large is None else (large["joints_per_s"] / 64.0 * 422 * 4) / (1024 * 2.4e9)},
            "cpu_baseline": cpu,
            "with_track_allgather": with_gather,
            "large_batch": large,
            "ray_pair_solves_per_s": value * (C * (C - 1) // 2),
        }
    for b_ in bts:
        b_.close()
    if dist is not None:
        dist.destroy_process_group()
    if line is not None:
        # the JSON line must be the LAST thing on stdout: RCCL prints a version banner through C stdio,
        # which would otherwise be flushed after Python's output at exit
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        if world > 1:
            time.sleep(1.0)                                # let the other ranks' exit-time output drain first
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

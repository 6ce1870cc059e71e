#!/usr/bin/env python3
"""bench.py -- joint-triangulations/s + HBM roofline of the fused hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames 10000] [--pool 32] [--repeats 7]

A "step" is one pass of the hot path (snowtri_triangulate_condense: pixel->ray, pairwise
triangulation + scoring, association/fusion) over one batch of synthetic input per GPU:
BASELINE.json configs[1] -- 4 cameras x 1 person x 133 joints x 10 000 frames -- already resident
in HBM.  Successive steps walk a POOL of distinct resident batches (default 32 x 64 MB in,
far larger than the 256 MB Infinity Cache) so every launch streams its input from HBM instead
of re-reading a cache-resident 85 MB working set (SURVEY.md §7 hard part 6).

Timed region: barrier + synchronize, K steps, synchronize + barrier; MAX over ranks.  The region is
repeated --repeats times in one run (the chip's clock follows its power budget and a 20-step region
lasts half a millisecond: single regions scatter by +-20 %); `value` / `ms_per_step` are the MEDIAN
region, `repeats` lists min / max / all.

N > 1: one rank per GPU (torch.distributed, backend nccl = RCCL).  Started WITHOUT a launcher
(`python bench.py --gpus 8`) the script re-executes itself under `python -m torch.distributed.run`
on 127.0.0.1; started by a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment.  Frames are sharded across ranks (weak scaling: every rank processes --frames per step)
and the path itself needs NO collective (frames are independent), so `value` is measured without one.
north_star also names a RCCL all-gather that reassembles the 3D track on every GPU: the same K steps
are then timed a second time with that all-gather per step (double-buffered on a side stream,
overlapping the next kernels) and reported as `with_track_allgather`.
`--dry-run` exercises the launch / rendezvous logic alone (gloo, no GPU work): used by the CPU tests.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline          dominant kernel: algorithmic bytes per launch / mean launch duration measured with HIP
                    events on the launch stream (one stream, launches back to back); peak = 8 TB/s HBM3E
  roofline_region   the same bytes / `ms_per_step` of the timed region (--streams streams, launches overlap)
  large_batch       one 2 000 000-frame launch (SURVEY 8d's roofline run)
  extra_workloads   BASELINE configs[2] (8 cameras x 4 persons, 10 000 frames) and the per-GPU share of
                    configs[4] (16 x 8, 12 500 frames): multi-person kernel, fp64-VALU roofline each
  cpu_baseline      the oracle (oracle/snowtri_oracle.c, OpenMP over frames) on the host cores, on a
                    bounded sample of the same workload (rank 0, N = 1 only), and its distance to the GPU result
  per_frame_api     the path main.py really calls, one frame at a time (reference main.py:50-71,106): median wall time
                    of add_human_2D_points x 4 -> Human_Triangulation -> Human_Triangulation_Condense ->
                    clear_2D_points on the floor rig (configs[0] shape, 300 frames), and of ONE F = 1 host call of
                    the fused entry, beside the reference's 24.9 ms per frame
  with_track_allgather  N > 1 (or --force-dist): the same K steps through the PRODUCT's sharded entry,
                    ShardedTriangulator.run (snowmocap_amd/sharded.py): the shard computed in --chunks pieces, every
                    output of a piece packed in one buffer and all-gathered with one RCCL collective on a side stream
                    under the next piece's kernel; every rank's kernel time is listed (a slow rank shows)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector fp64 (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)
FLOP_PER_SOLVE = 90.0          # one ray-pair solve + score + accumulate
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


def kernel_source_hash():
    """sha256 over the HIP sources of libsnowtri.so: ties a PMC traffic measurement to the kernels it was made on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "snowmocap_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hpp", ".hip")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: become N ranks on this node (rendezvous on 127.0.0.1)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("NCCL_DEBUG", "NONE")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """Launch / rendezvous logic only: gloo, no GPU.  Rank 0 prints the ranks it saw."""
    import torch.distributed as dist
    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ.get("WORLD_SIZE", world)))
    assert dist.get_world_size() == world, (dist.get_world_size(), world)
    seen = [None] * world
    dist.all_gather_object(seen, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid()})
    dist.barrier()
    # the gather leg of the real run (gather_leg -> snowmocap_amd.sharded.gather_track_chunked), with a stand-in for the
    # kernels: every rank fills its frame block with values that name (rank, local frame); checked on the gathered track
    import torch
    F = max(1, args.frames if args.frames < 1000 else 37)
    gather = {}
    from snowmocap_amd.sharded import auto_chunks
    for chunks in args.chunks:
        def fake_compute(lo, hi, views, _r=rank):
            views["xyzs"][: hi - lo] = (_r * 1000.0 + torch.arange(lo, hi, dtype=torch.float32)).view(-1, 1, 1, 1)
            views["pscore"][: hi - lo] = float(_r)
            views["count"][: hi - lo] = 1
            views["flags"][: hi - lo] = 4
        regions = {"xyzs": ((1, J, 4), torch.float32), "pscore": ((1,), torch.float32), "count": ((), torch.int32),
                   "flags": ((), torch.int32)}
        dt, full = gather_leg(dist, torch, None, lambda b, ch: sharded_run_generic(fake_compute, F, world * F, regions, ch, None),
                              steps=2, warmup=1, chunks=auto_chunks(F) if chunks == "auto" else chunks)
        want = torch.cat([r * 1000.0 + torch.arange(F, dtype=torch.float32) for r in range(world)])
        ok = bool(torch.equal(full["xyzs"][:, 0, 0, 0], want)) and bool((full["count"] == 1).all())
        gather[str(chunks)] = {"ok": ok, "frames_gathered": int(full["xyzs"].shape[0]), "s_per_step": dt / 2}
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": dist.get_world_size(), "backend": "gloo",
                          "ranks": seen, "gather_leg": gather}), flush=True)
    dist.destroy_process_group()


def sharded_run_generic(compute_block, n_local, F_total, regions, chunks, device):
    from snowmocap_amd.sharded import gather_track_chunked
    return gather_track_chunked(compute_block, n_local, F_total, regions, chunks=chunks, device=device)


def gather_leg(dist, torch, dev, run_step, steps, warmup, chunks):
    """`warmup` + `steps` calls of run_step(i, chunks) -- the product's sharded entry: compute the rank's shard in
    pieces, all-gather every piece -- between fences (synchronize + barrier); the clock is read between a rank's own
    device synchronisation and the barrier (as in timed_region: the barrier's latency is not the path's); returns
    (seconds MAX over ranks, the last gathered track)."""
    def sync():
        if dev is not None:
            torch.cuda.synchronize(dev)
    full = None
    for i in range(warmup):
        full = run_step(i, chunks)
    sync()
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        full = run_step(warmup + i, chunks)
    sync()
    dt = time.perf_counter() - t0
    dist.barrier()
    tt = torch.tensor([dt], dtype=torch.float64, device=dev if dev is not None else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item()), full


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=10000, help="frames per step per GPU (configs[1] = 10000)")
    ap.add_argument("--pool", type=int, default=32, help="distinct HBM-resident input batches cycled through")
    ap.add_argument("--repeats", type=int, default=7, help="how many times the K-step region is timed (median reported)")
    ap.add_argument("--large-frames", type=int, default=2000000,
                    help="extra single-launch roofline run (0 = skip); SURVEY §8d asks for >= 2e6 frames")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps are issued on round-robin (one context each); 2 lets the ramp-up / "
                         "tail of consecutive 10 000-frame launches overlap")
    ap.add_argument("--device-warmup-ms", type=float, default=100.0,
                    help="untimed steps issued for this long before the first timed region (the chip's clock ramp)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the RCCL all-gather of the track")
    ap.add_argument("--chunks", type=lambda v: [x if x == "auto" else max(1, int(x)) for x in v.split(",")], default=["auto"],
                    help="pieces the shard is cut into for the overlapped all-gather (ShardedTriangulator.run): `auto` (pieces of "
                         ">= 32 768 frames, at most 8: one piece for a 10 000-frame shard) or a number; a comma-separated list "
                         "times each (the first is reported as with_track_allgather, all of them in its `sweep`)")
    ap.add_argument("--no-per-frame", action="store_true", help="skip the per-frame API latency (main.py's own call sequence)")
    ap.add_argument("--method", choices=["pairwise", "dlt"], default="pairwise",
                    help="pairwise = the reference's algorithm (the metric); dlt = N-view DLT (row N3), for comparison only")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing aid: run the torch.distributed / all-gather code path even with --gpus 1")
    ap.add_argument("--one-device", action="store_true",
                    help="testing aid for a one-GPU box: every rank on cuda:0, torch.distributed over gloo (RCCL refuses two ranks "
                         "on one device) -- the N > 1 code path with the real kernels; the line says `backend: gloo`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the multi-person extra workloads")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--dry-run", action="store_true", help="rendezvous only (gloo, no GPU): CPU test of the launcher")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    if args.dry_run:
        return dry_run(args)

    import torch
    from snowmocap_amd import synth, _lib
    from snowmocap_amd.batch import BatchTriangulator

    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = "gloo" if args.one_device else "nccl"
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("NCCL_DEBUG", "NONE")      # no RCCL version banner on stdout
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=int(os.environ.get("WORLD_SIZE", world)),
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=int(os.environ.get("WORLD_SIZE", world)))
        assert dist.get_world_size() == world
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    me = {"rank": rank, "local_rank": local_rank, "device": int(torch.cuda.current_device()),
          "name": torch.cuda.get_device_name(dev)}
    ranks_seen = [me]
    if dist is not None:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)

    F, K_steps, W_steps = args.frames, args.steps, args.warmup
    wl = synth.config_workload(2, F, seed=1000 + rank)           # cfg2 shape, per-rank shard
    Kc, Rc, tc = wl["rig"]
    C, P, Pout = Kc.shape[0], 1, 1
    params = wl["params"]
    method = _lib.DLT if args.method == "dlt" else _lib.PAIRWISE
    # `bt`: one stream, for the per-kernel figures (launches back to back).  `bto`: the SAME library in its overlap mode
    # (snowtri_ctx_set_overlap, BatchTriangulator(streams=n)): a plain loop of calls on one caller stream, issued by the
    # library round-robin on n internal streams so that the ramp-up / tail of consecutive launches overlap -- what round 3's
    # bench did by hand with twin contexts and its own streams is now what any caller gets.
    bt = BatchTriangulator(Kc, Rc, tc, params, pout_max=Pout, out_dtype=np.float32, device=local_rank, method=method)
    nstreams = max(1, min(4, args.streams))
    bto = bt if nstreams == 1 else BatchTriangulator(Kc, Rc, tc, params, pout_max=Pout, out_dtype=np.float32, device=local_rank,
                                                     method=method, streams=nstreams)
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

    def timed_region():
        """W warm-up + K timed steps of the path itself (no collective: frames are independent); returns (seconds MAX over
        ranks, this rank's seconds).  Both fences are barrier + synchronize, but the CLOCK is read between the device
        synchronisation and the barrier: every rank times its own K steps (start: after the barrier that lines the ranks up
        and a synchronize that drains it; stop: right after its own synchronize) and the MAX over ranks is taken afterwards,
        so the latency of an N-rank RCCL barrier (tens of microseconds against a 0.4 ms region) is not booked to the path."""
        def step(i):
            b = i % len(pool)
            bto.run_torch(pool[b], None, out=outs[b])

        for i in range(W_steps):
            step(i)
        bto.join()
        torch.cuda.synchronize(dev)            # every stream of this device, the library's internal ones included
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)        # (the barrier is itself device work on RCCL's stream)
        t0 = time.perf_counter()
        for i in range(K_steps):
            step(W_steps + i)
        bto.join()                             # the caller's stream waits for every overlapped call ...
        torch.cuda.synchronize(dev)            # ... and the host for the device
        dt_own = time.perf_counter() - t0
        dt = dt_own
        if dist is not None:
            dist.barrier()
            tt = torch.tensor([dt_own], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, dt_own

    # Device warm-up (untimed, the same steps): the chip leaves its idle power state over tens of milliseconds -- without it
    # the regions of one run get faster one after the other (0.0244 ... 0.0207 ms per step over the seven regions of a
    # default run, and the driver's `--steps 20` regions all fall into the ramp).  The timed regions below are unchanged:
    # W warm-up steps, then exactly K steps between two fences.
    t_w = time.perf_counter()
    n_w = 0
    while (time.perf_counter() - t_w) * 1e3 < args.device_warmup_ms:
        for i in range(50):
            bto.run_torch(pool[i % len(pool)], None, out=outs[i % len(pool)])
        bto.join()
        torch.cuda.synchronize(dev)
        n_w += 50
    # `value`: frames sharded across ranks, no data-path collective (frames are independent: SURVEY 8e).
    regions_both = [timed_region() for _ in range(max(1, args.repeats))]
    regions = [r[0] for r in regions_both]
    elapsed = float(np.median(regions))
    # this rank's own K-step time of the median region, gathered below into `ms_per_step_per_rank`
    own_ms_per_step = sorted(regions_both)[len(regions_both) // 2][1] / K_steps * 1e3
    joints_per_step = F * Pout * J * world
    value = joints_per_step * K_steps / elapsed
    ms_per_step = elapsed / K_steps * 1e3
    # N > 1: the same K steps again WITH the all-gather of the 3D track that north_star names; reported
    # beside `value` (it is xGMI-bandwidth-bound: 16 B/joint over the links vs 64 B/joint over HBM).
    with_gather = None
    if can_gather:
        # the PRODUCT's sharded entry (snowmocap_amd/sharded.py): per step, ShardedTriangulator.run computes this rank's
        # shard in `chunks` pieces into one packed buffer per piece (joints, person scores, counts, flags) and all-gathers
        # each piece with ONE collective on a side stream under the next piece's kernel; buffers are reused step to step
        from snowmocap_amd.sharded import ShardedTriangulator
        sht = ShardedTriangulator(Kc, Rc, tc, params, pout_max=Pout, device=local_rank, chunks=args.chunks[0], reuse_buffers=True)
        sweep = []
        for chunks in args.chunks:
            e2s = []
            for _ in range(max(1, min(3, args.repeats))):
                e2, full = gather_leg(dist, torch, dev, lambda i, ch: sht.run(pool[i % len(pool)], world * F, chunks=ch),
                                      steps=K_steps, warmup=min(W_steps, 5), chunks=chunks)
                e2s.append(e2)
            e2 = float(np.median(e2s))
            ok = tuple(full["xyzs"].shape) == (world * F, Pout, J, 4) and bool((full["count"] == 1).all())
            sweep.append({"chunks": chunks, "pieces": sht.last_chunks, "value": joints_per_step * K_steps / e2, "ms_per_step": e2 / K_steps * 1e3,
                          "gathered_track_ok": ok, "gathered_bytes_per_rank_per_step": int(sht.last_gather_bytes)})
        sht.bt.close()
        with_gather = dict(sweep[0], sweep=sweep,
                           what="same steps through ShardedTriangulator.run: the shard in `chunks` pieces, each piece's outputs "
                                f"(joints, person scores, counts, flags: {(F * Pout * (J * 16 + 4) + 8 * F) / 1e6:.1f} MB per rank per step) "
                                "packed and all-gathered with one RCCL collective on a side stream under the next piece's kernel")

    # dominant-kernel duration: HIP events bracketing each launch on the launch stream
    # (snowtri_set_timing records them inside the C ABI around the fused kernel only).
    # (launches are queued back to back on ONE stream and their event pairs read afterwards: a synchronize
    # between launches would let the GPU idle and clock down, and stretch every launch by ~10 %)
    # Two passes of the same launches: (1) the event pair ATTACHED to the dispatch (hipExtLaunchKernelGGL start / stop events:
    # the kernel's own begin and end, the duration rocprofv3's kernel trace reports -- the figure `roofline` quotes, and the
    # one profiles/'s average must agree with); (2) the pair BRACKETING the launch (a record before and after: adds the command
    # processor's hand-over on both sides, ~1 us), kept in the line as `kernel_ms_mean_bracketed`.
    n_timed = min(max(K_steps, 200), 1000)
    import snowmocap_amd.batch as _batch
    trace_index = {}     # [first, last) ordinal of each loop's launches among this process's fused calls (scripts/gpu_event_check.sh)
    def timed_launches(attach):
        bt.ctx.set_timing(True, attach=attach)
        first = _batch.FUSED_CALLS
        for i in range(n_timed):
            bt.run_torch(pool[i % len(pool)], None, out=outs[i % len(pool)])
        trace_index["attached" if attach else "bracketed"] = [first, _batch.FUSED_CALLS]
        k = bt.ctx.timing_collect()
        bt.ctx.set_timing(False)
        torch.cuda.synchronize(dev)
        return k
    kms_bracketed = timed_launches(False)
    kms = timed_launches(True)
    # (1b) the same attached pairs while the launches are issued as the timed region issues them: a plain loop of calls in
    # the library's overlap mode.  Two consecutive launches then share the chip, each takes longer from its begin to its end,
    # and a step -- the time per launch of the loop -- is SHORTER than one launch alone: `ms_per_step` < `kernel_ms_mean` is
    # that overlap, and this figure shows it inside the line (round-4 review).
    kms_overlapped = None
    if bto is not bt:
        bto.ctx.set_timing(True, attach=True)
        for i in range(n_timed):
            bto.run_torch(pool[i % len(pool)], None, out=outs[i % len(pool)])
        bto.join()
        kms_overlapped = bto.ctx.timing_collect()
        bto.ctx.set_timing(False)
        torch.cuda.synchronize(dev)
    trace_index["step"] = [_batch.FUSED_CALLS, _batch.FUSED_CALLS + n_timed]
    # (3) the same launches with NO event in between, one pair around the whole loop: the time from one launch's end to the
    # next one's end on one stream (kernel + the command processor's hand-over to the next dispatch), `kernel_ms_step_one_stream`
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n_timed):
        bt.run_torch(pool[i % len(pool)], None, out=outs[i % len(pool)])
    e1.record()
    torch.cuda.synchronize(dev)
    kernel_ms_step = e0.elapsed_time(e1) / n_timed
    kernel_ms = float(np.mean(kms))
    kernel_ms_min = float(np.min(kms))
    kernel_ms_bracketed = float(np.mean(kms_bracketed))
    # the achieved rate also against a device-copy bandwidth MEASURED on this box (1 GiB read + 1 GiB written per
    # copy, best of 5 after a warm-up; bytes moved = 2 x the tensor)
    copy_GBs = None
    if not args.dry_run:
        csrc = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
        cdst = torch.empty_like(csrc)
        cdst.copy_(csrc)
        best = float("inf")
        for _ in range(5):
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            cdst.copy_(csrc)
            c1.record()
            torch.cuda.synchronize(dev)
            best = min(best, c0.elapsed_time(c1))
        copy_GBs = 2.0 * csrc.numel() * 4 / (best * 1e-3) / 1e9
        del csrc, cdst
    bpf = algorithmic_bytes_per_frame(C, P, Pout)
    ach = bpf * F / (kernel_ms * 1e-3) / 1e9
    ach_region = bpf * F / (ms_per_step * 1e-3) / 1e9          # per GPU: every rank streams its own shard

    # correctness guard inside the bench: every frame of every pool batch resolved (count == 1) on the fast path
    if os.environ.get("SNOWTRI_BENCH_NOCHECK") != "1":       # (timing-only development builds write wrong outputs)
        for o in outs:
            cnt = o["count"].cpu().numpy()
            flg = o["flags"].cpu().numpy()
            assert (cnt == 1).all() and ((flg & _lib.FLAG_FASTPATH) != 0).all(), "bench output is not the expected fast path"

    kernel_name = bt.ctx.last_kernel_names()      # what the fused call really launched (snowtri_last_kernel_names)
    # every rank's kernel time (a slow rank would otherwise hide behind rank 0's)
    kernel_ms_ranks = [kernel_ms]
    step_ms_ranks = [own_ms_per_step]
    if dist is not None:
        kdev = dev if backend == "nccl" else "cpu"        # (gloo gathers host tensors)
        kt = torch.zeros(2 * world, dtype=torch.float64, device=kdev)
        dist.all_gather_into_tensor(kt, torch.tensor([kernel_ms, own_ms_per_step], dtype=torch.float64, device=kdev))
        kt = kt.cpu().view(world, 2)
        kernel_ms_ranks = [float(x) for x in kt[:, 0]]
        step_ms_ranks = [float(x) for x in kt[:, 1]]

    large = None
    if args.large_frames and rank == 0 and world == 1:
        FL = args.large_frames
        reps = (FL + F - 1) // F
        big = torch.cat([pool[i % len(pool)] for i in range(reps)], dim=0)[:FL].contiguous()
        bout = bt.alloc_outputs(FL, dev)
        bt.ctx.set_timing(True)
        lms = []
        for _ in range(6):
            bt.run_torch(big, None, out=bout)
            lms.append(bt.ctx.last_kernel_ms()[0])
        bt.ctx.set_timing(False)
        lm = float(np.median(lms[1:]))
        large = {"frames": FL, "kernel": bt.ctx.last_kernel_names(), "kernel_ms": lm, "kernel_ms_all": lms[1:], "joints_per_s": FL * J / (lm * 1e-3),
                 "achieved_GBs": bpf * FL / (lm * 1e-3) / 1e9, "frac": bpf * FL / (lm * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del big, bout

    extra = None
    if rank == 0 and world == 1 and not args.no_extra and args.method == "pairwise":
        extra = extra_workloads(torch, dev, local_rank)

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
        # the oracle as the checker of this run's GPU output: pool batch 0 is the sample's own batch
        bt.run_torch(pool[0], None, out=outs[0])
        torch.cuda.synchronize(dev)
        got = outs[0]["xyzs"][: sample.shape[0]].cpu().numpy().astype(np.float64)
        err_m = float(np.abs(got[..., :3] - r["xyz"]).max())
        err_s = float((np.abs(got[..., 3] - r["kscore"]) / np.maximum(np.abs(r["kscore"]), 1e-30)).max())
        assert err_m < 1e-4, f"GPU joints differ from the oracle by {err_m} m"
        cpu = {"value": best[0], "unit": "joints/s", "cores": best[1], "kind": "port",
               "sample": f"cfg2 batch of {sample.shape[0]} frames x {best[2]} repeats ({best[3]:.1f} s) at the best of "
                         f"{trials} OpenMP threads ({ncores} logical CPUs visible); oracle/snowtri_oracle.c fp64",
               "gpu_vs_oracle_max_abs_m": err_m, "gpu_vs_oracle_max_rel_score": err_s}

    # HBM bytes per launch from the PMC counters (rocprofv3 FETCH_SIZE / WRITE_SIZE passes, scripts/profile.sh):
    # only quoted while the kernels are the ones it was measured on.
    traffic, traffic_note = None, "no PMC measurement on file"
    valu_per_64, valu_note = None, "no SQ_INSTS_VALU measurement of these kernel sources on file"
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path) and args.method != "dlt":
        try:
            pj = json.load(open(pmc_path))
            if pj.get("source_sha256") == kernel_source_hash() and pj.get("kernel") == kernel_name:
                if pj.get("frames_per_launch") == F:
                    traffic = pj.get("hbm_bytes_per_launch")
                    traffic_note = "rocprofv3 PMC passes of this kernel source (profiles/pmc_traffic.json: source_sha256 matches)"
                else:
                    traffic_note = "profiles/pmc_traffic.json was measured on another launch size: not quoted"
                if pj.get("valu", {}).get("per_64_joints"):
                    valu_per_64 = float(pj["valu"]["per_64_joints"])
                    valu_note = ("rocprofv3 SQ_INSTS_VALU of this kernel source on a %d-frame launch / wave-items "
                                 "(profiles/pmc_traffic.json: source_sha256 matches)" % pj["valu"].get("frames", 0))
            else:
                traffic_note = "profiles/pmc_traffic.json was measured on other kernel sources: not quoted"
        except Exception:
            pass

    per_frame = None
    if rank == 0 and world == 1 and not args.no_per_frame and args.method == "pairwise":
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from bench_per_frame import per_frame_api
        per_frame = per_frame_api(frames=300, warm=20)

    overrides = bt.ctx.overrides()
    line = None
    if rank == 0:
        line = {
            "metric": "joint-triangulations/sec", "value": value, "unit": "joints/s", "n_gpus": world,
            # what changed in the MEANING of a key, round by round (a figure is comparable only with records of the same schema
            # for that key): 3 = roofline.kernel_ms_mean / .frac from events attached to the dispatch (rounds 1-2: bracketing
            # records, still reported as kernel_ms_mean_bracketed / frac_bracketed); 4 = `value` issued as a plain loop of calls in
            # the library's overlap mode (round 3: two contexts alternated by the bench), extra_workloads[*].kernel_ms from calls
            # queued back to back (round 3: one synchronised call), N > 1 clock read before the barrier
            "schema": 4,
            "steps": K_steps, "warmup": W_steps, "ms_per_step": ms_per_step, "ms_per_step_per_rank": step_ms_ranks, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "rccl_ranks": dist.get_world_size() if dist is not None else 1,
            "backend": (backend if backend == "gloo" else "nccl (RCCL)") if dist is not None else None,
            "ranks": ranks_seen,
            "device_warmup": {"ms": args.device_warmup_ms, "untimed_steps": n_w},
            "repeats": {"n": len(regions), "statistic": "median", "ms_per_step_min": min(regions) / K_steps * 1e3,
                        "ms_per_step_max": max(regions) / K_steps * 1e3,
                        "ms_per_step_all": [r / K_steps * 1e3 for r in regions]},
            "config": {"workload": "BASELINE configs[1]: 4 cameras x 1 person x 133 joints x "
                                   f"{F} frames per step per GPU, floor rig, default thresholds; "
                                   f"steps cycle a pool of {len(pool)} distinct HBM-resident batches",
                       "frames_per_step_per_gpu": F, "cameras": C, "persons": P, "joints": J,
                       "method": "pairwise (reference-exact)" if args.method == "pairwise" else "dlt (N-view, NOT the reference's algorithm)",
                       "io": "fp32 in / fp32 out, fp64 math",
                       "streams": nstreams,
                       "how_issued": "one caller stream; the library's overlap mode (snowtri_ctx_set_overlap) rotates the calls over "
                                     f"{nstreams} internal streams; join + synchronize at the end of a region" if nstreams > 1 else "one stream",
                       "context_overrides": overrides,
                       "build": _lib.build_info(),
                       "parallelism": f"frames sharded x{world}, no data-path collective",
                       "extra_workloads": None if extra is None else [e["workload"] for e in extra]},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": kernel_name, "streams": 1,
                         "how": "HIP events on the launch stream, launches back to back on one stream; the event pair of a launch is attached "
                                "to its dispatch (hipExtLaunchKernelGGL start / stop events = the kernel's own begin and end, what a rocprofv3 "
                                "kernel trace reports; no event record between the launches); kernel_ms_mean_bracketed = the same launches "
                                "with a pair recorded before and after each (barrier packets between the launches: the interval adds the "
                                "command processor's hand-over on both sides, and the kernels no longer run back to back); "
                                "kernel_ms_step_one_stream = the same launches with no event in between, one pair around the loop / launches "
                                "(kernel + hand-over to the next dispatch: what a rocprofv3 --stats average of a one-stream run lands on)",
                         "kernel_ms_mean": kernel_ms, "kernel_ms_median": float(np.median(kms)), "kernel_ms_min": kernel_ms_min, "launches": len(kms),
                         "kernel_ms_mean_overlapped": None if kms_overlapped is None or not len(kms_overlapped) else float(np.mean(kms_overlapped)),
                         "overlapped_note": f"the same kernel's begin-to-end time while launches are issued as `value` issues them (overlap mode, "
                                            f"{nstreams} internal streams): two launches share the chip, each lasts longer, and ms_per_step "
                                            "(time per launch of that loop) is shorter than kernel_ms_mean (one launch alone on the chip)",
                         "kernel_ms_mean_bracketed": kernel_ms_bracketed,
                         "frac_bracketed": bpf * F / (kernel_ms_bracketed * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "measured_device_copy_GBs": copy_GBs,
                         "frac_of_measured_copy": None if not copy_GBs else ach / copy_GBs,
                         "trace_index": dict(trace_index, total_fused_calls=_batch.FUSED_CALLS),
                         "kernel_ms_step_one_stream": kernel_ms_step,
                         "frac_step_one_stream": bpf * F / (kernel_ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_mean_per_rank": kernel_ms_ranks,
                         "algorithmic_bytes_per_launch": bpf * F, "bytes_per_joint": bpf / (Pout * J)},
            "roofline_region": {"bound": "hbm", "achieved": ach_region, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": ach_region / HBM_PEAK_GBS, "streams": nstreams,
                                "how": "algorithmic bytes per step / ms_per_step of the timed region (per GPU; "
                                       f"a plain loop of calls in the library's overlap mode over {nstreams} internal streams, so consecutive launches overlap)"},
            # the roof that actually binds this kernel: fp64 VALU issue.  VALU wave-instructions per 64 joints
            # MEASURED (rocprofv3 SQ_INSTS_VALU, quoted only while profiles/pmc_traffic.json carries the hash of these kernel
            # sources), 4 issue cycles each, 1024 SIMDs at the 2.4 GHz peak clock.
            "fp64_valu_issue": None if args.method != "pairwise" else {
                "valu_insts_per_64_joints": valu_per_64, "source": valu_note, "simds": 1024, "peak_clock_GHz": 2.4,
                "frac_this_launch": None if valu_per_64 is None else
                    (F * Pout * J / 64.0 * valu_per_64 * 4) / (kernel_ms * 1e-3) / (1024 * 2.4e9),
                "frac_large_batch": None if large is None or valu_per_64 is None else
                    (large["joints_per_s"] / 64.0 * valu_per_64 * 4) / (1024 * 2.4e9)},
            "cpu_baseline": cpu,
            "per_frame_api": per_frame,
            "with_track_allgather": with_gather,
            "large_batch": large,
            "extra_workloads": extra,
            "ray_pair_solves_per_s": value * (C * (C - 1) // 2),
        }
        # LAST key of the line: the driver keeps `parsed` (the contract keys) and the last 8 KB of stdout, so what survives of
        # a 40 KB line used to be an accident of ordering (round 5: the float32 8 x 4 line fell out).  Every BASELINE config
        # and north_star's DLT, one short entry each, in the tail.
        line["summary"] = summary_of(line)
    for b_ in {id(bt): bt, id(bto): bto}.values():
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


def summary_of(line):
    """Compact digest of the line (every BASELINE configuration that fits one GPU, the DLT method, the per-frame API, the CPU
    baseline): numbers only, < 3 KB."""
    def r3(x):
        return None if x is None else float("%.4g" % x)
    ex = line.get("extra_workloads") or []

    def find(*needles, method="pairwise", zero_fill=True):
        for e in ex:
            if all(n in e["workload"] for n in needles) and e.get("method", "pairwise") == method and e.get("zero_fill", True) == zero_fill:
                return e
        return None

    def multi(e):
        return None if e is None else {"frames_per_s": r3(e["frames_per_s"]), "ms_per_call": r3(e["kernel_ms"]),
                                       "frac_fp64_nominal": r3(e["roofline"]["frac"]), "persons_per_frame": r3(e.get("mean_persons_per_frame")),
                                       "two_calls_in_flight_frames_per_s": r3(e.get("two_streams", {}).get("frames_per_s"))}

    def single(e):
        return None if e is None else {"joints_per_s": r3(e["joints_per_s"]), "us_per_call": r3(e["kernel_ms"] * 1e3), "frac_fp64": r3(e["roofline"]["frac"]),
                                       "frac_hbm": r3(e.get("hbm", {}).get("frac_of_8TBs")), "kernel": e["kernel"][:48]}
    rf, lb, cpu, pf = line["roofline"], line.get("large_batch"), line.get("cpu_baseline"), line.get("per_frame_api")
    nzf = find("NO_ZERO_FILL", zero_fill=False)
    s = {
        "configs1_4x1_10000_frames": {"joints_per_s": r3(line["value"]), "ms_per_step": r3(line["ms_per_step"]), "kernel_us_one_stream": r3(rf["kernel_ms_mean"] * 1e3),
                                      "frac_hbm": r3(rf["frac"]), "frac_of_measured_copy": r3(rf.get("frac_of_measured_copy")), "hbm_traffic_over_algorithmic": None if not rf.get("traffic") else r3(rf["traffic"] / rf["algorithmic_bytes_per_launch"]),
                                      "frac_hbm_2M_frame_launch": None if lb is None else r3(lb["frac"])},
        "configs2_8x4_10000_frames_f32": multi(find("configs[2]: 8 cameras")),
        "configs2_8x4_f64_out": multi(find("configs[2] with float64")),
        "configs2_8x4_no_zero_fill": None if nzf is None else dict(multi(nzf), output_MB_written_per_call=r3(nzf["output_bytes_written_per_frame"] * nzf["frames"] / 1e6),
                                                                   output_MB_written_per_call_default=r3(nzf["pout_max"] * (J * 16 + 4) * nzf["frames"] / 1e6)),
        "configs4_share_16x8_12500_frames": multi(find("configs[4]")),
        "dlt_4x1": single(find("DLT", "4 cameras x 1", method="dlt")),
        "dlt_8x1": single(find("DLT", "8 cameras x 1", method="dlt")),
        "dlt_8x4_with_association": multi(find("DLT", "configs[2]", method="dlt")),
        "single_4x1_f64_out": single(next((e for e in ex if e.get("out_dtype") == "float64" and "4 cameras x 1" in e["workload"] and e.get("method") == "pairwise"), None)),
        "single_6x1": single(next((e for e in ex if e.get("out_dtype") == "float32" and "6 cameras x 1" in e["workload"] and e.get("method") == "pairwise"), None)),
        "single_8x1": single(next((e for e in ex if e.get("out_dtype") == "float32" and "8 cameras x 1" in e["workload"] and e.get("method") == "pairwise"), None)),
        "per_frame_api_us": None if not pf else {k: r3(v) for k, v in pf.items() if isinstance(v, (int, float)) and "us" in k},
        "cpu_baseline": None if not cpu else {"joints_per_s": r3(cpu["value"]), "cores": cpu["cores"], "gpu_vs_oracle_max_abs_m": r3(cpu["gpu_vs_oracle_max_abs_m"])},
        "n_gpus": line["n_gpus"], "dtype": line["dtype"],
    }
    return s


def extra_workloads(torch, dev, device_index):
    """The multi-person configurations of BASELINE.json on this GPU (the streaming association): configs[2] = 8 cameras x
    4 persons x 10 000 frames, and one GPU's share of configs[4] = 16 cameras x 8 persons x 12 500 frames.  A few
    hundred distinct frames are generated on the host and tiled on the device (frames are independent; the kernel
    is fp64-VALU-bound, SURVEY 8d, so cache residency of the tiled input does not help it).  Roofline: fp64 VALU,
    algorithmic flops = frames x candidates x joints x 90 flop (one pair solve + score per candidate joint;
    the second solve of the surviving clusters in the fusion kernels is not credited, and neither is the shorter
    distance-only solve the candidate sums really use -- the figure prices the reference's work, not the kernel's)."""
    from snowmocap_amd import synth
    from snowmocap_amd.batch import BatchTriangulator
    from snowmocap_amd import _lib
    res = []
    for cfg, F, gen_frames, pout, label, odt, meth, zero_fill in (
            (3, 10000, 1000, 16, "BASELINE configs[2]: 8 cameras x 4 persons x 133 joints x 10 000 frames", np.float32, _lib.PAIRWISE, True),
            (5, 12500, 250, 32, "BASELINE configs[4] per-GPU share: 16 cameras x 8 persons x 133 joints x 12 500 frames", np.float32, _lib.PAIRWISE, True),
            # the reference's own output type (triangulation.py:136-148 returns float64 arrays): same route, Newton-refined 1/dist,
            # person scores from the fused joints
            (3, 10000, 1000, 16, "BASELINE configs[2] with float64 outputs: 8 cameras x 4 persons x 133 joints x 10 000 frames", np.float64, _lib.PAIRWISE, True),
            # the same call with SNOWTRI_CALL_NO_ZERO_FILL: the 16 - ~5 unused slots of every frame are not written
            (3, 10000, 1000, 16, "BASELINE configs[2] with SNOWTRI_CALL_NO_ZERO_FILL: 8 cameras x 4 persons x 133 joints x 10 000 frames", np.float32, _lib.PAIRWISE, False),
            # north_star's DLT (row N3; NOT the reference's algorithm) behind the reference's association: candidates + clustering as
            # above, then one N-view DLT per cluster and joint over its distinct observations (k_frame_recompute<1>)
            (3, 10000, 1000, 16, "DLT (method = SNOWTRI_DLT) on BASELINE configs[2]: 8 cameras x 4 persons x 133 joints x 10 000 frames, the reference's association, then DLT per cluster", np.float32, _lib.DLT, True)):
        wl = synth.config_workload(cfg, gen_frames)
        K, R, t = wl["rig"]
        C, P = K.shape[0], wl["kpts"].shape[2]
        rep = F // gen_frames
        kp = torch.from_numpy(wl["kpts"]).to(dev).repeat(rep, 1, 1, 1, 1).contiguous()
        npers = torch.from_numpy(wl["n_persons"]).to(dev).repeat(rep, 1).contiguous()
        bt = BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=odt, device=device_index, method=meth, zero_fill=zero_fill)
        out = bt.run_torch(kp, npers)
        torch.cuda.synchronize(dev)
        # ONE call = one fused entry on one caller stream (inside it the library alternates the call's segments between
        # that stream and an internal one).  Calls are queued back to back and their event pairs (the context's ring,
        # bracketing each call) read afterwards, as for the headline kernel: a synchronize between the calls lets the chip
        # idle and clock down, and round 3's figure of a single synchronised call was up to 10 % above the same call in a loop.
        ncalls = 24 if cfg == 3 else 6
        # device warm-up as for the headline region: the data set-up above left the chip idle for hundreds of milliseconds and
        # it leaves its idle power state over tens (a dozen 1 ms calls right after it ride the clock ramp: +8 % per call)
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) < 0.1:
            for _ in range(4 if cfg == 3 else 1):
                bt.run_torch(kp, npers, out=out)
            torch.cuda.synchronize(dev)
        for _ in range(2):
            bt.run_torch(kp, npers, out=out)
        bt.ctx.set_timing(True)
        for _ in range(ncalls):
            bt.run_torch(kp, npers, out=out)
        ms = bt.ctx.timing_collect()
        bt.ctx.set_timing(False)
        torch.cuda.synchronize(dev)
        cnt = out["count"].cpu().numpy()
        m = float(np.median(ms))
        counts = bt.ctx.last_stream_counts()
        # throughput with two calls in flight (two contexts on two streams, as the headline `value` is issued): the
        # latency-bound kernels of one call (k_associate, the member lists) run beside the VALU-bound ones of the other
        bt2 = BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=odt, device=device_index, method=meth, zero_fill=zero_fill)
        out2 = bt2.alloc_outputs(F, dev)
        if not zero_fill:          # (unspecified slots: compare like with like)
            out2["xyzs"].copy_(out["xyzs"])
        streams2 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        pairs2 = ((bt, out, streams2[0]), (bt2, out2, streams2[1]))
        def two_stream_round(calls):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(calls):
                b, o, st = pairs2[i & 1]
                b.run_torch(kp, npers, out=o, stream=st.cuda_stream)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / calls
        two_stream_round(2)
        ms2 = [two_stream_round(6) * 1e3 for _ in range(3)]
        m2 = float(np.median(ms2))
        same = bool((out2["count"] == out["count"]).all()) and bool(torch.equal(out2["xyzs"], out["xyzs"]))
        bt2.close()
        del out2
        kc = C * (C - 1) // 2 * P * P
        solves = F * kc * J / (m * 1e-3)
        persons = float(cnt.mean())
        bpf = 12 * C * P * J + (16 if odt == np.float32 else 32) * persons * J
        tflops = solves * FLOP_PER_SOLVE / 1e12
        # one fused call = the launches of the streaming association (snowtri_last_kernel_names lists them);
        # kernel_ms = HIP events around the whole call
        kernels = bt.ctx.last_kernel_names()
        handed = bt.ctx.last_handover_persons()
        osz = 4 if odt == np.float32 else 8
        res.append({"workload": label, "kernel": kernels, "frames": F, "kernel_ms": m, "io": "fp32 in / %s out, fp64 math" % ("fp32" if odt == np.float32 else "fp64"),
                    "method": "dlt" if meth == _lib.DLT else "pairwise", "pout_max": pout, "zero_fill": zero_fill,
                    # what the call WRITES to out_xyzs + out_pscore per frame: every slot (the default) / the persons only
                    "output_bytes_written_per_frame": (pout if zero_fill else persons) * (J * 4 * osz + osz),
                    "output_bytes_of_persons_per_frame": persons * (J * 4 * osz + osz),
                    "persons_handed_to_cluster_kernels_last_segment": {"complete_graph": handed[0], "member_list": handed[1]},
                    "kernel_ms_all": ms, "frames_per_s": F / (m * 1e-3),
                    "how": f"100 ms of untimed calls (device warm-up), then {ncalls} calls queued back to back on one stream, HIP events around each call, median",
                    "fall_back_frames_last_segment": {"second_association_launch": counts[0], "exact_candidate_sums": counts[1],
                                                      "k_frame_recompute": counts[2]},
                    "two_streams": {"ms_per_call": m2, "ms_per_call_all": ms2, "frames_per_s": F / (m2 * 1e-3),
                                    "frac": tflops * (m / m2) / FP64_VALU_PEAK_TFLOPS, "outputs_identical_on_both_contexts": same,
                                    "how": "two contexts on two streams, six calls alternating, wall time between two "
                                           "device synchronisations / calls (host launch time included)"},
                    "output_joints_per_s": float(cnt.clip(max=pout).sum()) * J / (m * 1e-3),
                    "pair_solves_per_s": solves, "mean_persons_per_frame": persons,
                    "roofline": {"bound": "fp64_valu", "achieved": tflops, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": tflops / FP64_VALU_PEAK_TFLOPS,
                                 "hbm_GBs": bpf * F / (m * 1e-3) / 1e9, "hbm_frac": bpf * F / (m * 1e-3) / 1e9 / HBM_PEAK_GBS}})
        bt.close()
        del kp, npers, out
    # measured VALU utilisation of the kernels of these calls (rocprofv3 SQ counters of scripts/pmc_multi.sh, profiles/pmc_multi.json),
    # quoted beside the NOMINAL roofline fraction only while the file carries the hash of these kernel sources
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_multi.json")))
        if pm.get("source_sha256") == kernel_source_hash():
            for e, cfg in zip(res, ("cfg3", "cfg5", "cfg3", "cfg3", "cfg3")):
                if "float64" in e["workload"] or e["method"] != "pairwise" or not e["zero_fill"]:
                    continue
                e["roofline"]["valu_busy"] = pm["workloads"].get(cfg)
                e["roofline"]["valu_busy_note"] = ("SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) per kernel of the call, one stream, "
                                                   "rocprofv3 (profiles/pmc_multi.json: source_sha256 matches); `frac` above prices the REFERENCE's "
                                                   "90 flop per candidate joint, the kernels execute about a third of them")
    except Exception:
        pass
    # ONE detection per camera on other shapes than the headline's: the reference's own output type on the headline rig
    # (k_fused_single<4,0,float,double>), and rigs of 6 and 8 cameras (the lean kernels on the complete-graph item, float32;
    # the streaming route without a candidate pass, float64) -- 15 / 28 pair solves per joint: fp64-VALU-bound
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from bench_single_rigs import measure
    for C_, rig_, odt in ((4, "floor", np.float64), (6, "ring", np.float32), (8, "ring", np.float32), (6, "ring", np.float64), (8, "ring", np.float64)):
        e = measure(C_, 10000, odt, rig=rig_, device_index=device_index)
        e["kernel"] = e.pop("kernels")
        e["kernel_ms"] = e["ms_per_call"]
        e["io"] = "fp32 in / %s out, fp64 math" % ("fp32" if odt == np.float32 else "fp64")
        res.append(e)
    # north_star's kernel (row N3): N-view DLT, one detection per camera, on the headline rig and on eight cameras --
    # k_fused_single<C,1,...>: A^T A accumulated per joint (64 flop per observation), Cholesky + shifted inverse iteration
    # (~60 + 60 per step, four steps typical), dehomogenise: 64 C + 310 flop per joint (stated model; the kernel's own count).
    # Four cameras: 64 B per joint -- against HBM like the headline; eight: 112 B, fp64 VALU.
    for C_, rig_ in ((4, "floor"), (8, "ring")):
        e = measure(C_, 10000, np.float32, rig=rig_, device_index=device_index, method=_lib.DLT)
        e["kernel"] = e.pop("kernels")
        e["kernel_ms"] = e["ms_per_call"]
        e["io"] = "fp32 in / fp32 out, fp64 math"
        res.append(e)
    return res


if __name__ == "__main__":
    main()

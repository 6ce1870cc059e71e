"""The 8-rank shapes of BASELINE configs[3] / configs[4] on CPU: world_size-8 gloo groups through the PRODUCT's sharding
code (snowmocap_amd/sharded.py) -- frame counts that do not divide by 8, empty trailing shards through the real
gather, `chunks = auto` on the configs[3] shard (125 000 frames per rank), a rank with an oversized block (nobody may
hang), the sharded smoothing exchange over 8 shards with an empty one, and bench.py's 8-rank dry run.  Stand-ins
replace the kernels (this is about the partitioning, the packing and the collectives): no GPU is needed."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from snowmocap_amd.sharded import auto_chunks, shard_bounds

WORLD = 8


def _init(rank, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)


def _port(salt):
    return 23000 + (os.getpid() * 7 + salt * 131) % 5000


def _gather_worker(rank, port, F, chunks, tmp):
    _init(rank, port)
    from snowmocap_amd.sharded import gather_track_chunked
    lo0, hi0, per = shard_bounds(F, WORLD, rank)
    calls = []

    def compute_block(lo, hi, views):      # stand-in for the kernels: every value names its GLOBAL frame and its rank
        calls.append((lo, hi))
        g = torch.arange(lo0 + lo, lo0 + hi)
        views["xyzs"][: hi - lo] = g.to(torch.float32).view(-1, 1, 1, 1) + torch.tensor([0.0, 0.25, 0.5, 0.75]).view(1, 1, 1, 4)
        views["pscore"][: hi - lo] = float(rank)
        views["count"][: hi - lo] = (g % 5).to(torch.int32).view(-1)
        views["flags"][: hi - lo] = 4

    regions = {"xyzs": ((1, 3, 4), torch.float32), "pscore": ((1,), torch.float32), "count": ((), torch.int32), "flags": ((), torch.int32)}
    ch = auto_chunks(per) if chunks == "auto" else chunks
    out = gather_track_chunked(compute_block, hi0 - lo0, F, regions, chunks=ch)
    ok = sum(b - a for a, b in calls) == hi0 - lo0 and all(b > a for a, b in calls)
    g = torch.arange(F)
    ok = ok and tuple(out["xyzs"].shape) == (F, 1, 3, 4)
    ok = ok and torch.equal(out["xyzs"][:, 0, 0, :], g.to(torch.float32).view(-1, 1) + torch.tensor([0.0, 0.25, 0.5, 0.75]))
    ok = ok and torch.equal(out["count"], (g % 5).to(torch.int32)) and bool((out["flags"] == 4).all())
    ok = ok and torch.equal(out["pscore"][:, 0], (g // per).to(torch.float32)) and not bool(out["rank_status"].any())
    np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([ok, hi0 - lo0, len(calls)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("F,chunks", [(61, 3), (9, 2), (1, 4), (1000, "auto")])
def test_eight_rank_chunked_gather_uneven_and_empty_trailing_shards(tmp_path, F, chunks):
    """61 frames: blocks of 8, the last rank holds 5; 9 frames: blocks of 2, rank 4 holds one frame and ranks 5-7 NONE
    (they still take part in every collective); 1 frame: only rank 0 has work.  Every rank ends up with the whole track."""
    mp.spawn(_gather_worker, args=(_port(F), F, chunks, str(tmp_path)), nprocs=WORLD, join=True)
    res = [np.load(tmp_path / f"ok{r}.npy") for r in range(WORLD)]
    assert all(r[0] for r in res), res
    assert [int(r[1]) for r in res] == [shard_bounds(F, WORLD, q)[1] - shard_bounds(F, WORLD, q)[0] for q in range(WORLD)]
    assert sum(int(r[1]) for r in res) == F


def _cfg3_worker(rank, port, tmp):
    _init(rank, port)
    from snowmocap_amd.sharded import gather_track_chunked
    F = 1_000_000                                    # BASELINE configs[3]: 125 000 frames per rank
    lo0, hi0, per = shard_bounds(F, WORLD, rank)
    pieces = []

    def compute_block(lo, hi, views):
        pieces.append(hi - lo)
        views["xyzs"][: hi - lo] = torch.arange(lo0 + lo, lo0 + hi, dtype=torch.float32).view(-1, 1)
        views["count"][: hi - lo] = 1

    regions = {"xyzs": ((1,), torch.float32), "count": ((), torch.int32)}
    ch = auto_chunks(per)
    out = gather_track_chunked(compute_block, hi0 - lo0, F, regions, chunks=ch)
    ok = per == 125000 and ch == 3 and len(pieces) == 3 and sum(pieces) == per and min(pieces) >= 32768
    ok = ok and torch.equal(out["xyzs"][:, 0], torch.arange(F, dtype=torch.float32)) and bool((out["count"] == 1).all())
    np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_configs3_shard_is_gathered_in_auto_pieces(tmp_path):
    """`chunks = auto` on the shard one rank holds in configs[3] (1 000 000 frames over 8 GPUs): three pieces of >= 32 768
    frames, three collectives, the million-frame track in order on every rank."""
    mp.spawn(_cfg3_worker, args=(_port(3), str(tmp_path)), nprocs=WORLD, join=True)
    assert all(np.load(tmp_path / f"ok{r}.npy").all() for r in range(WORLD))


def _oversized_worker(rank, port, tmp):
    _init(rank, port)
    from snowmocap_amd.sharded import gather_track_chunked
    regions = {"xyzs": ((2,), torch.float32)}
    F = 40                                            # blocks of 5
    n_local = 7 if rank == 3 else 5                   # ONLY rank 3 claims more than its block

    def compute_block(lo, hi, views):
        views["xyzs"][: hi - lo] = 1.0

    try:
        gather_track_chunked(compute_block, n_local, F, regions, chunks=2)
        raised = ""
    except ValueError as e:
        raised = str(e)
    # every rank is past the collectives (none hangs) and every rank raised: the offender names its own count, the others
    # name the offender
    good = ("holds 7 frames" in raised) if rank == 3 else ("rank(s) [3]" in raised)
    np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([good]))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_with_an_oversized_block_fails_everywhere_without_a_hang(tmp_path):
    """the block-size check used to raise on the offending rank BEFORE the collective, leaving the other ranks
    waiting in it.  The offender now takes part with an empty block and a status word; every rank raises afterwards."""
    mp.spawn(_oversized_worker, args=(_port(5), str(tmp_path)), nprocs=WORLD, join=True)
    assert all(np.load(tmp_path / f"ok{r}.npy").all() for r in range(WORLD))


def _compact_worker(rank, port, F, chunks, P, tmp):
    _init(rank, port)
    from snowmocap_amd.sharded import gather_track_compact, compact_to_padded
    kn = 3
    lo0, hi0, per = shard_bounds(F, WORLD, rank)

    def persons_of(g):                     # ragged person counts per GLOBAL frame, some beyond the slots (overflow frames)
        return (g * 7 + g // 3) % (P + 2)

    def compute_block(lo, hi, views):      # stand-in for the kernels: padded outputs, zeros in the unused slots
        g = torch.arange(lo0 + lo, lo0 + hi)
        cnt = persons_of(g)
        views["count"][: hi - lo] = cnt.to(torch.int32)
        views["flags"][: hi - lo] = (g % 3).to(torch.int32)
        x = g.to(torch.float32).view(-1, 1, 1, 1) * 100 + torch.arange(P, dtype=torch.float32).view(1, P, 1, 1) * 10 \
            + torch.arange(kn, dtype=torch.float32).view(1, 1, kn, 1) + torch.tensor([0.0, 0.25, 0.5, 0.75]).view(1, 1, 1, 4)
        used = torch.arange(P).view(1, P) < cnt.view(-1, 1)
        views["xyzs"][: hi - lo] = x * used.view(-1, P, 1, 1)
        views["pscore"][: hi - lo] = (g.view(-1, 1) + 0.5 * torch.arange(P).view(1, P)).to(torch.float32) * used

    out = gather_track_compact(compute_block, hi0 - lo0, F, kn, P, chunks=chunks)
    g = torch.arange(F)
    cnt = persons_of(g)
    sto = cnt.clamp(max=P)
    ok = torch.equal(out["count"], cnt.to(torch.int32)) and torch.equal(out["stored"], sto.to(torch.int32))
    ok = ok and torch.equal(out["flags"], (g % 3).to(torch.int32)) and not bool(out["rank_status"].any())
    ok = ok and int(out["persons"].shape[0]) == int(sto.sum()) and tuple(out["persons"].shape[1:]) == (kn, 4)
    pad = compact_to_padded(out, P)
    used = torch.arange(P).view(1, P) < cnt.view(-1, 1)
    want = (g.to(torch.float32).view(-1, 1, 1, 1) * 100 + torch.arange(P, dtype=torch.float32).view(1, P, 1, 1) * 10
            + torch.arange(kn, dtype=torch.float32).view(1, 1, kn, 1) + torch.tensor([0.0, 0.25, 0.5, 0.75]).view(1, 1, 1, 4)) * used.view(-1, P, 1, 1)
    ok = ok and torch.equal(pad["xyzs"], want)
    ok = ok and torch.equal(pad["pscore"], (g.view(-1, 1) + 0.5 * torch.arange(P).view(1, P)).to(torch.float32) * used)
    # what this rank received: the persons + 8 bytes per frame slot, not Pout_max slots per frame
    padded_bytes = WORLD * per * (P * kn * 16 + P * 4 + 8)
    np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([ok, out["gather_bytes"], padded_bytes, int(sto.sum())]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("F,chunks,P", [(61, 3, 4), (9, 2, 2), (1, 4, 3), (1000, 2, 32), (40, 1, 8)])
def test_eight_rank_compact_gather_of_ragged_person_counts(tmp_path, F, chunks, P):
    """gather_track_compact ("or gather compacted persons + counts"): per piece the counts, then the persons packed
    by prefix sum; ragged counts incl. frames without persons and frames beyond the slots, uneven and empty trailing blocks.
    The padded track rebuilt from it equals the padded gather; the bytes received follow the persons, not the slots."""
    mp.spawn(_compact_worker, args=(_port(F + P), F, chunks, P, str(tmp_path)), nprocs=WORLD, join=True)
    res = [np.load(tmp_path / f"ok{r}.npy") for r in range(WORLD)]
    assert all(r[0] for r in res), res
    if F == 1000:      # 32 slots for ~15 persons per frame (uniform 0..33 clipped): well under half the padded bytes
        assert all(r[1] < 0.75 * r[2] for r in res), res


# ---------------------------------------------------------------------------------- sharded smoothing, 8 shards
def _coeffs(f, z, r, dt):
    pi = np.pi
    k1, k2, k3 = z / (pi * f), 1 / (2 * pi * f) ** 2, r * z / (2 * pi * f)
    A = np.array([[1.0, dt], [-dt / k2, 1 - dt * dt / k2 - dt * k1 / k2]])
    return A, dt / k2, k3 / k2


def _smooth_worker(rank, port, T, tmp):
    _init(rank, port)
    from snowmocap_amd.sharded import combine_carries, smooth_exchange
    f, z, r, dt = 2.5, 0.75, 0.6, 1 / 30
    A, cx, cxd = _coeffs(f, z, r, dt)
    n = 7
    x = np.cumsum(np.random.default_rng(3).normal(0, 0.01, size=(T, n)), axis=0) + 1.0      # the same track on every rank
    lo, hi, _ = shard_bounds(T, WORLD, rank)

    # NumPy stand-ins with the conventions of snowtri_smooth_shard_local / _fix (the host twins of the kernels)
    def local_fn(xl, first, y, payload):
        xv = xl.numpy()
        s = np.zeros((n, 2))
        yv = np.zeros_like(xv)
        tb = 1 if first else 0
        if first:
            yv[0] = xv[0]
        xp = xv[0].copy()
        for t in range(tb, xv.shape[0]):
            c = cx * xv[t] + cxd * (xv[t] - xp)
            xp = xv[t]
            s = s @ A.T + np.stack([np.zeros(n), c], axis=1)
            yv[t] = s[:, 0]
        y.copy_(torch.from_numpy(yv))
        payload[: 2 * n] = torch.from_numpy(s.reshape(-1))

    def combine_fn(allp, rk, start):
        a = allp.numpy()
        payloads = [(a[q, : 2 * n].reshape(n, 2), a[q, 2 * n:3 * n], a[q, 3 * n:4 * n], a[q, 4 * n]) for q in range(WORLD)]
        start.copy_(torch.from_numpy(combine_carries(payloads, rk, A, cxd)))

    def fix_fn(y, first, start):
        v = start.numpy().copy()
        yv = y.numpy()
        for t in range(1 if first else 0, yv.shape[0]):
            v = v @ A.T
            yv[t] += v[:, 0]

    y = smooth_exchange(torch.from_numpy(x[lo:hi].copy()), local_fn, combine_fn, fix_fn)
    np.save(os.path.join(tmp, f"y{rank}.npy"), y.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T", [41, 100, 3])
def test_eight_rank_sharded_smoothing_exchange_matches_the_sequential_filter(tmp_path, T):
    """smooth_exchange (the protocol smooth_track_sharded runs around the C ABI) over 8 gloo ranks with the host twins of
    the kernels: 41 frames = blocks of 6, rank 6 holds 5 and rank 7 NONE; 3 frames = five empty shards.  One all-gather
    of 4n + 1 doubles; the concatenated blocks equal the unsharded recurrence."""
    from oracle import oracle as orc
    mp.spawn(_smooth_worker, args=(_port(T + 11), T, str(tmp_path)), nprocs=WORLD, join=True)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(WORLD)])
    x = np.cumsum(np.random.default_rng(3).normal(0, 0.01, size=(T, 7)), axis=0) + 1.0
    want = orc.second_order_track(x, 2.5, 0.75, 0.6, 1 / 30)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11)


def _blender_worker(rank, port, T, P, tmp):
    """hold_exchange + smooth_exchange (what blender_smooth_sharded runs around the C ABI) with NumPy twins of the kernels and
    PER-LANE coefficients: lane = (person, control point, component), coefficients per control point."""
    _init(rank, port)
    from snowmocap_amd.sharded import combine_carries, hold_exchange, smooth_exchange
    rng = np.random.default_rng(21)
    pts = np.cumsum(rng.normal(0, 0.01, size=(T, P, 24, 4)), axis=0) + rng.uniform(-1, 1, size=(1, P, 24, 4))
    val = (rng.uniform(size=(T, P, 24)) > 0.25).astype(np.uint8)
    val[0, 0, :5] = 0                         # invalid points in the very first frame: seeded with zeros
    val[:, P - 1, 7] = 0                      # a point that is never valid
    val[T // 3: 2 * T // 3, 0, 9] = 0         # a gap longer than a block
    fzr = np.stack([rng.uniform(1.0, 4.0, 24), rng.uniform(0.4, 1.2, 24), rng.uniform(-0.5, 1.0, 24)], axis=1)
    dt = 1 / 30
    n = P * 96
    coef = [_coeffs(*fzr[b], dt) for b in range(24)]
    lane_b = (np.arange(n) // 4) % 24
    A = np.stack([coef[b][0] for b in lane_b])            # [n, 2, 2]
    cx = np.array([coef[b][1] for b in lane_b])
    cxd = np.array([coef[b][2] for b in lane_b])
    lo, hi, per = shard_bounds(T, WORLD, rank)
    grp = np.arange(n) // 4                               # validity group of a lane

    def last_fn(p, v, payload):
        pv, vv = p.numpy().reshape(-1, n), v.numpy().reshape(-1, n // 4)
        out = np.zeros(2 * n)
        for ln in range(n):
            ok = np.nonzero(vv[:, grp[ln]])[0]
            if ok.size:
                out[ln], out[n + ln] = pv[ok[-1], ln], 1.0
        payload.copy_(torch.from_numpy(out))

    def apply_fn(allp, rk, p, v, held):
        a = allp.numpy()
        ent = np.zeros(n)
        for q in range(rk):
            f = a[q, n:] != 0
            ent[f] = a[q, :n][f]
        pv, vv = p.numpy().reshape(-1, n), v.numpy().reshape(-1, n // 4)
        h = np.empty_like(pv)
        cur = ent
        for t_ in range(pv.shape[0]):
            cur = np.where(vv[t_, grp] != 0, pv[t_], cur)
            h[t_] = cur
        held.copy_(torch.from_numpy(h.reshape(held.shape)))

    held = hold_exchange(torch.from_numpy(pts[lo:hi].copy()), torch.from_numpy(val[lo:hi].copy()), last_fn, apply_fn)

    def local_fn(xl, first, y, payload):
        xv = xl.numpy().reshape(-1, n)
        s_ = np.zeros((n, 2))
        yv = np.zeros_like(xv)
        if first:
            yv[0] = xv[0]
        xp = xv[0].copy()
        for t_ in range(1 if first else 0, xv.shape[0]):
            c = cx * xv[t_] + cxd * (xv[t_] - xp)
            xp = xv[t_]
            s_ = np.einsum("nij,nj->ni", A, s_) + np.stack([np.zeros(n), c], axis=1)
            yv[t_] = s_[:, 0]
        y.copy_(torch.from_numpy(yv.reshape(y.shape)))
        payload[: 2 * n] = torch.from_numpy(s_.reshape(-1))

    def combine_fn(allp, rk, start):
        a = allp.numpy()
        out = np.zeros((n, 2))
        for ln in range(n):                   # lane by lane with the lane's own A (combine_carries is the one-coefficient twin)
            pay = [(a[q, 2 * ln:2 * ln + 2].reshape(1, 2), a[q, 2 * n + ln:2 * n + ln + 1], a[q, 3 * n + ln:3 * n + ln + 1], a[q, 4 * n])
                   for q in range(WORLD)]
            out[ln] = combine_carries(pay, rk, A[ln], cxd[ln])[0]
        start.copy_(torch.from_numpy(out))

    def fix_fn(y, first, start):
        v = start.numpy().copy()
        yv = y.numpy().reshape(-1, n)
        for t_ in range(1 if first else 0, yv.shape[0]):
            v = np.einsum("nij,nj->ni", A, v)
            yv[t_] += v[:, 0]

    y = smooth_exchange(held, local_fn, combine_fn, fix_fn)
    if rank == 0 and hi > lo:
        y[0] = torch.from_numpy(pts[0])
    np.save(os.path.join(tmp, f"y{rank}.npy"), y.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T,P", [(41, 2), (100, 1), (5, 1)])
def test_eight_rank_sharded_blender_smoothing_matches_the_sequential_filters(tmp_path, T, P):
    """Row N2 sharded (round-4 review, item 4b): hold exchange (last valid input of every block) + carry exchange with per-bone
    coefficients, over 8 gloo ranks with empty trailing blocks, invalid first points, a never-valid point and a gap longer
    than a block -- against oracle/blender.py::smooth_track (the reference's frame-by-frame filters, pinned on G6 / G8)."""
    from oracle import blender as ob
    mp.spawn(_blender_worker, args=(_port(T + 31 * P), T, P, str(tmp_path)), nprocs=WORLD, join=True)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(WORLD)])
    rng = np.random.default_rng(21)
    pts = np.cumsum(rng.normal(0, 0.01, size=(T, P, 24, 4)), axis=0) + rng.uniform(-1, 1, size=(1, P, 24, 4))
    val = (rng.uniform(size=(T, P, 24)) > 0.25).astype(np.uint8)
    val[0, 0, :5] = 0
    val[:, P - 1, 7] = 0
    val[T // 3: 2 * T // 3, 0, 9] = 0
    fzr = np.stack([rng.uniform(1.0, 4.0, 24), rng.uniform(0.4, 1.2, 24), rng.uniform(-0.5, 1.0, 24)], axis=1)
    want = ob.smooth_track(pts, val, fzr, 1 / 30)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)


def _smooth2_worker(rank, port, T, tmp):
    """smooth_exchange2 -- the two-pass protocol: reduce (the block's zero-state end state), combine, scan (the block from its
    true entering state) -- with the host twins of the kernels."""
    _init(rank, port)
    from snowmocap_amd.sharded import combine_carries, smooth_exchange2
    f, z, r, dt = 2.5, 0.75, 0.6, 1 / 30
    A, cx, cxd = _coeffs(f, z, r, dt)
    n = 7
    x = np.cumsum(np.random.default_rng(3).normal(0, 0.01, size=(T, n)), axis=0) + 1.0
    lo, hi, _ = shard_bounds(T, WORLD, rank)

    def run(xv, first, s):                      # the recurrence of a block from state s; returns (y, end state)
        yv = np.zeros_like(xv)
        if first:
            yv[0] = xv[0]
        xp = xv[0].copy()
        for t in range(1 if first else 0, xv.shape[0]):
            c = cx * xv[t] + cxd * (xv[t] - xp)
            xp = xv[t]
            s = s @ A.T + np.stack([np.zeros(n), c], axis=1)
            yv[t] = s[:, 0]
        return yv, s

    def reduce_fn(xl, first, payload):
        payload[: 2 * n] = torch.from_numpy(run(xl.numpy(), first, np.zeros((n, 2)))[1].reshape(-1))

    def combine_fn(allp, rk, start):
        a = allp.numpy()
        payloads = [(a[q, : 2 * n].reshape(n, 2), a[q, 2 * n:3 * n], a[q, 3 * n:4 * n], a[q, 4 * n]) for q in range(WORLD)]
        start.copy_(torch.from_numpy(combine_carries(payloads, rk, A, cxd)))

    def scan_fn(xl, first, start, y):
        y.copy_(torch.from_numpy(run(xl.numpy(), first, start.numpy().copy())[0]))

    y = smooth_exchange2(torch.from_numpy(x[lo:hi].copy()), reduce_fn, combine_fn, scan_fn)
    np.save(os.path.join(tmp, f"y{rank}.npy"), y.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T", [41, 3])
def test_eight_rank_two_pass_smoothing_exchange_matches_the_sequential_filter(tmp_path, T):
    from oracle import oracle as orc
    mp.spawn(_smooth2_worker, args=(_port(T + 77), T, str(tmp_path)), nprocs=WORLD, join=True)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(WORLD)])
    x = np.cumsum(np.random.default_rng(3).normal(0, 0.01, size=(T, 7)), axis=0) + 1.0
    np.testing.assert_allclose(got, orc.second_order_track(x, 2.5, 0.75, 0.6, 1 / 30), rtol=0, atol=1e-11)


def _ragged_worker(rank, port, T, P, tmp):
    """The sharded form of the reference's list-index semantics (triangulation.py:169-171: banks of frame 0, zip truncation) with
    the host twins of the kernels: tracked_counts hands count[0] round, then slot i is filtered over the frames with
    tracked > i -- per rank a block of the slot's own sequence, possibly EMPTY -- with the carry exchange."""
    _init(rank, port)
    from snowmocap_amd.sharded import combine_carries, smooth_exchange, tracked_counts
    f, z, r, dt = 2.5, 0.75, 0.6, 1 / 30
    A, cx, cxd = _coeffs(f, z, r, dt)
    n = 5
    rng = np.random.default_rng(9)
    x = np.cumsum(rng.normal(0, 0.01, size=(T, P, n)), axis=0) + 1.0          # the same track and counts on every rank
    count = rng.integers(0, P + 3, size=T).astype(np.int32)
    count[0] = P - 1                                                          # the banks of frame 0: P - 1 slots are ever tracked
    if T > 9:
        count[5:9] = 0                                                        # frames nobody is tracked in
    lo, hi, _ = shard_bounds(T, WORLD, rank)
    tracked, count0, bits = tracked_counts(torch.from_numpy(count[lo:hi].copy()), T, P, flag_bits=4 if rank == 3 else 0)
    ok = count0 == P - 1 and bits == 4 and np.array_equal(tracked.numpy(), np.minimum(count[lo:hi], count0))

    def local_fn(xl, first, y, payload):
        xv = xl.numpy()
        s = np.zeros((n, 2))
        yv = np.zeros_like(xv)
        if first:
            yv[0] = xv[0]
        xp = xv[0].copy()
        for t in range(1 if first else 0, xv.shape[0]):
            c = cx * xv[t] + cxd * (xv[t] - xp)
            xp = xv[t]
            s = s @ A.T + np.stack([np.zeros(n), c], axis=1)
            yv[t] = s[:, 0]
        y.copy_(torch.from_numpy(yv))
        payload[: 2 * n] = torch.from_numpy(s.reshape(-1))

    def combine_fn(allp, rk, start):
        a = allp.numpy()
        payloads = [(a[q, : 2 * n].reshape(n, 2), a[q, 2 * n:3 * n], a[q, 3 * n:4 * n], a[q, 4 * n]) for q in range(WORLD)]
        start.copy_(torch.from_numpy(combine_carries(payloads, rk, A, cxd)))

    def fix_fn(y, first, start):
        v = start.numpy().copy()
        yv = y.numpy()
        for t in range(1 if first else 0, yv.shape[0]):
            v = v @ A.T
            yv[t] += v[:, 0]

    out = np.zeros((hi - lo, P, n))
    holds0 = lo == 0 and hi > 0
    for i in range(count0):                                                   # every rank walks the same slots: the exchanges are collectives
        idx = np.nonzero(tracked.numpy() > i)[0]
        xi = torch.from_numpy(x[lo:hi][idx, i].copy())
        yi = smooth_exchange(xi, local_fn, combine_fn, fix_fn, first=holds0 and len(idx) > 0)
        out[idx, i] = yi.numpy()
    np.save(os.path.join(tmp, f"y{rank}.npy"), out)
    np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T,P", [(97, 4), (11, 3), (5, 2)])
def test_eight_rank_list_index_semantics_of_a_varying_person_count(tmp_path, T, P):
    """Against the reference's own protocol run frame by frame: filter banks created at frame 0 for its persons, `zip(persons,
    banks)` per later frame -- a bank is stepped only in frames that carry its slot (its time stands still otherwise), persons
    beyond the banks are dropped.  11 and 5 frames over 8 ranks: blocks of 2 / 1 frames and EMPTY trailing blocks, and ranks
    whose frames carry no person of a slot at all."""
    from oracle import oracle as orc
    mp.spawn(_ragged_worker, args=(_port(T * 3 + P), T, P, str(tmp_path)), nprocs=WORLD, join=True)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(WORLD)])
    assert all(np.load(tmp_path / f"ok{r}.npy").all() for r in range(WORLD))
    rng = np.random.default_rng(9)
    x = np.cumsum(rng.normal(0, 0.01, size=(T, P, 5)), axis=0) + 1.0
    count = rng.integers(0, P + 3, size=T).astype(np.int32)
    count[0] = P - 1
    if T > 9:
        count[5:9] = 0
    want = np.zeros_like(x)
    for i in range(P - 1):                                                    # bank i sees exactly the frames whose list is long enough
        idx = np.nonzero(np.minimum(count, P - 1) > i)[0]
        want[idx, i] = orc.second_order_track(x[idx, i], 2.5, 0.75, 0.6, 1 / 30)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11)
    assert not got[:, P - 1].any()                                            # the slot frame 0 did not fill is never tracked


def test_bench_dry_run_with_eight_ranks():
    """`python bench.py --gpus 8 --dry-run`: the launcher logic and bench's gather leg (gather_track_chunked) with the rank
    count the driver's SCALE run uses."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--chunks", "auto,3", "--frames", "43"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["dry_run"] and d["n_gpus"] == 8 and d["rccl_ranks"] == 8
    assert sorted(r["rank"] for r in d["ranks"]) == list(range(8)) and len({r["pid"] for r in d["ranks"]}) == 8
    for chunks, g in d["gather_leg"].items():
        assert g["ok"] and g["frames_gathered"] == 8 * 43, (chunks, g)

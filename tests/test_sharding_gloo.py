"""N > 1 path on CPU: world_size-2 gloo.  Each rank resolves its contiguous frame block (with the
oracle standing in for the GPU kernel -- this test is about the sharding arithmetic and the single
all-gather), the gathered track must equal the unsharded result bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from snowmocap_amd.sharded import shard_bounds, gather_track


def test_shard_bounds_cover_and_are_contiguous():
    for F in (0, 1, 7, 10, 10000, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(F, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == F
            for (lo, hi, per), (lo2, _, _) in zip(spans[:-1], spans[1:]):
                assert hi == lo2 and hi - lo <= per
            assert sum(hi - lo for lo, hi, _ in spans) == F


def _worker(rank, world, port, F, tmp):
    sys.path.insert(0, ROOT)
    from snowmocap_amd import synth
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = synth.config_workload(2, F, seed=11)
    K, R, t = wl["rig"]
    prm = orc.make_params(**wl["params"])
    lo, hi, per = shard_bounds(F, world, rank)
    loc = orc.triangulate_condense_batch(K, R, t, wl["kpts"][lo:hi], wl["n_persons"][lo:hi], prm, 1, nthreads=1)
    packed = np.concatenate([loc["xyz"], loc["kscore"][..., None]], axis=-1).astype(np.float32)
    full = gather_track(torch.from_numpy(packed), F)
    cnt = gather_track(torch.from_numpy(loc["count"]), F)
    if rank == 0:
        np.save(os.path.join(tmp, "full.npy"), full.numpy())
        np.save(os.path.join(tmp, "cnt.npy"), cnt.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("F", [37, 64])
def test_two_rank_gather_equals_unsharded(tmp_path, F):
    from snowmocap_amd import synth
    from oracle import oracle as orc
    port = 29500 + (os.getpid() % 500) + F
    mp.spawn(_worker, args=(2, port, F, str(tmp_path)), nprocs=2, join=True)
    wl = synth.config_workload(2, F, seed=11)
    K, R, t = wl["rig"]
    ref = orc.triangulate_condense_batch(K, R, t, wl["kpts"], wl["n_persons"], orc.make_params(**wl["params"]), 1, nthreads=1)
    want = np.concatenate([ref["xyz"], ref["kscore"][..., None]], axis=-1).astype(np.float32)
    got = np.load(tmp_path / "full.npy")
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(np.load(tmp_path / "cnt.npy"), ref["count"])


def _chunked_worker(rank, world, port, F, chunks, tmp):
    sys.path.insert(0, ROOT)
    from snowmocap_amd import synth
    from snowmocap_amd.sharded import gather_track_chunked
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = synth.config_workload(2, F, seed=11)
    K, R, t = wl["rig"]
    prm = orc.make_params(**wl["params"])
    lo0, hi0, per = shard_bounds(F, world, rank)
    calls = []

    def compute_block(lo, hi, views):          # the oracle stands in for the kernel: this test is about the plumbing
        calls.append((lo, hi))
        r = orc.triangulate_condense_batch(K, R, t, wl["kpts"][lo0 + lo:lo0 + hi], wl["n_persons"][lo0 + lo:lo0 + hi], prm, 1, nthreads=1)
        views["xyzs"][: hi - lo] = torch.from_numpy(np.concatenate([r["xyz"], r["kscore"][..., None]], axis=-1).astype(np.float32))
        views["pscore"][: hi - lo] = torch.from_numpy(r["pscore"].astype(np.float32))
        views["count"][: hi - lo] = torch.from_numpy(r["count"].astype(np.int32))
        views["flags"][: hi - lo] = 4

    regions = {"xyzs": ((1, 133, 4), torch.float32), "pscore": ((1,), torch.float32), "count": ((), torch.int32),
               "flags": ((), torch.int32)}
    out = gather_track_chunked(compute_block, hi0 - lo0, F, regions, chunks=chunks, group=None, device=None)
    assert sum(b - a for a, b in calls) == hi0 - lo0 and all(b > a for a, b in calls)
    if rank == 1:       # any rank holds the whole track
        np.savez(os.path.join(tmp, "chunked.npz"), **{k: v.numpy() for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("F,chunks", [(37, 4), (64, 3), (5, 100), (41, 1)])
def test_two_rank_chunked_packed_gather_equals_unsharded(tmp_path, F, chunks):
    """ShardedTriangulator's plumbing (gather_track_chunked): the shard in pieces, every output of a piece in ONE buffer
    and ONE collective, uneven blocks and padded last pieces -- bit-identical to the unsharded result."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    port = 29000 + (os.getpid() % 400) + F + 7 * chunks
    mp.spawn(_chunked_worker, args=(2, port, F, chunks, str(tmp_path)), nprocs=2, join=True)
    wl = synth.config_workload(2, F, seed=11)
    K, R, t = wl["rig"]
    ref = orc.triangulate_condense_batch(K, R, t, wl["kpts"], wl["n_persons"], orc.make_params(**wl["params"]), 1, nthreads=1)
    got = np.load(tmp_path / "chunked.npz")
    want = np.concatenate([ref["xyz"], ref["kscore"][..., None]], axis=-1).astype(np.float32)
    assert got["xyzs"].shape == want.shape and np.array_equal(got["xyzs"], want)
    assert np.array_equal(got["pscore"], ref["pscore"].astype(np.float32))
    assert np.array_equal(got["count"], ref["count"]) and (got["flags"] == 4).all()


# ------------------------------------------------------------------ N1 on a sharded track: carry exchange
def _seq_filter(x, f, z, r, dt):
    from oracle import oracle as orc
    return orc.second_order_track(x, f, z, r, dt)


def _local_and_fix_numpy(x, first, A, cx, cxd):
    """NumPy stand-in for snowtri_smooth_shard_local: zero-state response + end state (same convention)."""
    T, n = x.shape
    y = np.zeros_like(x)
    s = np.zeros((n, 2))
    tb = 1 if first else 0
    if first:
        y[0] = x[0]
    xp = x[0].copy()
    for t in range(tb, T):
        c = cx * x[t] + cxd * (x[t] - xp)
        xp = x[t]
        s = s @ A.T + np.stack([np.zeros(n), c], axis=1)
        y[t] = s[:, 0]
    return y, s


def _homogeneous(T, first, start, A):
    tb = 1 if first else 0
    out = np.zeros((T, start.shape[0]))
    v = start.copy()
    for t in range(tb, T):
        v = v @ A.T
        out[t] = v[:, 0]
    return out


@pytest.mark.parametrize("cuts", [[0, 40, 97, 150], [0, 1, 2, 150], [0, 150, 150, 150]])
def test_carry_combination_reproduces_sequential_filter(cuts):
    """combine_carries (the host math between the two shard calls) on 3 shards == the unsharded recurrence."""
    from snowmocap_amd.sharded import combine_carries
    rng = np.random.default_rng(3)
    f, z, r, dt = 2.5, 0.75, 0.6, 1 / 30
    pi = np.pi
    k1, k2, k3 = z / (pi * f), 1 / (2 * pi * f) ** 2, r * z / (2 * pi * f)
    A = np.array([[1.0, dt], [-dt / k2, 1 - dt * dt / k2 - dt * k1 / k2]])
    cx, cxd = dt / k2, k3 / k2
    x = np.cumsum(rng.normal(0, 0.01, size=(150, 7)), axis=0) + 1.0
    want = _seq_filter(x, f, z, r, dt)
    shards = [x[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    locs = [(_local_and_fix_numpy(sh, q == 0, A, cx, cxd) if len(sh) else (sh, np.zeros((7, 2)))) for q, sh in enumerate(shards)]
    payloads = [(E, sh[0] if len(sh) else np.zeros(7), sh[-1] if len(sh) else np.zeros(7), len(sh)) for (y, E), sh in zip(locs, shards)]
    got = []
    for q, sh in enumerate(shards):
        if not len(sh):
            continue
        start = combine_carries(payloads, q, A, cxd)
        got.append(locs[q][0] + _homogeneous(len(sh), q == 0, start, A))
    np.testing.assert_allclose(np.concatenate(got), want, rtol=0, atol=1e-11)


def _edge_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    from snowmocap_amd.sharded import gather_track_chunked
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    regions = {"xyzs": ((1, 5, 4), torch.float32), "count": ((), torch.int32)}
    never = lambda lo, hi, views: (_ for _ in ()).throw(AssertionError("nothing to compute"))
    empty = gather_track_chunked(never, 0, 0, regions, chunks=4)                      # an empty track
    ok = empty["xyzs"].shape == (0, 1, 5, 4) and empty["count"].shape == (0,) and empty["count"].dtype == torch.int32
    try:                                                                              # 6 frames on 2 ranks: blocks of 3
        gather_track_chunked(never, 4, 6, regions, chunks=2)
        refused = False
    except ValueError as e:
        refused = "contiguous blocks" in str(e)
    # a reused workspace: two calls, the second overwrites the first's tensors in place; results as without it
    ws = {}
    outs = []
    for call in range(2):
        def compute_block(lo, hi, views, _c=call):
            views["xyzs"][: hi - lo] = float(100 * _c + 10 * rank) + torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1)
            views["count"][: hi - lo] = 1 + _c
        lo, hi, per = shard_bounds(7, world, rank)
        got = gather_track_chunked(compute_block, hi - lo, 7, regions, chunks=3, workspace=ws)
        outs.append((got["xyzs"].data_ptr(), got["xyzs"][:, 0, 0, 0].clone(), got["count"].clone()))
    same_buffer = outs[0][0] == outs[1][0]
    want1 = torch.tensor([100.0, 101, 102, 103, 110, 111, 112])
    good = torch.equal(outs[1][1], want1) and torch.equal(outs[0][1], want1 - 100) and bool((outs[1][2] == 2).all())
    if rank == 0:
        np.save(os.path.join(tmp, "edge.npy"), np.array([ok, refused, same_buffer, good]))
    dist.barrier()
    dist.destroy_process_group()


def test_chunked_gather_edges_empty_track_oversized_block_reused_workspace(tmp_path):
    """gather_track_chunked: F_total == 0 returns empty tensors (no division by zero), a rank that holds more than its
    contiguous block is refused instead of silently truncated, a caller-kept workspace is reused without changing
    results."""
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_edge_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert np.load(tmp_path / "edge.npy").all(), np.load(tmp_path / "edge.npy")

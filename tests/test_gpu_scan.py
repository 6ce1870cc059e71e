"""Rows N1 / N2 as ONE pass over HBM (round 6): k_smooth_scan -- a chained scan over 256-frame workgroups of 8 waves x 32 frames
in registers, decoupled look-back between the workgroups of a lane column -- against the sequential recurrence of the CPU oracle
(oracle.second_order_track = triangulation.py:4-22; oracle/blender.py for the per-bone filters with the hold).  Shapes around
every boundary of the kernel: the 32 frames of a wave, the 256 of a workgroup, 64 lanes of a column, one lane, one frame, and a
track long enough (100 001 frames) for hundreds of workgroups per column to chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,n", [(1, 5), (2, 64), (33, 65), (34, 1), (256, 63), (257, 64), (258, 532), (289, 128), (513, 7), (1000, 399),
                                  (5000, 532), (100001, 96)])
def test_one_pass_smoothing_equals_the_sequential_recurrence(T, n):
    import snowmocap_amd as api
    from oracle import oracle as orc
    rng = np.random.default_rng(T * 1000 + n)
    x = np.cumsum(rng.normal(0, 0.01, size=(T, n)), axis=0) + rng.uniform(-3, 3, size=(1, n))      # a walk: metres around the room
    for f, z, r in ((2.0, 0.75, 0.0), (4.5, 0.4, 1.2)):
        want = orc.second_order_track(x, f, z, r, 1 / 30)
        got = api.smooth_track(x, f=f, z=z, r=r, delta_time=1 / 30)
        assert np.abs(got - want).max() < 2e-10, (T, n, f, np.abs(got - want).max())
        assert np.array_equal(got[0], x[0])                                     # frame 0 passes through


@pytest.mark.parametrize("T,m", [(1, 3), (40, 133), (300, 133), (777, 266), (20000, 133)])
def test_joint_track_filters_the_points_and_copies_the_scores(T, m):
    """snowtri_smooth_joint_track: records (x, y, z, score) -- three lanes filtered, the fourth copied bit for bit, also when it
    holds inf / NaN (scores of exact intersections) or zeros (ungated joints)."""
    import ctypes as ct
    from snowmocap_amd import _lib
    from oracle import oracle as orc
    rng = np.random.default_rng(T + m)
    x = np.cumsum(rng.normal(0, 0.01, size=(T, m, 4)), axis=0) + rng.uniform(-3, 3, size=(1, m, 4))
    x[..., 3] = rng.uniform(0, 9, size=(T, m))
    if T > 5:
        x[3, 0, 3] = np.inf
        x[4, 1, 3] = np.nan
        x[5, :, 3] = 0.0
    y = np.empty_like(x)
    ctx = _lib.scratch_context()
    _lib.check(ctx.L.snowtri_smooth_joint_track(ctx.handle, T, m, _lib.ptr(x), 2.5, 0.75, 0.5, 1 / 30, _lib.ptr(y), _lib.HOST, None),
               "snowtri_smooth_joint_track")
    want = orc.second_order_track(x[..., :3], 2.5, 0.75, 0.5, 1 / 30)
    assert np.abs(y[..., :3] - want).max() < 2e-10
    assert np.array_equal(y[..., 3].view(np.int64), x[..., 3].view(np.int64))


@pytest.mark.parametrize("T,P", [(2, 1), (31, 2), (257, 1), (600, 3), (30000, 1)])
def test_one_pass_blender_smoothing_with_the_hold(T, P):
    """The per-bone filters (24 coefficient sets, blender.py:171) with invalid points repeating their filter's previous input
    (:157-160): the held input crosses wave and workgroup boundaries -- runs of invalid points of every length, also from frame
    0 on (a filter whose first point is invalid is seeded with zeros, :172-173)."""
    from snowmocap_amd import blender as bl
    from oracle import blender as ob
    rng = np.random.default_rng(T * 7 + P)
    pts = np.cumsum(rng.normal(0, 0.01, size=(T, P, 24, 4)), axis=0) + rng.uniform(-2, 2, size=(1, P, 24, 4))
    val = (rng.uniform(size=(T, P, 24)) > 0.2).astype(np.uint8)
    for p in range(P):
        for start, length in ((0, 3), (20, 40), (250, 300), (1000, 700)):
            if start < T:
                val[start:start + length, p, (start + p) % 24] = 0            # long runs: across a wave (32), across a workgroup (256)
    val[:, 0, 23] = 0                                                           # a point that is never valid
    pts[~val.astype(bool)] = np.nan
    # (stable filters, as the reference's profile holds them -- f 1.5 ... 3 Hz at 30 fps, configs/blender_smooth_profile.json; above
    # ~4.7 Hz at z = 0.75 the semi-implicit Euler step itself diverges, in the reference as here)
    prof = {nm: [1.5 + 0.1 * i, 0.5 + 0.015 * i, 0.1 * (i % 4)] for i, nm in enumerate(bl.CONTROL_POINT_NAMES)}
    fzr = np.array([prof[nm] for nm in bl.CONTROL_POINT_NAMES])
    want = ob.smooth_track(pts, val, fzr, 1 / 30)
    got = bl.blender_smooth_track(pts, val, prof, 1 / 30)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    fin = np.isfinite(want)
    assert np.abs(got[fin] - want[fin]).max(initial=0.0) < 2e-10


def test_the_scan_is_deterministic_and_reuses_its_scratch():
    """The ticket, the flags and the published states live in the context's scratch and are reset per call: twice the same
    track, then another length, then the first again -- bit for bit."""
    import snowmocap_amd as api
    rng = np.random.default_rng(5)
    a = np.cumsum(rng.normal(0, 0.01, size=(3000, 532)), axis=0)
    b = np.cumsum(rng.normal(0, 0.01, size=(700, 64)), axis=0)
    ya = api.smooth_track(a, f=2.5, z=0.75, r=0.5, delta_time=1 / 30)
    yb = api.smooth_track(b, f=2.5, z=0.75, r=0.5, delta_time=1 / 30)
    for _ in range(3):
        assert np.array_equal(api.smooth_track(a, f=2.5, z=0.75, r=0.5, delta_time=1 / 30), ya)
        assert np.array_equal(api.smooth_track(b, f=2.5, z=0.75, r=0.5, delta_time=1 / 30), yb)


@pytest.mark.parametrize("n", [5, 399])
def test_two_pass_shard_protocol_equals_the_five_pass_one_and_the_sequential_filter(n):
    """snowtri_smooth_shard_reduce / _scan (one read for the carry, one pass for the block) against snowtri_smooth_shard_local /
    _fix (the five-pass form) and the sequential recurrence, on uneven frame blocks -- a one-frame first block, a block of exactly
    one workgroup (256 frames), a long one, an empty one in the middle."""
    import ctypes as ct
    import torch
    from snowmocap_amd import _lib
    from snowmocap_amd.sharded import combine_carries, smooth_coeffs
    from oracle import oracle as orc
    f, z, r, dt = 2.5, 0.75, 0.6, 1 / 30
    rng = np.random.default_rng(n)
    sizes = [1, 256, 0, 1900, 33]
    T = sum(sizes)
    x = np.cumsum(rng.normal(0, 0.01, size=(T, n)), axis=0) + 1.0
    want = orc.second_order_track(x, f, z, r, dt)
    A, cx, cxd = smooth_coeffs(f, z, r, dt)
    ctx = _lib.scratch_context()
    L = ctx.L
    dev = torch.device("cuda", 0)
    bounds = np.cumsum([0] + sizes)
    blocks = [torch.from_numpy(x[bounds[q]:bounds[q + 1]].copy()).to(dev) for q in range(len(sizes))]
    first_q = next(q for q, s_ in enumerate(sizes) if s_ > 0)
    payloads = []
    for q, xb in enumerate(blocks):
        Tq = int(xb.shape[0])
        E = torch.zeros(2 * n, dtype=torch.float64, device=dev)
        E5 = torch.zeros(2 * n, dtype=torch.float64, device=dev)
        if Tq:
            _lib.check(L.snowtri_smooth_shard_reduce(ctx.handle, Tq, n, ct.c_void_p(xb.data_ptr()), 1 if q == first_q else 0, f, z, r, dt,
                                                     ct.c_void_p(E.data_ptr()), None), "reduce")
            y5 = torch.empty_like(xb)
            _lib.check(L.snowtri_smooth_shard_local(ctx.handle, Tq, n, ct.c_void_p(xb.data_ptr()), 1 if q == first_q else 0, f, z, r, dt,
                                                    ct.c_void_p(y5.data_ptr()), ct.c_void_p(E5.data_ptr()), _lib.DEVICE, None), "local")
            torch.cuda.synchronize()
            assert float((E - E5).abs().max()) < 1e-11, q           # the same carry from one read of x
        Eh = E.cpu().numpy().reshape(n, 2)
        payloads.append((Eh, x[bounds[q]] if Tq else np.zeros(n), x[bounds[q + 1] - 1] if Tq else np.zeros(n), Tq))
    got = np.zeros_like(x)
    for q, xb in enumerate(blocks):
        Tq = int(xb.shape[0])
        if not Tq:
            continue
        start = torch.from_numpy(combine_carries(payloads, q, A, cxd)).to(dev)
        yb = torch.empty_like(xb)
        _lib.check(L.snowtri_smooth_shard_scan(ctx.handle, Tq, n, ct.c_void_p(xb.data_ptr()), 1 if q == first_q else 0, ct.c_void_p(start.data_ptr()),
                                               f, z, r, dt, ct.c_void_p(yb.data_ptr()), None), "scan")
        torch.cuda.synchronize()
        got[bounds[q]:bounds[q + 1]] = yb.cpu().numpy()
    assert np.abs(got - want).max() < 2e-10, np.abs(got - want).max()


def test_long_tracks_repeat_bit_for_bit_under_thousands_of_chained_workgroups():
    """The look-back reads states other workgroups -- on other XCDs, whose L2s are not coherent with this one's -- published
    through agent-scope atomics, the flag behind the acknowledged state.  A missing ordering would show as a rare wrong carry:
    200 001 frames x 532 lanes = 782 workgroups down each of 9 columns, twenty times, every run equal to the first bit for
    bit and the first equal to the sequential recurrence."""
    import ctypes as ct
    import torch
    from snowmocap_amd import _lib
    from oracle import oracle as orc
    T, n = 200001, 532
    rng = np.random.default_rng(42)
    x = np.cumsum(rng.normal(0, 0.01, size=(T, n)), axis=0) + 1.5
    ctx = _lib.scratch_context()
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x).to(dev)
    first = None
    for rep in range(20):
        yd = torch.empty_like(xd)
        _lib.check(ctx.L.snowtri_smooth_track(ctx.handle, T, n, ct.c_void_p(xd.data_ptr()), 2.5, 0.75, 0.5, 1 / 30, ct.c_void_p(yd.data_ptr()),
                                              _lib.DEVICE, None), "snowtri_smooth_track")
        torch.cuda.synchronize(dev)
        if first is None:
            first = yd
            want = orc.second_order_track(x, 2.5, 0.75, 0.5, 1 / 30)
            assert np.abs(first.cpu().numpy() - want).max() < 2e-10
        else:
            assert torch.equal(yd.view(torch.int64), first.view(torch.int64)), rep

"""k_fused_lean (snowtri_lean.hpp): the production shape of the fast path -- pairwise method, float32 outputs,
one detection per camera, keypoint_num == J == 133, one output slot -- against the CPU oracle, through the C ABI.

The generic tests (test_gpu_parity.py) reach this kernel through the golden 4-camera fixtures and the BASELINE
workloads; the cases here aim at what is specific to it: frames that break the speculation and are re-done by the
in-launch fallback (spread over waves, tiles and tile ordinals), every camera count it is instantiated for, float64
inputs, special values, the launch shapes (one tile per wave / several tiles per wave / many workgroups), and
agreement with k_fused_single on the same inputs.

Tolerances (tests/test_gpu_parity.py): float32 outputs <= 2e-6 m, scores <= 3e-7 relative.
"""
import numpy as np
import pytest

from conftest import assert_scores_close, assert_xyz_close

pytestmark = pytest.mark.gpu

XYZ_F32 = 2e-6
J = 133


@pytest.fixture(scope="module")
def api():
    import snowmocap_amd as sm
    from snowmocap_amd import _lib
    assert _lib.lib().snowtri_device_count() > 0, "these tests need the HIP device"
    return sm


def _run(api, K, R, t, prm, kp, npers, env=None, knobs=None):
    if env:      # (a forced route: the test build of the library, conftest.Knobs)
        for k, v in env.items():
            knobs.set(k, v)
    bt = api.BatchTriangulator(K, R, t, prm, pout_max=1, out_dtype=np.float32)
    out = bt.run_host(kp, npers)
    out["slow"] = bt.ctx.last_slow_frames()
    bt.close()
    if env:
        knobs.clear(*env)
    return out


def _check_frames(out, ref, frames, msg=""):
    for f in frames:
        m = min(int(ref["count"][f]), 1)
        assert out["count"][f] == ref["count"][f], f"{msg} frame {f}: count {out['count'][f]} vs {ref['count'][f]}"
        if m:
            assert_scores_close(out["xyzs"][f, :1, :, 3], ref["kscore"][f, :1], rtol=3e-7, what=f"{msg} kscore frame {f}")
            assert_xyz_close(out["xyzs"][f, :1, :, :3], ref["xyz"][f, :1], XYZ_F32, score_ref=ref["kscore"][f, :1],
                             what=f"{msg} xyz frame {f}")
            assert_scores_close(out["pscore"][f, :1], ref["pscore"][f, :1], rtol=3e-7, nterms=J, what=f"{msg} pscore frame {f}")
        else:
            assert not out["xyzs"][f].any(), f"{msg} frame {f}: an empty frame must be zero-filled"


def _break_some_frames(rng, kp, npers, F, n_each=3):
    """Frames that fail the speculation, each for a different reason; returns {frame: reason}."""
    broken = {}
    picks = rng.choice(F, size=min(F, 5 * n_each), replace=False)
    for i, f in enumerate(picks):
        kind = i % 5
        if kind == 0:
            kp[f, 2, 0, :, :2] += 400.0          # one camera far off: centre joints further apart than the tolerance
        elif kind == 1:
            npers[f, 1] = 0                       # a camera without detection
        elif kind == 2:
            kp[f, :, 0, :, 2] = 0.0               # everything gated: fused mean 0 < condense_score_tol
        elif kind == 3:
            kp[f, 0, 0, 7, 0] = np.nan            # NaN pixel: the IEEE-exact routine decides
        else:
            kp[f, 3, 0, 20:40, :2] += 250.0       # 20 joints of one camera off: distance gate only, stays fast
        broken[int(f)] = kind
    return broken


@pytest.mark.parametrize("F,tiles_per_wave", [(1, None), (3, None), (37, None), (700, None), (4099, None), (30011, None),
                                              (30011, "1"), (52000, "2")])
def test_lean_fallback_frames_against_oracle(api, F, tiles_per_wave, knobs):
    """Broken frames sprinkled over a batch: they lose the fast-path flag, are re-done inside the same launch, and
    every checked frame -- broken or not -- equals the oracle.  F sweeps the launch shapes: fewer frames than waves,
    one tile per wave, several tiles per wave (tile ordinals > 0 in the slow-frame bit words), many workgroups."""
    from snowmocap_amd import synth, _lib
    from oracle import oracle as orc
    rng = np.random.default_rng(1000 + F)
    wl = synth.config_workload(2, F, seed=77 + F)
    kp = wl["kpts"]
    kp[..., 2] = rng.uniform(2.0, 8.0, size=kp.shape[:-1]).astype(np.float32)     # exercise the keypoint gate
    npers = wl["n_persons"].copy()
    broken = _break_some_frames(rng, kp, npers, F) if F >= 3 else {}
    prm = dict(wl["params"], condense_distance_tol=0.5, condense_score_tol=0.2)
    K, R, t = wl["rig"]
    env = {"SNOWTRI_LEAN_TILES_PER_WAVE": tiles_per_wave} if tiles_per_wave else None
    out = _run(api, K, R, t, prm, kp, npers, env, knobs)
    assert out["status"] in (_lib.OK, _lib.ERR_OVERFLOW)      # a broken frame may resolve to more persons than the one slot
    fast = (out["flags"] & _lib.FLAG_FASTPATH) != 0
    for f, kind in broken.items():
        assert fast[f] == (kind == 4), (f, kind)
    others = np.setdiff1d(np.arange(F), np.fromiter(broken.keys(), dtype=np.int64, count=len(broken)))
    assert fast[others].all()
    assert out["slow"] == int((~fast).sum())
    check = sorted(set(broken) | set(rng.choice(F, size=min(F, 48), replace=False).tolist()) | {0, F - 1})
    sel = np.asarray(check)
    ref = orc.triangulate_condense_batch(K, R, t, kp[sel], npers[sel], orc.make_params(**prm), 4)
    sub = {k: (v[sel] if isinstance(v, np.ndarray) and v.shape[:1] == (F,) else v) for k, v in out.items()}
    _check_frames(sub, ref, range(len(sel)), msg=f"F={F}")
    for i, f in enumerate(sel):
        if broken.get(int(f)) == 2:
            assert ref["count"][i] == 0


@pytest.mark.parametrize("C", [3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("in_dtype", [np.float32, np.float64])
def test_lean_camera_counts_and_input_types(api, C, in_dtype):
    """Every instantiation (3..8 cameras, float32 / float64 keypoints): ring rigs with one person vs the oracle;
    a tight condense_distance_tol sends some frames (opposite cameras) through the fallback."""
    from snowmocap_amd import synth, _lib
    from oracle import oracle as orc
    rng = np.random.default_rng(400 + C)
    K, R, t = synth.ring_rig(C)
    F = 90
    X = synth.make_people(rng, F, 1)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(2.0, 8.0), dtype=in_dtype)
    prm = dict(synth.default_thresholds(), condense_distance_tol=0.5)
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 4)
    out = _run(api, K, R, t, prm, kp, npers)
    assert ((out["flags"] & _lib.FLAG_FASTPATH) != 0).mean() > 0.9
    _check_frames(out, ref, range(F), msg=f"C={C} {np.dtype(in_dtype).name}")


def test_lean_random_thresholds_and_person_lists(api):
    """Randomised thresholds / centre index / missing detections on the production shape (J = kn = 133, one slot):
    fast frames and fallback frames, all against the oracle."""
    from snowmocap_amd import synth, _lib
    from oracle import oracle as orc
    rng = np.random.default_rng(4242)
    fast_frames = slow_frames = 0
    for trial in range(24):
        C = int(rng.integers(3, 7))
        F = int(rng.choice([2, 25, 130]))
        K, R, t = synth.ring_rig(C, radius=float(rng.uniform(3, 6)))
        X = synth.make_people(rng, F, 1, J=J)
        kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=float(rng.choice([0.3, 1.0, 4.0])), score_range=(2.0, 8.0),
                                         dtype=np.float64 if trial % 3 == 0 else np.float32)
        npers = npers.copy()
        for _ in range(int(rng.integers(0, 3))):
            npers[rng.integers(0, F), rng.integers(0, C)] = 0
        use_np = trial % 4 != 1
        prm = dict(keypoint_score_threshold=float(rng.choice([0.0, 3.0, 5.0])), average_score_threshold=0.0,
                   distance_threshold=float(rng.choice([0.02, 0.05, 1.0])), condense_distance_tol=float(rng.choice([0.01, 0.05, 10.0])),
                   condense_person_num_tol=int(rng.choice([0, 2, C * (C - 1) // 2])), condense_score_tol=float(rng.choice([0.0, 0.0, 0.8])),
                   center_point_index=int(rng.integers(0, J)), keypoint_num=J)
        if not use_np:
            npers[:] = 1
        ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 32)
        out = _run(api, K, R, t, prm, kp, npers if use_np else None)
        fastf = (out["flags"] & _lib.FLAG_FASTPATH) != 0
        fast_frames += int(fastf.sum())
        slow_frames += int((~fastf).sum())
        _check_frames(out, ref, range(F), msg=f"trial {trial}: C={C} F={F} {prm}")
    assert fast_frames > 200 and slow_frames > 200, (fast_frames, slow_frames)


def test_lean_special_values(api):
    """Exactly intersecting rays (dist == 0 -> inf score -> NaN fused joint in the reference), NaN pixels, NaN /
    negative / zero confidences on the production shape: zero / finite / blown-up classes and counts as the oracle."""
    from oracle import oracle as orc
    rng = np.random.default_rng(515)
    compared = 0
    for trial in range(12):
        C = int(rng.integers(3, 6))
        F = int(rng.integers(1, 5))
        K = np.tile(np.eye(3), (C, 1, 1))
        R = np.tile(np.eye(3), (C, 1, 1))
        t = np.zeros((C, 3))
        t[:, 0] = 2.0 * np.arange(C)
        t[1::2, 1] = 2.0
        # joints on a dyadic grid in front of the cameras: pixels and rays are exactly representable
        X = np.stack([rng.integers(-4, 5, (F, 1, J)) / 2.0, rng.integers(-4, 5, (F, 1, J)) / 2.0,
                      rng.choice([2.0, 4.0, 8.0], (F, 1, J))], axis=-1)
        kp = np.zeros((F, C, 1, J, 3))
        for c in range(C):
            kp[:, c, :, :, 0] = (X[..., 0] - t[c, 0]) / X[..., 2]
            kp[:, c, :, :, 1] = (X[..., 1] - t[c, 1]) / X[..., 2]
        kp[..., 2] = rng.choice([5.0, 5.0, 5.0, 1.0, 0.25, -2.0], size=kp.shape[:-1])
        if trial % 3 == 0:
            kp[..., :2] += rng.normal(0, 1e-3, size=kp[..., :2].shape)          # some trials: near-exact instead
        for _ in range(int(rng.integers(0, 3))):
            kp[rng.integers(0, F), rng.integers(0, C), 0, rng.integers(0, J), rng.integers(0, 3)] = np.nan
        if trial % 2:
            kp = kp.astype(np.float32)
        npers = np.ones((F, C), np.int32)
        prm = dict(keypoint_score_threshold=float(rng.choice([3.0, 0.5])), average_score_threshold=0.0,
                   distance_threshold=float(rng.choice([0.05, 1.0])), condense_distance_tol=float(rng.choice([0.3, 10.0])),
                   condense_person_num_tol=int(rng.choice([0, 2])), condense_score_tol=float(rng.choice([0.0, -1.0, 0.5])),
                   center_point_index=int(rng.integers(0, J)), keypoint_num=J)
        ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 32)
        out = _run(api, K, R, t, prm, kp, npers)
        msg = f"trial {trial}: C={C} F={F} {prm}"
        for f in range(F):
            if ref["status"][f] != 0:                     # singular pair: the reference raises, outputs are unspecified
                assert out["flags"][f] & 1, msg
                continue
            assert out["count"][f] == ref["count"][f], msg
            m = min(int(ref["count"][f]), 1)
            if not m:
                continue
            g, o = out["xyzs"][f, :1, :, 3].astype(np.float64), ref["kscore"][f, :1]

            def classes(s):
                return np.where(~np.isfinite(s) | (np.abs(s) > 1e9), 2, np.where(s == 0.0, 0, 1))
            np.testing.assert_array_equal(classes(g), classes(o), err_msg=msg)
            fin = classes(o) == 1
            np.testing.assert_allclose(g[fin], o[fin], rtol=1e-6, atol=1e-12, err_msg=msg)
            gx, ox = out["xyzs"][f, :1, :, :3], ref["xyz"][f, :1]
            zero = classes(o) == 0
            assert not gx[zero].any() and not ox[zero].any(), msg
            np.testing.assert_allclose(gx[fin], ox[fin], rtol=1e-6, atol=2e-6, err_msg=msg)
            compared += 1
    assert compared > 10


def test_lean_agrees_with_fused_single(api, knobs):
    """SNOWTRI_LEAN_MODE=0 keeps float32-output batches on k_fused_single: same counts / flags, joints within the
    float32 tolerance of each other (the two kernels round 1/dist differently), on a 10 000-frame batch."""
    from snowmocap_amd import synth, _lib
    wl = synth.config_workload(2, 10000, seed=3)
    K, R, t = wl["rig"]
    a = _run(api, K, R, t, wl["params"], wl["kpts"], wl["n_persons"])
    b = _run(api, K, R, t, wl["params"], wl["kpts"], wl["n_persons"], {"SNOWTRI_LEAN_MODE": "0"}, knobs)
    assert np.array_equal(a["count"], b["count"]) and np.array_equal(a["flags"], b["flags"])
    assert (a["count"] == 1).all()
    assert np.abs(a["xyzs"][..., :3].astype(np.float64) - b["xyzs"][..., :3]).max() < XYZ_F32
    s0, s1 = a["xyzs"][..., 3].astype(np.float64), b["xyzs"][..., 3].astype(np.float64)
    # each kernel is within 3e-7 of the float64 score (float32 rounding + its own raw v_rsq_f64 of a different argument)
    assert np.abs(s0 - s1).max() <= 6e-7 * np.abs(s1).max()
    assert np.allclose(a["pscore"], b["pscore"], rtol=6e-7)


def test_lean_is_deterministic_and_split_invariant(api):
    """Same batch twice -> bit-identical; any split of the batch over launches -> bit-identical (tiles are whole
    frames, a frame's result does not depend on its neighbours or on the launch shape)."""
    import torch
    from snowmocap_amd import synth
    F = 26000                                   # several tiles per wave
    wl = synth.config_workload(2, F, seed=9)
    K, R, t = wl["rig"]
    bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float32)
    dev = torch.device("cuda", 0)
    kp = torch.from_numpy(wl["kpts"]).to(dev)
    full = bt.run_torch(kp)
    torch.cuda.synchronize()
    a = full["xyzs"].cpu().numpy().copy()
    ps = full["pscore"].cpu().numpy().copy()
    again = bt.run_torch(kp)
    torch.cuda.synchronize()
    assert np.array_equal(a, again["xyzs"].cpu().numpy()) and np.array_equal(ps, again["pscore"].cpu().numpy())
    cuts = [0, 1, 777, 13000, 13001, 25999, F]
    parts, pparts = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        o = bt.run_torch(kp[lo:hi].contiguous())
        torch.cuda.synchronize()
        parts.append(o["xyzs"].cpu().numpy())
        pparts.append(o["pscore"].cpu().numpy())
    assert np.array_equal(a, np.concatenate(parts)) and np.array_equal(ps, np.concatenate(pparts))
    bt.close()


@pytest.mark.parametrize("chunks", [1, 4, 7])
def test_sharded_triangulator_overlapped_gather_single_rank_group(api, chunks):
    """ShardedTriangulator.run on the GPU in a 1-rank RCCL group: the shard in pieces, the packed all-gather of piece i
    on a side stream under the kernel of piece i + 1 -- bit-identical to one unsharded launch (sharded execution on
    several GPUs is the driver's SCALE run; the 2-rank plumbing runs on CPU in tests/test_sharding_gloo.py)."""
    import os
    import torch
    import torch.distributed as dist
    from snowmocap_amd import synth
    from snowmocap_amd.sharded import ShardedTriangulator
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 200))
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        F = 12501                               # not a multiple of the piece count
        wl = synth.config_workload(2, F, seed=21)
        K, R, t = wl["rig"]
        dev = torch.device("cuda", 0)
        kp = torch.from_numpy(wl["kpts"]).to(dev)
        st = ShardedTriangulator(K, R, t, wl["params"], pout_max=1, device=0, chunks=chunks)
        got = st.run(kp, F)
        torch.cuda.synchronize()
        ref = st.bt.run_torch(kp)
        torch.cuda.synchronize()
        for k in ("xyzs", "pscore", "count", "flags"):
            assert got[k].shape == ref[k].shape and torch.equal(got[k], ref[k]), k
        assert (got["count"] == 1).all()
        st.bt.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_raw_rsq_accuracy_behind_the_float32_score_contract(api):
    """The float32-output kernels take 1/dist from the raw v_rsq_f64 (no Newton step).  Its relative error over the
    normal range, measured: must stay <= 2^-23 (measured 2^-24.2), which with the float32 rounding of the stored score (2^-24) keeps the
    scores inside the 3e-7 the parity tests allow."""
    from snowmocap_amd import _lib
    ctx = _lib.scratch_context()
    rng = np.random.default_rng(1)
    n = 400000
    x = rng.uniform(1, 4, n) * 4.0 ** rng.integers(-200, 201, n)       # every mantissa / exponent parity
    x[:1000] = rng.uniform(1e-12, 1e-2, 1000)                          # dist^2 of real rigs: 1e-12 .. 1e-2 m^2
    rc, rs = np.empty(n), np.empty(n)
    _lib.check(_lib.lib().snowtri_fastmath_probe_raw(ctx.handle, n, _lib.ptr(x), _lib.ptr(rc), _lib.ptr(rs)), "probe_raw")
    err_rsq = np.abs(rs * np.sqrt(x) - 1.0).max()
    err_rcp = np.abs(rc * x - 1.0).max()
    print(f"raw v_rsq_f64 max rel err {err_rsq:.3e} (2^{np.log2(err_rsq):.2f}), raw v_rcp_f64 {err_rcp:.3e} (2^{np.log2(err_rcp):.2f})")
    assert err_rsq <= 2.0 ** -23 and err_rcp <= 2.0 ** -23


@pytest.mark.parametrize("F", [1, 2, 3, 5, 511, 513, 1024, 4000, 10000, 16384, 16385])
def test_small_launches_cooperative_kernel_equals_wave_autonomous_kernel(api, F, knobs):
    """k_fused_lean_coop takes launches of up to 32 frames per resident workgroup (16 384 frames on an MI355X): a
    workgroup tile whose PASSES are dealt to the waves (frames straddle waves), one cooperative epilogue.  Items, check
    and mean are the functions k_fused_lean uses: the same frames through SNOWTRI_LEAN_COOP=0 are bit-identical --
    fall-back frames (every reason, spread over the tiles) included -- and both agree with the oracle."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(9000 + F)
    wl = synth.config_workload(2, F, seed=77)
    K, R, t = wl["rig"]
    kp, npers = wl["kpts"].copy(), wl["n_persons"].copy()
    broken = _break_some_frames(rng, kp, npers, F, n_each=4) if F >= 5 else {}
    if F >= 3:
        kp[F - 1, 2, 0, :, :2] += 400.0          # the batch's last frame (the short end of the last tile) falls back too
        broken[F - 1] = 0
    bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float32)
    a = bt.run_host(kp, npers)
    names = bt.ctx.last_kernel_names()
    bt.close()
    assert names.startswith("k_fused_lean_coop<4,float,133>" if F <= 16384 else "k_fused_lean<4,float,133>"), names
    b = _run(api, K, R, t, wl["params"], kp, npers, {"SNOWTRI_LEAN_COOP": "0"}, knobs)
    for key in ("xyzs", "pscore", "count", "flags"):
        assert np.array_equal(a[key], b[key], equal_nan=True), f"F={F}: {key} differs between the two kernels"
    from snowmocap_amd import _lib
    fast = (a["flags"] & _lib.FLAG_FASTPATH) != 0
    for f, kind in broken.items():      # a camera without detection, a NaN pixel: re-done by the exact routine
        assert kind not in (1, 3) or not fast[f], (f, kind)
    assert fast.sum() >= F - len(broken)
    check = sorted(set(list(broken)[:12]) | set(int(x) for x in rng.choice(F, size=min(F, 24), replace=False)))
    ref = orc.triangulate_condense_batch(K, R, t, kp[check], npers[check], orc.make_params(**wl["params"]), 1)
    sub = {k: a[k][check] for k in ("xyzs", "pscore", "count")}
    _check_frames(sub, ref, range(len(check)), msg=f"F={F}")


@pytest.mark.parametrize("F", [2000, 40000])
def test_event_timing_attached_and_bracketed(api, F):
    """snowtri_set_timing(ctx, 1) brackets every fused call with an event pair; (ctx, 2) attaches the pair to the dispatch of a
    single-kernel call (hipExtLaunchKernelGGL start / stop events: the kernel's own begin and end, no event record between the
    launches).  Both rings return one positive duration per call, the attached one not longer than the bracketed one by
    more than noise, and the outputs do not depend on the mode.  F = 2000: k_fused_lean_coop, 40000: k_fused_lean."""
    import torch
    from snowmocap_amd import synth
    wl = synth.config_workload(2, F, seed=5)
    K, R, t = wl["rig"]
    bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float32)
    dev = torch.device("cuda", 0)
    kp = torch.from_numpy(wl["kpts"].astype(np.float32)).to(dev)
    outs = {}
    means = {}
    for mode, attach in (("bracketed", False), ("attached", True)):
        out = bt.alloc_outputs(F, dev)
        for _ in range(5):
            bt.run_torch(kp, None, out=out)
        bt.ctx.set_timing(True, attach=attach)
        for _ in range(40):
            bt.run_torch(kp, None, out=out)
        ms = bt.ctx.timing_collect()
        bt.ctx.set_timing(False)
        torch.cuda.synchronize(dev)
        assert len(ms) == 40 and all(0.0 < m < 50.0 for m in ms), (mode, ms)
        means[mode] = float(np.mean(ms))
        outs[mode] = {k: v.cpu().numpy() for k, v in out.items()}
    print(f"F={F}: {bt.ctx.last_kernel_names()} bracketed {means['bracketed'] * 1e3:.2f} us, attached {means['attached'] * 1e3:.2f} us")
    assert means["attached"] <= means["bracketed"] * 1.5      # (typically 0.75-0.92; a loose bound: the test is about the plumbing)
    for k in outs["attached"]:
        assert np.array_equal(outs["attached"][k], outs["bracketed"][k], equal_nan=True), k
    bt.close()


@pytest.mark.parametrize("streams", [2, 3])
def test_overlap_mode_gives_a_plain_loop_of_calls_the_same_bits(api, streams):
    """BatchTriangulator(streams=n) = snowtri_ctx_set_overlap: consecutive device calls rotate over n internal streams behind the
    caller's stream; after join() every call's outputs equal the ones of the same calls issued one after the other -- also
    for frames that take the in-launch fall-back (their slabs belong to the stream set, not to the context)."""
    import torch
    from snowmocap_amd import synth, _lib
    rng = np.random.default_rng(900 + streams)
    F = 3000
    wl = synth.config_workload(2, F, seed=5)
    K, R, t = wl["rig"]
    dev = torch.device("cuda", 0)
    batches = []
    for b in range(7):
        kp = wl["kpts"].copy()
        kp[..., :2] += rng.normal(0, 0.3, size=kp.shape[:-1] + (2,)).astype(np.float32)
        npers = wl["n_persons"].copy()
        _break_some_frames(rng, kp, npers, F)
        batches.append((torch.from_numpy(kp).to(dev), torch.from_numpy(npers).to(dev)))
    prm = dict(wl["params"], condense_distance_tol=0.5)
    seq = api.BatchTriangulator(K, R, t, prm, pout_max=1, out_dtype=np.float32)
    want = [{k: v.clone() for k, v in seq.run_torch(kp, npers).items()} for kp, npers in batches]
    torch.cuda.synchronize(dev)
    seq.close()
    ovl = api.BatchTriangulator(K, R, t, prm, pout_max=1, out_dtype=np.float32, streams=streams)
    assert ovl.ctx.overrides() == ""
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):                       # the caller's stream is not the null stream
        for rep in range(2):
            got = [ovl.run_torch(kp, npers) for kp, npers in batches]
            ovl.join()
            flat = [{k: v.clone() for k, v in g.items()} for g in got]     # reads on the caller's stream, behind the join
            side.synchronize()
            for g, w in zip(flat, want):
                for k in ("xyzs", "pscore", "count", "flags"):     # bit patterns: a NaN pixel gives NaN joints in both
                    assert torch.equal(g[k].view(torch.int32), w[k].view(torch.int32)), (rep, k)
    slow = sum(int(((w["flags"] & _lib.FLAG_FASTPATH) == 0).sum()) for w in want)
    assert slow > 0
    ovl.close()

"""ONE detection per camera on rigs of five and more cameras (round 5): the lean kernels on cluster_item (6-8 cameras,
float32 outputs, J = keypoint_num = 133) and, for every other shape, the streaming route WITHOUT its candidate pass
(`sumless`: k_associate -> k_cluster_fuse / _wide / k_cluster_members -> k_person_scores).  All against the CPU oracle
through the C ABI; the route is read back from snowtri_last_kernel_names.

Tolerances as tests/test_gpu_parity.py: float32 outputs <= 2e-6 m (+ one float32 ulp of the value), scores <= 3e-7
relative; float64 outputs <= 1e-8 m, scores <= 1e-9 relative (+ the conditioning term of assert_scores_close).
"""
import numpy as np
import pytest

from conftest import assert_scores_close, assert_xyz_close

pytestmark = pytest.mark.gpu
J = 133


@pytest.fixture(scope="module")
def api():
    import snowmocap_amd as sm
    from snowmocap_amd import _lib
    assert _lib.lib().snowtri_device_count() > 0, "these tests need the HIP device"
    return sm


def _run(api, K, R, t, prm, kp, npers, out_dtype, pout=1, env=None, knobs=None):
    for k, v in (env or {}).items():      # (a forced route: the test build of the library, conftest.Knobs)
        knobs.set(k, v)
    bt = api.BatchTriangulator(K, R, t, prm, pout_max=pout, out_dtype=out_dtype)
    out = bt.run_host(kp, npers)
    out["names"] = bt.ctx.last_kernel_names()
    bt.close()
    if env:
        knobs.clear(*env)
    return out


def _compare(out, ref, F, out_dtype, kn, msg, npairs=1):
    """npairs: members a fused joint score averages over at most (C(C,2) for one detection per camera): the conditioning term of
    assert_scores_close bounds a MEAN of n scores through mean(s^2) <= n mean(s)^2 -- one pair of 120 whose rays pass within
    1e-8 m carries the whole mean, and its 1/dist moves by 1e-7 relative between any two formulations (the IEEE spill kernel
    included: soak rounds 306-357)."""
    f32 = np.dtype(out_dtype) == np.float32
    assert np.array_equal(out["count"], ref["count"]), msg      # (the count is the frame's persons, also beyond the slots: SNOWTRI_FLAG_OVERFLOW)
    for f in range(F):
        m = min(int(ref["count"][f]), out["xyzs"].shape[1])
        assert not out["xyzs"][f, m:].any(), f"{msg} frame {f}: slots beyond the count must be zero-filled"
        if not m:
            continue
        assert_scores_close(out["xyzs"][f, :m, :, 3], ref["kscore"][f, :m], rtol=3e-7 if f32 else 1e-9, nterms=npairs, what=f"{msg} kscore frame {f}")
        assert_xyz_close(out["xyzs"][f, :m, :, :3], ref["xyz"][f, :m], 2e-6 if f32 else 1e-8, score_ref=ref["kscore"][f, :m],
                         what=f"{msg} xyz frame {f}")
        assert_scores_close(out["pscore"][f, :m], ref["pscore"][f, :m], rtol=3e-7 if f32 else 1e-9, nterms=kn * npairs, what=f"{msg} pscore frame {f}")


@pytest.mark.parametrize("C", [6, 7, 8])
@pytest.mark.parametrize("F", [3, 700, 20000])
def test_rolled_lean_kernels(api, C, F):
    """6-8 cameras on the production shape: k_fused_lean_coop (small launches) / k_fused_lean (large) on cluster_item; frames
    that break the speculation (a camera without detection, a camera far off, everything gated) take the in-launch fall-back."""
    from snowmocap_amd import synth, _lib
    from oracle import oracle as orc
    rng = np.random.default_rng(900 + 10 * C + F % 7)
    K, R, t = synth.ring_rig(C)
    gen = min(F, 160)
    X = synth.make_people(rng, gen, 1)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(2.0, 8.0))
    kp, npers = kp.copy(), npers.copy()
    if gen >= 40:
        npers[5, 1] = 0
        kp[11, 2, 0, :, :2] += 400.0
        kp[17, :, 0, :, 2] = 0.0
        kp[23, 3, 0, 20:40, :2] += 250.0
    reps = (F + gen - 1) // gen
    kpf, npf = np.tile(kp, (reps, 1, 1, 1, 1))[:F], np.tile(npers, (reps, 1))[:F]
    prm = dict(synth.default_thresholds(), condense_distance_tol=2.0)
    out = _run(api, K, R, t, prm, kpf, npf, np.float32)
    assert out["names"].startswith(f"k_fused_lean_coop<{C},float,133>" if F <= 16384 else f"k_fused_lean<{C},float,133>"), out["names"]
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 4)
    sel = np.arange(min(F, gen))
    sub = {k: (v[sel] if isinstance(v, np.ndarray) else v) for k, v in out.items()}
    fast = (sub["flags"] & _lib.FLAG_FASTPATH) != 0
    if gen >= 40:
        assert not fast[5] and not fast[11] and fast[23]
        assert fast.mean() > 0.9      # (opposite cameras of a ring see nearly parallel rays: a few frames take the fall-back)
    # the one output slot holds the oracle's first person (a broken frame may resolve to several)
    _compare(sub, ref, len(sel), np.float32, J, f"C={C} F={F}")
    # the tail of a tiled batch equals its head (bit for bit: the same frames)
    if F > gen:
        tail = slice((reps - 1) * gen, F)
        n = F - (reps - 1) * gen
        assert np.array_equal(out["xyzs"][tail], out["xyzs"][:n], equal_nan=True)
        assert np.array_equal(out["pscore"][tail], out["pscore"][:n], equal_nan=True)


@pytest.mark.parametrize("C", [5, 6, 7, 8])
@pytest.mark.parametrize("F,in_dtype", [(700, np.float32), (20000, np.float32), (300, np.float64)])
def test_lean_kernels_with_float64_outputs(api, C, F, in_dtype):
    """The reference's own output type on the production shape, five to eight cameras: the lean kernels on cluster_item's
    Newton-refined branch (k_fused_lean<C,TIn,133,double>), 32-byte joint records; float64 tolerances (1e-8 m, 1e-9 relative)."""
    from snowmocap_amd import synth, _lib
    from oracle import oracle as orc
    rng = np.random.default_rng(640 + C)
    K, R, t = synth.ring_rig(C)
    gen = min(F, 120)
    X = synth.make_people(rng, gen, 1)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(2.0, 8.0), dtype=in_dtype)
    kp, npers = kp.copy(), npers.copy()
    npers[5, 1] = 0
    kp[11, 2, 0, :, :2] += 400.0
    kp[17, :, 0, :, 2] = 0.0
    reps = (F + gen - 1) // gen
    kpf, npf = np.tile(kp, (reps, 1, 1, 1, 1))[:F], np.tile(npers, (reps, 1))[:F]
    prm = dict(synth.default_thresholds(), condense_distance_tol=2.0)
    out = _run(api, K, R, t, prm, kpf, npf, np.float64)
    tin = "float" if in_dtype == np.float32 else "double"
    assert out["names"].startswith(f"k_fused_lean_coop<{C},{tin},133,double>" if F <= 16384 else f"k_fused_lean<{C},{tin},133,double>"), out["names"]
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 4)
    sub = {k: (v[:gen] if isinstance(v, np.ndarray) else v) for k, v in out.items()}
    fast = (sub["flags"] & _lib.FLAG_FASTPATH) != 0
    assert not fast[5] and not fast[11] and fast.mean() > 0.9
    _compare(sub, ref, gen, np.float64, J, f"C={C} F={F}")
    if F > gen:
        n = F - (reps - 1) * gen
        assert np.array_equal(out["xyzs"][(reps - 1) * gen:], out["xyzs"][:n], equal_nan=True)
        assert np.array_equal(out["pscore"][(reps - 1) * gen:], out["pscore"][:n], equal_nan=True)


@pytest.mark.parametrize("C", [5, 6, 8, 12, 16])
@pytest.mark.parametrize("out_dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kn", [133, 30])
def test_sumless_route_against_oracle(api, C, out_dtype, kn):
    """Shapes the lean kernels do not take (float64 outputs, keypoint_num < J, two slots): no candidate pass, the mean scores
    from the fused joints; frames with a missing detection (member-list clusters), a far-off camera (two clusters) and
    gated confidences included."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    pout = 2 if C <= 8 else 1 + (C % 2)      # (up to 8 cameras one slot with keypoint_num = J would be the lean shape)
    rng = np.random.default_rng(77 * C + kn)
    K, R, t = synth.ring_rig(C)
    F = 60
    X = synth.make_people(rng, F, 1)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(2.0, 8.0))
    kp, npers = kp.copy(), npers.copy()
    npers[3, 1] = 0
    npers[9, C - 1] = 0
    kp[14, 2, 0, :, :2] += 400.0
    kp[20, :, 0, :, 2] = 0.0
    kp[26, 0, 0, 7, 0] = np.nan
    prm = dict(synth.default_thresholds(), condense_distance_tol=0.5, keypoint_num=kn, center_point_index=min(18, kn - 1))
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), pout)
    out = _run(api, K, R, t, prm, kp, npers, out_dtype, pout=pout)
    assert "k_candidate_sums" not in out["names"] and out["names"].startswith("k_singular_scan<") and "k_associate<" in out["names"], out["names"]
    assert ("k_cluster_fuse<%d," % C if C <= 8 else "k_cluster_fuse_wide<") in out["names"], out["names"]
    _compare(out, ref, F, out_dtype, kn, f"C={C} {np.dtype(out_dtype).name} kn={kn}")


@pytest.mark.parametrize("C", [5, 8, 12])
def test_sumless_equals_candidate_pass(api, C, knobs):
    """SNOWTRI_SUMLESS_MODE=0 keeps the candidate pass: the same counts, the same joints bit for bit (the same fusion
    kernels on the same descriptors), the persons' mean scores within the float32 contract (candidate sums vs fused joints)."""
    from snowmocap_amd import synth
    rng = np.random.default_rng(5 + C)
    K, R, t = synth.ring_rig(C)
    F = 300
    X = synth.make_people(rng, F, 1)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1.0, score_range=(2.0, 8.0))
    npers = npers.copy()
    npers[7, 0] = 0
    prm = dict(synth.default_thresholds(), condense_distance_tol=0.5)
    a = _run(api, K, R, t, prm, kp, npers, np.float32, pout=2)
    b = _run(api, K, R, t, prm, kp, npers, np.float32, pout=2, env={"SNOWTRI_SUMLESS_MODE": "0"}, knobs=knobs)
    assert "k_candidate_sums" in b["names"] and "k_candidate_sums" not in a["names"]
    assert np.array_equal(a["count"], b["count"])
    assert np.array_equal(a["xyzs"], b["xyzs"], equal_nan=True)
    np.testing.assert_allclose(a["pscore"], b["pscore"], rtol=3e-7)


def test_sumless_flags_a_singular_pair(api):
    """Two cameras with parallel rays at one joint: the reference's np.linalg.inv raises (triangulation.py:26); the fused
    entry reports SNOWTRI_FLAG_SINGULAR for that frame on the route without a candidate pass as well."""
    from snowmocap_amd import synth, _lib
    C = 6
    rng = np.random.default_rng(1)
    K = np.tile(np.eye(3), (C, 1, 1))                      # normalised image coordinates: pixel (0, 0) is the ray (0, 0, 1) exactly
    R = np.tile(np.eye(3), (C, 1, 1))                      # identical orientation: equal pixels = parallel rays
    t = rng.uniform(-2, 2, size=(C, 3))
    t[:, 2] = 0.0
    F = 8
    X = rng.uniform(-0.5, 0.5, size=(F, 1, J, 3)) + np.array([0, 0, 5.0])
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=1e-3, score_range=(3.5, 8.0))
    kp = kp.copy()
    kp[4, :2, 0, 40, :2] = 0.0                              # frame 4, joint 40: cameras 0 and 1 both look straight down +z
                                                            # (H^T H = [[1, 1], [1, 1]]: exactly singular, as tests/golden g4 "parallel_rays")
    prm = dict(synth.default_thresholds())
    bt = api.BatchTriangulator(K, R, t, prm, pout_max=2, out_dtype=np.float64)
    out = bt.run_host(kp, npers)
    names = bt.ctx.last_kernel_names()
    bt.close()
    assert names.startswith("k_singular_scan<") and "k_associate<" in names, names
    sing = (out["flags"] & _lib.FLAG_SINGULAR) != 0
    assert sing[4] and sing.sum() == 1
    assert out["status"] == _lib.ERR_SINGULAR


def _twin_camera_rig(C, rng):
    """A ring rig whose camera 1 is camera 0 moved sideways: the same K and R, another centre.  Equal pixels in the two are
    bit-identical, PARALLEL rays under a general K and R (not the K = R = I of the test above): a = b = c in H^T H, the pair is
    singular as the reference's LU sees it, np.linalg.inv raises (triangulation.py:26)."""
    from snowmocap_amd import synth
    K, R, t = synth.ring_rig(C, radius=4.5)
    K, R, t = K.copy(), R.copy(), t.copy()
    K[1], R[1] = K[0], R[0]
    t[1] = t[0] + R[0] @ np.array([0.4, 0.1, 0.0])
    return K, R, t


@pytest.mark.parametrize("C,out_dtype,kn,pout,joint,route", [
    (4, np.float32, J, 1, 40, "k_fused_lean"),            # lean_item (the bench's kernel)
    (4, np.float64, J, 1, 7, "k_fused_single<4,0"),       # pairwise_item: one reciprocal for the six determinants
    (4, np.float32, 30, 1, 100, "k_fused_single<4,0"),    # ... a singular pair at a joint BEHIND keypoint_num
    (8, np.float32, J, 1, 132, "k_fused_lean"),           # cluster_item inside the lean kernel
    (6, np.float64, J, 2, 3, "k_singular_scan"),          # the streaming route without its candidate pass, a fused joint
    (6, np.float32, 30, 1, 90, "k_singular_scan"),        # ... a joint behind keypoint_num: only the scan meets it
    (12, np.float32, J, 1, 50, "k_cluster_fuse_wide"),    # nine and more cameras: the LDS-resident item
    (12, np.float32, 20, 1, 77, "k_cluster_fuse_wide"),
])
def test_equal_rays_under_a_general_rig_are_flagged_on_every_single_detection_route(api, C, out_dtype, kn, pout, joint, route, knobs):
    """Round-5 advice: with a FUSED determinant fma(a, c, -(b b)) equal rays give the rounding error of b b, which is positive
    in ~18 % of the cases -- a finite, wrong joint and no flag.  Every fast item now forms det from separately rounded products
    (singular <=> det == 0 exactly -> the sum is not finite -> the exact routine flags the frame) or tests a c == b b itself, and
    the route without a candidate pass scans every listed pair at every joint.  64 frames with their own pixel each, so that
    several roundings of b b are met; the flag must be on every one of them and nowhere else, whatever route the call takes --
    and, where a knob forces another route (the test build), the same frames."""
    from snowmocap_amd import synth, _lib
    rng = np.random.default_rng(900 + C + kn)
    K, R, t = _twin_camera_rig(C, rng)
    F = 96
    X = synth.make_people(rng, F, 1)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.8, score_range=(3.5, 8.0))
    kp = kp.copy()
    hit = np.arange(F) % 3 != 1                            # two frames of three carry the singular pair
    px = rng.uniform(200, 1000, size=(F, 2)).astype(np.float32)
    kp[hit, 1, 0, joint, :2] = px[hit]
    kp[hit, 0, 0, joint, :2] = px[hit]                     # cameras 0 and 1: the same pixel -> the same ray, bit for bit
    prm = dict(synth.default_thresholds(), condense_distance_tol=10.0, keypoint_num=kn, center_point_index=0)
    out = _run(api, K, R, t, prm, kp, npers, out_dtype, pout=pout)
    assert route in out["names"], out["names"]
    sing = (out["flags"] & _lib.FLAG_SINGULAR) != 0
    assert np.array_equal(sing, hit), (np.nonzero(sing != hit)[0], out["names"])
    assert out["status"] == _lib.ERR_SINGULAR
    if route == "k_singular_scan":                         # the flag does not depend on the route: the candidate pass finds the same frames
        b = _run(api, K, R, t, prm, kp, npers, out_dtype, pout=pout, env={"SNOWTRI_SUMLESS_MODE": "0"}, knobs=knobs)
        assert "k_candidate_sums" in b["names"], b["names"]
        assert np.array_equal((b["flags"] & _lib.FLAG_SINGULAR) != 0, hit)


def test_sumless_flags_a_singular_pair_of_a_candidate_outside_every_kept_cluster(api, knobs):
    """... and a candidate no kept cluster holds: with a tight condense_distance_tol the candidate of the twin cameras (its 3D
    points are garbage) clusters with nobody, condense_person_num_tol = 2 drops it, nothing of it is ever fused -- the reference
    has raised long before (triangulation.py:26 runs in Human_Triangulation)."""
    from snowmocap_amd import synth, _lib
    rng = np.random.default_rng(77)
    C = 7
    K, R, t = _twin_camera_rig(C, rng)
    F = 24
    X = synth.make_people(rng, F, 1)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.5, score_range=(3.5, 8.0))
    kp = kp.copy()
    kp[5, 0, 0, 60, :2] = kp[5, 1, 0, 60, :2]
    prm = dict(synth.default_thresholds(), condense_distance_tol=0.05, condense_person_num_tol=2, keypoint_num=40, center_point_index=0)
    a = _run(api, K, R, t, prm, kp, npers, np.float32, pout=3)
    b = _run(api, K, R, t, prm, kp, npers, np.float32, pout=3, env={"SNOWTRI_SUMLESS_MODE": "0"}, knobs=knobs)
    assert "k_singular_scan" in a["names"] and "k_candidate_sums" in b["names"]
    for o in (a, b):
        sing = (o["flags"] & _lib.FLAG_SINGULAR) != 0
        assert sing[5] and sing.sum() == 1, np.nonzero(sing)[0]


@pytest.mark.parametrize("C", [5, 6, 8])
@pytest.mark.parametrize("in_dtype", [np.float32, np.float64])
def test_dlt_wide_rigs(api, C, in_dtype, knobs):
    """DLT on 5-8 cameras vs oracle/dlt.py on both kernels: k_dlt_coop (the Wholebody skeleton, one slot: workgroup tiles, buffer
    loads) and k_fused_single<C,1> (every other shape; forced here by SNOWTRI_LEAN_MODE=0, and taken with keypoint_num < J).  The
    item is the same function: the joints agree bit for bit."""
    from snowmocap_amd import synth, _lib
    from oracle import dlt as odlt
    rng = np.random.default_rng(60 + C)
    K, R, t = synth.ring_rig(C)
    F = 50
    X = synth.make_people(rng, F, 1)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.5, score_range=(2.0, 8.0), dtype=in_dtype)
    npers[3, 1] = 0                        # a camera that lists nobody in frame 3
    prm = dict(synth.default_thresholds())
    want, wps, wcnt = odlt.dlt_batch(K, R, t, kp * (npers[:, :, None, None, None] > 0), prm["keypoint_score_threshold"], prm["keypoint_num"])
    outs = {}
    for kernel in ("k_dlt_coop", "k_fused_single"):
        if kernel == "k_fused_single":
            knobs.set("SNOWTRI_LEAN_MODE", "0")
        bt = api.BatchTriangulator(K, R, t, prm, pout_max=1, out_dtype=np.float64, method=_lib.DLT)
        out = outs[kernel] = bt.run_host(kp, npers)
        names = bt.ctx.last_kernel_names()
        bt.close()
        knobs.clear()
        assert names.startswith(f"k_dlt_coop<{C}," if kernel == "k_dlt_coop" else f"k_fused_single<{C},1,"), names
        assert (out["count"] == 1).all()
        err = np.abs(out["xyzs"][..., :3] - want[..., :3]).max()
        assert err < 1e-9, err
        np.testing.assert_allclose(out["xyzs"][..., 3], want[..., 3], rtol=1e-6)
        np.testing.assert_allclose(out["pscore"], wps, rtol=1e-6)
    assert np.array_equal(outs["k_dlt_coop"]["xyzs"].view(np.uint8), outs["k_fused_single"]["xyzs"].view(np.uint8))
    # a joint's bits do not depend on its wave: frames whose cameras all count (the wave adds their rows without the mask product and sums
    # the confidences as they are) next to frames with gated cameras, cut at frame offsets that move them between the two kinds of waves
    kp2 = kp.copy()
    kp2[:25, :, :, :, 2] = np.maximum(kp2[:25, :, :, :, 2], 3.5)
    np2 = np.ones_like(npers)
    bt = api.BatchTriangulator(K, R, t, prm, pout_max=1, out_dtype=np.float64, method=_lib.DLT)
    whole = bt.run_host(kp2, np2)
    for lo in (1, 7, 24, 26):
        part = bt.run_host(np.ascontiguousarray(kp2[lo:]), np.ascontiguousarray(np2[lo:]))
        assert np.array_equal(part["xyzs"].view(np.uint8), whole["xyzs"][lo:].view(np.uint8)), lo
        assert np.array_equal(part["pscore"].view(np.uint8), whole["pscore"][lo:].view(np.uint8)), lo
    bt.close()
    # keypoint_num < J: not the Wholebody shape -> k_fused_single
    bt = api.BatchTriangulator(K, R, t, dict(prm, keypoint_num=100), pout_max=1, out_dtype=np.float64, method=_lib.DLT)
    o2 = bt.run_host(kp, npers)
    assert bt.ctx.last_kernel_names().startswith(f"k_fused_single<{C},1,")
    bt.close()
    assert np.array_equal(o2["xyzs"][:, :, :100].view(np.uint8), outs["k_dlt_coop"]["xyzs"][:, :, :100].view(np.uint8))


def test_random_single_person_rigs_on_every_route(api):
    """Randomised sweep over what decides the route of a one-detection-per-camera batch: 5-16 cameras, output type, one or
    more slots, keypoint_num = J or less, thresholds that keep / break the speculation (average_score_threshold > 0,
    condense_score_tol > 0, tight condense_distance_tol), missing detections, gated and NaN keypoints -- lean kernels (both
    output types), the streaming route with and without its candidate pass, k_frame_recompute: all against the oracle."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(7117)
    routes = {}
    for trial in range(36):
        C = int(rng.choice([5, 6, 7, 8, 9, 12, 16]))
        F = int(rng.choice([3, 40, 150]))
        out_dtype = np.float64 if rng.uniform() < 0.5 else np.float32
        kn = J if rng.uniform() < 0.6 else int(rng.integers(1, J))
        pout = int(rng.choice([1, 1, 2, 3]))
        K, R, t = synth.ring_rig(C, radius=float(rng.uniform(3.5, 6)))
        X = synth.make_people(rng, F, 1)
        kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=float(rng.choice([0.3, 1.0, 3.0])), score_range=(2.0, 8.0),
                                         dtype=np.float64 if trial % 5 == 0 else np.float32)
        kp, npers = kp.copy(), npers.copy()
        for _ in range(int(rng.integers(0, 3))):
            npers[rng.integers(0, F), rng.integers(0, C)] = 0
        if rng.uniform() < 0.3:
            kp[rng.integers(0, F), rng.integers(0, C), 0, rng.integers(0, J), 0] = np.nan
        if rng.uniform() < 0.3:
            kp[rng.integers(0, F), :, 0, :, 2] = 0.0
        prm = dict(keypoint_score_threshold=float(rng.choice([0.0, 3.0, 5.0])), average_score_threshold=float(rng.choice([0.0, 0.0, 0.5])),
                   distance_threshold=float(rng.choice([0.02, 0.05, 1.0])), condense_distance_tol=float(rng.choice([0.05, 0.5, 10.0])),
                   condense_person_num_tol=int(rng.choice([0, 2, C * (C - 1) // 2])), condense_score_tol=float(rng.choice([0.0, 0.0, 0.6])),
                   center_point_index=int(rng.integers(0, kn)), keypoint_num=kn)
        ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), max(pout, 4))
        out = _run(api, K, R, t, prm, kp, npers, out_dtype, pout=pout)
        route = out["names"].split("<")[0] + ("+sums" if "k_candidate_sums" in out["names"] else "")
        routes[route] = routes.get(route, 0) + 1
        ok = ref["status"] == 0
        sub = {k: (v[ok] if isinstance(v, np.ndarray) and v.shape[:1] == (F,) else v) for k, v in out.items()}
        refs = {k: (v[ok] if isinstance(v, np.ndarray) and v.shape[:1] == (F,) else v) for k, v in ref.items()}
        _compare(sub, refs, int(ok.sum()), out_dtype, kn, f"trial {trial}: C={C} F={F} {np.dtype(out_dtype).name} pout={pout} {prm} [{out['names'][:60]}]",
                 npairs=C * (C - 1) // 2)
    assert len(routes) >= 3, routes

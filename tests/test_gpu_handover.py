"""Hand-over of the clusters from k_frame_recompute to k_cluster_fuse (snowtri_cluster.hpp): the
multi-person path with float32 outputs and keypoint_num == J, against the CPU oracle and against the same launch with
the hand-over switched off (SNOWTRI_HANDOVER_MODE=0: phase 3 stays inside k_frame_recompute), through the C ABI.

What the cases aim at: clusters that are complete graphs (k_cluster_fuse), persons missing in a camera or ragged person
lists (clusters of another shape: its member-list passes, in the SAME launch), persons dropped by the mean-score filter
(slots move up), more persons than Pout_max (overflow flag, only the first Pout_max written), every camera count the
cluster kernel is instantiated for, float64 inputs, exact intersections with a gated confidence (the joints the fast
item cannot finish), determinism.

Tolerances (tests/test_gpu_parity.py): float32 outputs <= 2e-6 m, scores <= 3e-7 relative.
"""
import numpy as np
import pytest

from conftest import assert_scores_close, assert_xyz_close

pytestmark = pytest.mark.gpu

XYZ_F32 = 2e-6
PRM = dict(keypoint_score_threshold=3.0, average_score_threshold=0.3, distance_threshold=0.05, condense_distance_tol=0.3,
           condense_person_num_tol=2, condense_score_tol=0.0, center_point_index=0)


@pytest.fixture(scope="module")
def api():
    import snowmocap_amd as sm
    from snowmocap_amd import _lib
    assert _lib.lib().snowtri_device_count() > 0, "these tests need the HIP device"
    return sm


def _run(api, K, R, t, prm, kp, npers, pout, knobs, handover=True, mode=None, out_dtype=np.float32):
    """mode 1 (default): the streaming association (k_candidate_sums -> k_associate -> cluster kernels, <= 16 cameras);
    mode 2: descriptors written by k_frame_recompute itself (<= 8 cameras); mode 0: everything inside k_frame_recompute."""
    m = int(mode) if mode is not None else (1 if handover else 0)
    if m != 1:      # (a forced route: the test build of the library, conftest.Knobs; mode 1 is the product's default)
        knobs.set("SNOWTRI_HANDOVER_MODE", m)
    bt = api.BatchTriangulator(K, R, t, prm, pout_max=pout, out_dtype=out_dtype)
    out = bt.run_host(kp, npers)      # (overflow / singular come back as out["status"], not as exceptions)
    out["handed"] = bt.ctx.last_handover_persons()
    out["kernels"] = bt.ctx.last_kernel_names()
    out["stream_counts"] = bt.ctx.last_stream_counts()
    bt.close()
    if m != 1:
        knobs.clear("SNOWTRI_HANDOVER_MODE")
    return out


def _check(out, ref, pout, J, msg, xyz_tol=XYZ_F32):
    np.testing.assert_array_equal(out["count"], ref["count"], err_msg=msg)
    for f in range(len(ref["count"])):
        m = min(int(ref["count"][f]), pout)
        assert not out["xyzs"][f, m:].any(), f"{msg} frame {f}: unused slots must be zero"
        if m:
            assert_scores_close(out["xyzs"][f, :m, :, 3], ref["kscore"][f, :m], rtol=3e-7, what=f"{msg} kscore frame {f}")
            assert_xyz_close(out["xyzs"][f, :m, :, :3], ref["xyz"][f, :m], xyz_tol, score_ref=ref["kscore"][f, :m],
                             what=f"{msg} xyz frame {f}")
            assert_scores_close(out["pscore"][f, :m], ref["pscore"][f, :m], rtol=3e-7, nterms=J, what=f"{msg} pscore frame {f}")


def _same(a, b, msg, xyz_tol=XYZ_F32):
    """hand-over on vs off: same counts, same NaN pattern, values within the float32 tolerances of both."""
    np.testing.assert_array_equal(a["count"], b["count"], err_msg=msg)
    xa, xb = a["xyzs"].astype(np.float64), b["xyzs"].astype(np.float64)
    assert np.array_equal(np.isnan(xa), np.isnan(xb)), msg
    fin = np.isfinite(xa) & np.isfinite(xb)
    ulp32 = 2.0 ** -23 if a["xyzs"].dtype == np.float32 else 0.0     # (far-out joints: one ulp of the storage type each)
    assert (np.abs(xa[..., :3] - xb[..., :3]) - 2 * ulp32 * np.abs(xb[..., :3]))[fin[..., :3]].max(initial=0.0) < 2 * xyz_tol, msg
    sa, sb = xa[..., 3][fin[..., 3]], xb[..., 3][fin[..., 3]]
    assert np.all(np.abs(sa - sb) <= 6e-7 * np.abs(sb)), msg


@pytest.mark.parametrize("C,P,in_dtype", [(8, 4, np.float32), (4, 3, np.float64), (2, 2, np.float32), (3, 4, np.float32),
                                          (5, 2, np.float32), (6, 3, np.float64), (7, 2, np.float32), (8, 2, np.float64)])
def test_complete_clusters_are_handed_over_and_match_the_oracle(api, C, P, in_dtype, knobs):
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(100 * C + P)
    F, J = 12, 133 if C in (8, 4) else 40
    K, R, t = synth.ring_rig(C, radius=4.5)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=in_dtype)
    prm = dict(PRM, keypoint_num=J, condense_person_num_tol=1 if C == 2 else 2)
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    pout = P + 1
    out = _run(api, K, R, t, prm, kp, npers, pout, knobs)
    off = _run(api, K, R, t, prm, kp, npers, pout, knobs, handover=False)
    msg = f"C={C} P={P}"
    _check(out, ref, pout, J, msg)
    _check(off, ref, pout, J, msg + " (hand-over off)")
    _same(out, off, msg)
    assert off["handed"] == (-1, -1)
    # the second hand-over route (descriptors written by k_frame_recompute): same persons handed over, same results
    in_kernel = _run(api, K, R, t, prm, kp, npers, pout, knobs, mode=2)
    _check(in_kernel, ref, pout, J, msg + " (hand-over from k_frame_recompute)")
    _same(in_kernel, off, msg + " (hand-over from k_frame_recompute)")
    assert sum(in_kernel["handed"]) == sum(out["handed"]), (in_kernel["handed"], out["handed"])
    # clean synthetic people, everyone seen by every camera: (nearly) every output person is a complete-graph cluster; a
    # ghost candidate that joins a cluster (or forms its own) makes a cluster of another shape
    assert sum(out["handed"]) == int(np.minimum(ref["count"], pout).sum()), (out["handed"], ref["count"])
    assert out["handed"][0] >= 0.6 * sum(out["handed"]) > 0, out["handed"]


@pytest.mark.parametrize("C,P,J,in_dtype", [(9, 3, 133, np.float32), (12, 2, 133, np.float64), (16, 3, 133, np.float32),
                                            (16, 8, 133, np.float32), (13, 2, 40, np.float32), (10, 4, 20, np.float32)])
def test_wide_rigs_hand_their_clusters_to_the_lds_resident_kernel(api, C, P, J, in_dtype, knobs):
    """9-16 cameras (BASELINE configs[4] is 16 x 8): the streaming association hands complete-graph clusters to
    k_cluster_fuse_wide (rays in LDS, four lanes per (person, joint)) and every other cluster to k_cluster_members.
    Against the oracle, against the same launch with phase 3 inside k_frame_recompute, with the routes read back.
    Some frames get a person one camera missed and a ragged list, so both kernels have work in the same launch."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(1600 + 10 * C + P)
    F = 3 if P == 8 else 6
    K, R, t = synth.ring_rig(C, radius=5.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=in_dtype)
    npers = npers.copy()
    kp[1, 2, 0, :, 2] = 0.0            # frame 1: one detection of camera 2 has no confidence -> a cluster with fewer members
    npers[2, C - 1] = P - 1            # frame 2: the last camera lists one person less
    prm = dict(PRM, keypoint_num=J, condense_person_num_tol=10)
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    pout = P + 2
    out = _run(api, K, R, t, prm, kp, npers, pout, knobs)
    off = _run(api, K, R, t, prm, kp, npers, pout, knobs, handover=False)
    msg = f"C={C} P={P} J={J}"
    _check(out, ref, pout, J, msg)
    _check(off, ref, pout, J, msg + " (hand-over off)")
    _same(out, off, msg)
    assert off["handed"] == (-1, -1)
    n_complete, n_other = out["handed"]
    assert n_complete + n_other == int(np.minimum(ref["count"], pout).sum()), (out["handed"], ref["count"])
    # (a ghost candidate that joins a person's cluster makes it a member-list cluster: with 120 camera pairs that is common)
    assert n_complete >= P and n_other >= 2, out["handed"]


def test_random_wide_rigs_against_oracle_and_phase3(api, knobs):
    """Randomised sweep over 9..16 cameras: 1..3 detections per camera with ragged lists, thresholds that switch the
    filters on and off, float32 / float64 keypoints, Pout_max below and above the person count."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(91600)
    routes = np.zeros(2, np.int64)
    for trial in range(14):
        C = int(rng.integers(9, 17))
        P = int(rng.integers(1, 4))
        J = int(rng.choice([5, 33, 40, 133]))
        F = int(rng.integers(1, 5))
        K, R, t = synth.ring_rig(C, radius=float(rng.uniform(4, 6)))
        X = synth.make_people(rng, F, P, J=J)
        kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=float(rng.choice([0.3, 1.0, 3.0])), score_range=(2.0, 8.0),
                                         permute_persons=True, dtype=np.float64 if trial % 3 == 0 else np.float32)
        npers = npers.copy()
        for _ in range(int(rng.integers(0, 3))):
            npers[rng.integers(0, F), rng.integers(0, C)] = rng.integers(0, P + 1)
        prm = dict(keypoint_score_threshold=float(rng.choice([0.0, 3.0, 3.0, 5.0])),
                   average_score_threshold=float(rng.choice([0.0, 0.3, 1.5])),
                   distance_threshold=float(rng.choice([0.02, 0.05, 1.0])),
                   condense_distance_tol=float(rng.choice([0.05, 0.3, 10.0])),
                   condense_person_num_tol=int(rng.choice([0, 2, 10])),
                   condense_score_tol=float(rng.choice([0.0, 0.0, 0.3, 2.0])),
                   center_point_index=int(rng.integers(0, J)), keypoint_num=J)
        pout = int(rng.choice([1, 4, 16]))
        ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
        out = _run(api, K, R, t, prm, kp, npers, pout, knobs)
        off = _run(api, K, R, t, prm, kp, npers, pout, knobs, handover=False)
        msg = f"trial {trial}: C={C} P={P} J={J} F={F} {prm} pout={pout} n={npers.tolist()}"
        # float32 outputs take 1/dist from the raw v_rsq_f64 (5e-8 relative): the fused point sees it through the weights
        # s_q / sum s, i.e. times the SPREAD of the member points.  With the distance gate practically off (1 m) a ring of
        # 9-16 cameras fuses pairs of nearly opposite cameras whose ill-conditioned points lie tens of metres apart:
        # 5e-8 x 50 m on top of the float32 rounding (soak: 3.2e-6 m).  The required bar is 1e-4 m.
        tol = XYZ_F32 if prm["distance_threshold"] < 1.0 else 8e-6
        _check(out, ref, pout, J, msg, xyz_tol=tol)
        _same(out, off, msg, xyz_tol=tol)
        assert out["handed"][0] >= 0 and sum(out["handed"]) <= int(np.minimum(ref["count"], pout).sum()), (msg, out["handed"])
        routes += out["handed"]
    # (with 36-120 camera pairs a ghost candidate joins most clusters: complete graphs are the minority in random scenes)
    assert routes[0] > 0 and routes[1] > 5, routes


def test_mixed_launch_incomplete_and_ragged_frames_stay_in_phase3(api, knobs):
    """One launch, three kinds of frames: complete clusters (k_cluster_fuse), a person whose detection in one camera is
    below the keypoint threshold everywhere or missing from the list (its cluster has fewer members: a member-list descriptor),
    empty frames."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(5)
    C, P, F, J = 6, 3, 40, 133
    K, R, t = synth.ring_rig(C, radius=5.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
    npers = npers.copy()
    kinds = {}
    for f in range(F):
        k = f % 5
        if k == 1:      # one camera lists one person less
            npers[f, rng.integers(0, C)] = P - 1
            kinds[f] = "ragged"
        elif k == 2:    # one detection has no confidence at all: its candidates are dropped by the mean-score gate
            kp[f, rng.integers(0, C), rng.integers(0, P), :, 2] = 0.0
            kinds[f] = "occluded"
        elif k == 3 and f % 10 == 3:
            npers[f] = 0
            kinds[f] = "empty"
    prm = dict(PRM, keypoint_num=J)
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    out = _run(api, K, R, t, prm, kp, npers, P, knobs)
    off = _run(api, K, R, t, prm, kp, npers, P, knobs, handover=False)
    _check(out, ref, P, J, "mixed launch")
    _same(out, off, "mixed launch")
    full = int(np.minimum(ref["count"], P).sum())
    n_complete, n_other = out["handed"]
    assert n_complete + n_other == full, (out["handed"], full)
    # every "ragged" / "occluded" frame has a person seen by one camera less: C(C-1, 2) members instead of C(C, 2)
    assert n_other >= sum(1 for k in kinds.values() if k != "empty") and n_complete >= 2 * n_other, out["handed"]


@pytest.mark.parametrize("score_tol,pout", [(0.0, 2), (1.2, 4), (1.2, 1), (50.0, 4)])
def test_mean_score_filter_and_overflow_with_handover(api, score_tol, pout, knobs):
    """condense_score_tol in the middle of the persons' mean scores: some clusters are dropped and the later ones move up
    (the slots are decided by the association kernel from the candidate means); pout < persons: overflow flag, only
    the first pout persons written."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(17)
    C, P, F, J = 4, 4, 24, 133
    K, R, t = synth.ring_rig(C, radius=4.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.5, score_range=(3.5, 9.0), permute_persons=False, dtype=np.float32)
    prm = dict(PRM, keypoint_num=J, condense_score_tol=0.0)
    ref0 = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    assert (ref0["count"] >= P).all(), ref0["count"]
    # a tolerance in the middle of the persons' mean scores (they are ~1/dist: they differ a lot from person to person)
    ps = np.concatenate([ref0["pscore"][f, :ref0["count"][f]] for f in range(F)])
    tol = {0.0: 0.0, 1.2: float(np.median(ps)), 50.0: float(ps.max() * 2)}[score_tol]
    prm["condense_score_tol"] = tol
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    if score_tol == 1.2:    # the filter really fires, and not for the last persons only: later ones move up
        assert 0.3 * len(ps) < ref["count"].sum() < 0.7 * len(ps), (ref["count"], ref0["count"])
        assert any((ref0["pscore"][f, 0] < tol) and ref["count"][f] > 0 for f in range(F))
    out = _run(api, K, R, t, prm, kp, npers, pout, knobs)
    off = _run(api, K, R, t, prm, kp, npers, pout, knobs, handover=False)
    msg = f"score_tol={score_tol} pout={pout}"
    _check(out, ref, pout, J, msg)
    _same(out, off, msg)
    np.testing.assert_array_equal(out["flags"] & 2, np.where(ref["count"] > pout, 2, 0), err_msg=msg)
    # (the median tolerance IS one person's mean score: that frame is "undecided" on the fast sums and stays in phase 3)
    total = int(np.minimum(ref["count"], pout).sum())
    assert total - (P if score_tol == 1.2 else 0) <= sum(out["handed"]) <= total, (out["handed"], ref["count"])


def test_exact_intersections_and_gated_confidences_in_handed_over_clusters(api, knobs):
    """dist == 0 gives an inf pair score (triangulation.py:72); a confidence below the threshold ASSIGNS 0 to that pair
    (:73-74).  The fast item multiplies (0 * inf = NaN) and must re-do such joints member by member.  Exactly
    representable geometry (K = R = I) so that rays really intersect."""
    from oracle import oracle as orc
    C, P, J = 3, 2, 6
    K = np.tile(np.eye(3), (C, 1, 1)); R = np.tile(np.eye(3), (C, 1, 1))
    t = np.array([[0.0, 0, 0], [2.0, 0, 0], [0, 2.0, 0]])
    X = np.array([[[1.0, 0.0, 4.0], [0.5, 0.5, 2.0], [1.0, 1.0, 4.0], [0.0, 1.0, 2.0], [0.25, 0.75, 2.0], [1.5, 0.5, 4.0]],
                  [[3.0, 2.0, 8.0], [2.5, 2.5, 4.0], [3.0, 3.0, 8.0], [2.0, 3.0, 4.0], [2.25, 2.75, 4.0], [3.5, 2.5, 8.0]]])
    rng = np.random.default_rng(3)
    F = 6
    kp = np.zeros((F, C, P, J, 3), np.float32)
    for f in range(F):
        for c in range(C):
            for p in range(P):
                kp[f, c, p, :, 0] = (X[p, :, 0] - t[c, 0]) / X[p, :, 2]
                kp[f, c, p, :, 1] = (X[p, :, 1] - t[c, 1]) / X[p, :, 2]
                kp[f, c, p, :, 2] = 5.0
        # inexact joints (a real distance) mixed with the exact ones, and some gated confidences on exact joints
        kp[f, :, :, 4:, :2] += rng.normal(0, 1e-3, (C, P, J - 4, 2)).astype(np.float32)
        kp[f, rng.integers(0, C), 0, 1, 2] = 1.0
        kp[f, rng.integers(0, C), 1, 2, 2] = 1.0
    npers = np.full((F, C), P, np.int32)
    prm = dict(keypoint_score_threshold=3.0, average_score_threshold=0.0, distance_threshold=0.05, condense_distance_tol=0.5,
               condense_person_num_tol=0, condense_score_tol=0.0, center_point_index=4, keypoint_num=J)
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    out = _run(api, K, R, t, prm, kp, npers, 4, knobs)
    off = _run(api, K, R, t, prm, kp, npers, 4, knobs, handover=False)
    np.testing.assert_array_equal(out["count"], ref["count"])
    np.testing.assert_array_equal(off["count"], ref["count"])
    for name, o in (("hand-over", out), ("phase 3", off)):
        for f in range(F):
            m = min(int(ref["count"][f]), 4)
            got, want = o["xyzs"][f, :m].astype(np.float64), np.concatenate([ref["xyz"][f, :m], ref["kscore"][f, :m][..., None]], -1)
            bad = np.argwhere((np.isnan(got) != np.isnan(want)) | (np.isinf(got) != np.isinf(want)))
            assert len(bad) == 0, (name, f, bad[:4].tolist(), [(got[tuple(b[:2])].tolist(), want[tuple(b[:2])].tolist()) for b in bad[:3]])
            fin = np.isfinite(want)
            assert np.abs(got[fin] - want[fin]).max() < 1e-5 * max(1.0, np.abs(want[fin]).max())


def test_handover_is_deterministic_and_independent_of_the_batch_split(api, knobs):
    """The descriptor list is filled in whatever order the workgroups finish their frames; the outputs must not
    depend on it, nor on how the frames are split over launches."""
    from snowmocap_amd import synth
    rng = np.random.default_rng(23)
    C, P, F, J = 8, 4, 300, 133
    K, R, t = synth.ring_rig(C, radius=5.0)
    X = synth.make_people(rng, 30, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
    kp = np.tile(kp, (F // 30, 1, 1, 1, 1)); npers = np.tile(npers, (F // 30, 1))
    prm = dict(PRM, keypoint_num=J)
    a = _run(api, K, R, t, prm, kp, npers, P, knobs)
    b = _run(api, K, R, t, prm, kp, npers, P, knobs)
    assert a["handed"] == b["handed"] and sum(a["handed"]) == int(np.minimum(a["count"], P).sum())
    for key in ("xyzs", "pscore", "count"):
        assert np.array_equal(a[key], b[key]), key
    # the frames repeat with period 30: so must the outputs, wherever a frame sits in the launch
    assert np.array_equal(a["xyzs"][:30], a["xyzs"][270:])
    parts = [_run(api, K, R, t, prm, kp[s:e], npers[s:e], P, knobs) for s, e in ((0, 7), (7, 130), (130, 300))]
    assert np.array_equal(np.concatenate([p["xyzs"] for p in parts]), a["xyzs"])
    assert np.array_equal(np.concatenate([p["pscore"] for p in parts]), a["pscore"])


def test_random_rigs_with_handover_against_oracle_and_phase3(api, knobs):
    """Randomised sweep of what the hand-over can meet: 2..8 cameras, 2..4 detections per camera with ragged (also empty)
    person lists, 5..40 joints, thresholds that switch every filter on and off, ghost candidates that form clusters of
    their own, persons merged by a wide condense_distance_tol, Pout_max below and above the person count -- float32
    outputs, keypoint_num == J.  Against the oracle, and against the same launch with phase 3 kept in the kernel."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(4242)
    routes = np.zeros(2, np.int64)
    checked = 0
    for trial in range(40):
        C = int(rng.integers(2, 9))
        P = int(rng.integers(2, 5))
        J = int(rng.choice([5, 20, 33, 40]))
        F = int(rng.integers(1, 7))
        K, R, t = synth.ring_rig(C, radius=float(rng.uniform(3, 6)))
        X = synth.make_people(rng, F, P, J=J)
        kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=float(rng.choice([0.3, 1.0, 3.0])), score_range=(2.0, 8.0),
                                         permute_persons=True, dtype=np.float64 if trial % 3 == 0 else np.float32)
        npers = npers.copy()
        for _ in range(int(rng.integers(0, 3))):                          # ragged / empty person lists
            npers[rng.integers(0, F), rng.integers(0, C)] = rng.integers(0, P + 1)
        prm = dict(keypoint_score_threshold=float(rng.choice([0.0, 3.0, 3.0, 5.0])),
                   average_score_threshold=float(rng.choice([0.0, 0.0, 0.3, 1.5])),
                   distance_threshold=float(rng.choice([0.02, 0.05, 1.0])),
                   condense_distance_tol=float(rng.choice([0.05, 0.3, 10.0])),
                   condense_person_num_tol=int(rng.choice([0, 0, 1, 2])),
                   condense_score_tol=float(rng.choice([0.0, 0.0, 0.3, 2.0])),
                   center_point_index=int(rng.integers(0, J)), keypoint_num=J)
        pout = int(rng.choice([1, 4, 16]))
        ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
        out = _run(api, K, R, t, prm, kp, npers, pout, knobs)
        off = _run(api, K, R, t, prm, kp, npers, pout, knobs, handover=False)
        msg = f"trial {trial}: C={C} P={P} J={J} F={F} {prm} pout={pout} n={npers.tolist()}"
        _check(out, ref, pout, J, msg)
        _same(out, off, msg)
        assert out["handed"][0] >= 0 and sum(out["handed"]) <= int(np.minimum(ref["count"], pout).sum()), (msg, out["handed"])
        routes += out["handed"]
        checked += int(np.minimum(ref["count"], pout).sum())
    # both routes of the streaming kernel are exercised, and most persons take one of them
    assert routes[0] > 20 and routes[1] > 20 and routes.sum() > 0.5 * checked, (routes, checked)


def test_long_batches_are_cut_into_segments(api, knobs):
    """The descriptor and member lists are sized per segment of a long batch (<= 2 M persons, 32 M member words); the
    test knob SNOWTRI_HANDOVER_SEG_FRAMES makes the segments 7 frames short: same outputs, bit for bit."""
    from snowmocap_amd import synth
    rng = np.random.default_rng(8)
    C, P, F, J = 5, 3, 45, 40
    K, R, t = synth.ring_rig(C, radius=4.5)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
    npers = npers.copy()
    npers[::6, 1] = P - 1
    prm = dict(PRM, keypoint_num=J)
    whole = _run(api, K, R, t, prm, kp, npers, P, knobs)
    knobs.set("SNOWTRI_HANDOVER_SEG_FRAMES", "7")
    cut = _run(api, K, R, t, prm, kp, npers, P, knobs)
    knobs.clear("SNOWTRI_HANDOVER_SEG_FRAMES")
    for key in ("xyzs", "pscore", "count", "flags"):
        assert np.array_equal(whole[key], cut[key]), key
    assert sum(whole["handed"]) == int(np.minimum(whole["count"], P).sum())
    # (the counters read back belong to the LAST segment: frames 42..44)
    assert sum(cut["handed"]) == int(np.minimum(whole["count"][42:], P).sum())


@pytest.mark.parametrize("chunks", [1, 5])
def test_sharded_multi_person_batch_with_handover_single_rank_group(api, chunks):
    """The device-pointer route: ShardedTriangulator.run (1-rank RCCL group) cuts a multi-person shard into pieces whose
    inputs / outputs are slices of larger device buffers, each piece = association kernel + cluster kernel on the
    current stream, gathered on a side stream -- bit-identical to one unsharded launch, and equal to the host route."""
    import os
    import torch
    import torch.distributed as dist
    from snowmocap_amd import synth
    from snowmocap_amd.sharded import ShardedTriangulator
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29450 + os.getpid() % 200))
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        rng = np.random.default_rng(61)
        C, P, F, J = 6, 3, 203, 133
        K, R, t = synth.ring_rig(C, radius=5.0)
        X = synth.make_people(rng, 29, P, J=J)
        kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
        kp = np.tile(kp, (7, 1, 1, 1, 1)); npers = np.tile(npers, (7, 1)).copy()
        npers[::9, 2] = P - 1                   # some ragged frames: member-list descriptors
        prm = dict(PRM, keypoint_num=J)
        dev = torch.device("cuda", 0)
        kpd, npd = torch.from_numpy(kp).to(dev), torch.from_numpy(npers).to(dev)
        st = ShardedTriangulator(K, R, t, prm, pout_max=P + 1, device=0, chunks=chunks)
        got = st.run(kpd, F, npd)
        torch.cuda.synchronize()
        ref = st.bt.run_torch(kpd, npd)
        torch.cuda.synchronize()
        handed = st.bt.ctx.last_handover_persons()
        host = st.bt.run_host(kp, npers)
        for k in ("xyzs", "pscore", "count", "flags"):
            assert got[k].shape == ref[k].shape and torch.equal(got[k], ref[k]), k
            assert np.array_equal(ref[k].cpu().numpy().astype(host[k].dtype), host[k], equal_nan=True), k
        assert sum(handed) == int(np.minimum(host["count"], P + 1).sum()) and handed[1] > 0, handed
        st.bt.close()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("C,P,armed", [(4, 16, True), (8, 16, True), (3, 17, False)])
def test_person_index_limits_of_the_descriptor(api, C, P, armed, knobs):
    """The descriptor packs one 4-bit person index per camera: 16 detections per camera is the last shape that is handed
    over (person 15 must survive the packing), 17 keeps phase 3 in the association kernel."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(1000 + C * P)
    F, J = 2, 8
    K, R, t = synth.ring_rig(C, radius=9.0)
    X = synth.make_people(rng, F, P, J=J)
    X[..., 0] += np.linspace(-6, 6, P)[None, :, None]       # spread the crowd out: most persons form their own cluster
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.3, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
    prm = dict(PRM, keypoint_num=J, condense_person_num_tol=1)
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 128)
    out = _run(api, K, R, t, prm, kp, npers, 128, knobs)
    off = _run(api, K, R, t, prm, kp, npers, 128, knobs, handover=False)
    msg = f"C={C} P={P}"
    _check(out, ref, 128, J, msg)
    _same(out, off, msg)
    if armed:
        assert sum(out["handed"]) == int(ref["count"].sum()) and out["handed"][0] >= P, (out["handed"], ref["count"])
    else:
        assert out["handed"] == (-1, -1)


def test_frames_whose_member_lists_do_not_fit_the_staging_stay_in_phase3(api, knobs):
    """Every filter off and a huge condense_distance_tol: all 7 168 candidates of an 8 x 16 frame are kept and fall into
    one cluster.  Its member list does not fit the LDS staging of the hand-over: the frame keeps phase 3 (nothing handed
    over although the hand-over is armed), results as the oracle's."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(77)
    C, P, F, J = 8, 16, 2, 5
    K, R, t = synth.ring_rig(C, radius=9.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.3, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
    prm = dict(keypoint_score_threshold=0.0, average_score_threshold=0.0, distance_threshold=1e9, condense_distance_tol=1e9,
               condense_person_num_tol=0, condense_score_tol=0.0, center_point_index=0, keypoint_num=J)
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 4)
    assert (ref["count"] == 1).all()
    out = _run(api, K, R, t, prm, kp, npers, 4, knobs)
    _check(out, ref, 4, J, "one cluster of 7 168 members")
    assert out["handed"] == (0, 0), out["handed"]


@pytest.mark.parametrize("C,P,J,forced", [
    (16, 8, 133, {"SNOWTRI_SUMS_THREADS": "512"}),                                   # 960 tiles on 8 waves: two rounds through csum per chunk
    (16, 8, 133, {"SNOWTRI_SUMS_THREADS": "256", "SNOWTRI_SUMS_LDS_KB": "64"}),      # four rounds, three joints per buffer
    (8, 4, 133, {"SNOWTRI_SUMS_LDS_KB": "24"}),                                      # one wave of tiles, four joint sub-ranges, 8-joint chunks
    (8, 4, 40, {"SNOWTRI_SUMS_THREADS": "1024", "SNOWTRI_SUMS_LDS_KB": "160"}),      # the whole frame in one chunk: no second buffer used
    (6, 3, 33, {"SNOWTRI_SUMS_THREADS": "512"}),                                     # odd person count: one candidate per lane
    (8, 4, 133, {"SNOWTRI_SPLIT_SEGMENTS": "1"}),                                    # the whole call on the caller's stream
])
def test_candidate_sums_launch_shapes_against_oracle(api, C, P, J, forced, knobs):
    """k_candidate_sums picks its workgroup shape from the rig (256 threads x 3 per CU ... 1024 x 1), keeps a tile's sums in
    registers when one pass of the workgroup covers the tiles and walks them in rounds otherwise, and pipelines the joint
    chunks through two LDS buffers.  The development knobs force the shapes the BASELINE rigs do not take by themselves:
    every one of them against the oracle and against the default shape (same decisions, sums within the fast arithmetic)."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(4200 + 10 * C + P + J)
    F = 5 if C == 16 else 30
    K, R, t = synth.ring_rig(C, radius=5.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
    npers = npers.copy()
    npers[1, C - 1] = P - 1            # a ragged frame between full ones (one candidate per lane, every slot defined)
    npers[F - 1, 0] = max(P - 2, 0)
    prm = dict(PRM, keypoint_num=J, condense_person_num_tol=10 if C == 16 else 2)
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    pout = P + 2
    base = _run(api, K, R, t, prm, kp, npers, pout, knobs)
    for k, v in forced.items():
        knobs.set(k, v)
    out = _run(api, K, R, t, prm, kp, npers, pout, knobs)
    knobs.clear(*forced)
    msg = f"C={C} P={P} J={J} {forced}"
    _check(out, ref, pout, J, msg)
    _same(out, base, msg)
    assert sum(out["handed"]) == sum(base["handed"]) == int(np.minimum(ref["count"], pout).sum()), (out["handed"], base["handed"])


@pytest.mark.parametrize("C,P,segments", [(8, 4, 2), (8, 4, 5), (16, 8, 3), (5, 3, 2)])
def test_one_call_split_over_two_stream_sets_is_bit_identical(api, C, P, segments, knobs):
    """Round 4: a multi-person call is cut into segments that alternate between the caller's stream and an internal one (two
    scratch sets, event fork / join inside the call).  SNOWTRI_SPLIT_SEGMENTS=n forces the cut on a small batch (odd n is
    rounded up to an even count): outputs equal the uncut call bit for bit, and the oracle."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(77 + C + segments)
    F, J = (9 if C == 16 else 41), 133
    K, R, t = synth.ring_rig(C, radius=5.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
    npers = npers.copy()
    npers[2, 1] = P - 1
    prm = dict(PRM, keypoint_num=J, condense_person_num_tol=10 if C == 16 else 2)
    pout = P + 2
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    knobs.set("SNOWTRI_SPLIT_SEGMENTS", "1")
    whole = _run(api, K, R, t, prm, kp, npers, pout, knobs)
    knobs.set("SNOWTRI_SPLIT_SEGMENTS", str(segments))
    cut = _run(api, K, R, t, prm, kp, npers, pout, knobs)
    again = _run(api, K, R, t, prm, kp, npers, pout, knobs)
    knobs.clear("SNOWTRI_SPLIT_SEGMENTS")
    _check(cut, ref, pout, J, f"split {segments}")
    for k in ("xyzs", "pscore", "count", "flags"):
        assert np.array_equal(whole[k], cut[k], equal_nan=True), k
        assert np.array_equal(again[k], cut[k], equal_nan=True), k
    assert sum(whole["handed"]) == int(np.minimum(ref["count"], pout).sum())
    assert 0 < sum(cut["handed"]) < sum(whole["handed"])          # (the last segment alone)


@pytest.mark.parametrize("cfg,gen,rep,pout", [(3, 220, 10, 16), (5, 70, 30, 32)])
def test_batches_that_fill_the_chip_twice_are_split_by_default(api, cfg, gen, rep, pout, knobs):
    """No knob: a multi-person batch of >= 2 x 4 x 256 frames is cut in two by the library itself (8 x 4 and, since the
    internal stream is probed, 16 x 8 as well: 17.20 -> 17.01 ms per 12 500 frames).  Bit-identical to the uncut call, and
    the hand-over counters (last segment only) show that the cut happened."""
    import torch
    from snowmocap_amd import synth
    dev = torch.device("cuda", 0)
    wl = synth.config_workload(cfg, gen)
    K, R, t = wl["rig"]
    kp = torch.from_numpy(wl["kpts"]).to(dev).repeat(rep, 1, 1, 1, 1).contiguous()
    npers = torch.from_numpy(wl["n_persons"]).to(dev).repeat(rep, 1).contiguous()
    res = {}
    for mode in ("uncut", "default"):
        if mode == "uncut":
            knobs.set("SNOWTRI_SPLIT_SEGMENTS", "1")
        bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=np.float32)
        if mode == "uncut":
            knobs.clear("SNOWTRI_SPLIT_SEGMENTS")
        out = bt.run_torch(kp, npers)
        torch.cuda.synchronize(dev)
        res[mode] = ({k: v.clone() for k, v in out.items()}, sum(bt.ctx.last_handover_persons()), bt.ctx.stream_probes(), bt.ctx.last_stream_counts())
        bt.close()
    for k in ("xyzs", "pscore", "count", "flags"):
        assert torch.equal(res["uncut"][0][k].view(torch.int32), res["default"][0][k].view(torch.int32)), k
    total = int(res["uncut"][0]["count"].clamp(max=pout).sum())
    assert res["uncut"][1] == total and res["uncut"][2] == (0, 0, -1)
    assert 0 < res["default"][1] < total                     # the last of two segments
    assert res["default"][2][0] >= 1 and res["default"][2][2] == 1
    assert res["default"][3] == (0, 0, 0)


def test_internal_streams_are_probed_to_run_beside_the_callers(api, knobs):
    """The HIP runtime multiplexes a process's streams over a few hardware queues (4 by default, least-used first); an
    internal stream on the caller's queue serialises the split (8 x 4 float64: 1.19 -> 1.36 ms once the process had created
    some 40 streams).  Every internal stream is therefore probed when it is created (k_probe_wait / k_probe_set) and replaced
    if it does not run beside the caller's.  Here: a process with torch's stream pool alive, a dozen contexts one after the
    other on the null stream and on pool streams -- each reports a probe, a verdict of 1, and the uncut call's bits."""
    import torch
    from snowmocap_amd import synth
    dev = torch.device("cuda", 0)
    wl = synth.config_workload(3, 48)
    K, R, t = wl["rig"]
    kp = torch.from_numpy(wl["kpts"]).to(dev)
    npers = torch.from_numpy(wl["n_persons"]).to(dev)
    knobs.set("SNOWTRI_SPLIT_SEGMENTS", "1")
    bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=8, out_dtype=np.float64)
    whole = {k: v.clone() for k, v in bt.run_torch(kp, npers).items()}
    torch.cuda.synchronize(dev)
    assert bt.ctx.stream_probes() == (0, 0, -1)          # no internal stream without the split
    bt.close()
    knobs.set("SNOWTRI_SPLIT_SEGMENTS", "2")
    pool = [torch.cuda.Stream(device=dev) for _ in range(5)]
    discarded = 0
    for i in range(12):
        st = pool[i % 5] if i % 3 else torch.cuda.current_stream(dev)
        bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=8, out_dtype=np.float64)
        with torch.cuda.stream(st):
            out = bt.run_torch(kp, npers)
            out = bt.run_torch(kp, npers, out=out)       # (the same caller stream again: no second probe)
        torch.cuda.synchronize(dev)
        probes, replaced, verdict = bt.ctx.stream_probes()
        assert probes >= 1 and verdict == 1, (i, probes, replaced, verdict)
        assert probes == replaced + 1, (i, probes, replaced)
        discarded += replaced
        if i == 11:                                       # a call from another stream: one more probe, still side by side
            with torch.cuda.stream(pool[(i + 1) % 5]):
                out = bt.run_torch(kp, npers, out=out)
            torch.cuda.synchronize(dev)
            p2, r2, v2 = bt.ctx.stream_probes()
            assert p2 > probes and v2 == 1, (p2, r2, v2)
        for k in ("xyzs", "pscore", "count", "flags"):
            assert torch.equal(out[k].view(torch.int32), whole[k].view(torch.int32)), (i, k)
        bt.close()
    knobs.clear("SNOWTRI_SPLIT_SEGMENTS")
    print("internal streams discarded by the probe:", discarded)


def test_reference_workloads_take_no_fall_back_of_the_streaming_route(api):
    """parity alone cannot see a regression that sends every frame of the streaming route down one of its
    fall-backs (second association launch, exact candidate sums, k_frame_recompute).  On the BASELINE multi-person shapes
    all three counters are 0, the context reports no override, and the route is the streaming one."""
    from snowmocap_amd import synth
    for cfg, F, pout in ((3, 200, 16), (5, 24, 32)):
        wl = synth.config_workload(cfg, F)
        K, R, t = wl["rig"]
        bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=np.float32)
        out = bt.run_host(wl["kpts"], wl["n_persons"])
        assert out["status"] == 0
        assert bt.ctx.overrides() == ""
        assert "k_candidate_sums" in bt.ctx.last_kernel_names() and "k_associate" in bt.ctx.last_kernel_names()
        assert bt.ctx.last_stream_counts() == (0, 0, 0), (cfg, bt.ctx.last_stream_counts())
        bt.close()


@pytest.mark.parametrize("out_dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kn,center", [(133, 0), (30, 18), (1, 5)])
@pytest.mark.parametrize("C,P", [(8, 4), (16, 4), (5, 3)])
def test_streaming_route_for_float64_outputs_and_keypoint_num_below_J(api, C, P, kn, center, out_dtype, knobs):
    """the reference returns float64 arrays and its signature defaults are center_point_index = 18,
    keypoint_num = 30 (triangulation.py:95-100,136-148).  Both used to drop a multi-person batch to k_frame_recompute; now the
    streaming kernels are templated on the output type and fuse only the first keypoint_num joints, and the persons' mean
    scores (:150) come from the fused joints (k_person_scores).  Against the oracle (float64: 1e-8 m, 1e-9 relative), against
    the same batch inside k_frame_recompute (SNOWTRI_HANDOVER_MODE=0), with the persons that took each route read back."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    J = 133
    rng = np.random.default_rng(500 + 10 * C + kn)
    F = 6 if C == 16 else 16
    K, R, t = synth.ring_rig(C, radius=5.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(2.5, 9.0), permute_persons=True, dtype=np.float32)
    npers = npers.copy()
    npers[1, C - 1] = P - 1                 # a person one camera missed: a member-list cluster
    npers[3, :] = 0                         # an empty frame
    prm = dict(PRM, keypoint_num=kn, center_point_index=center, condense_person_num_tol=6 if C == 16 else 2)
    pout = P + 2
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    out = _run(api, K, R, t, prm, kp, npers, pout, knobs, out_dtype=out_dtype)
    off = _run(api, K, R, t, prm, kp, npers, pout, knobs, handover=False, out_dtype=out_dtype)
    assert out["xyzs"].dtype == out_dtype and out["xyzs"].shape == (F, pout, kn, 4)
    assert "k_associate" in out["kernels"] and "k_associate" not in off["kernels"]
    assert sum(out["handed"]) == int(np.minimum(ref["count"], pout).sum()) and out["handed"][0] > 0 and out["handed"][1] > 0, out["handed"]
    f64 = out_dtype == np.float64
    xyz_tol, rtol = (1e-8, 1e-9) if f64 else (XYZ_F32, 3e-7)
    for o, name in ((out, "streaming"), (off, "k_frame_recompute")):
        np.testing.assert_array_equal(o["count"], ref["count"], err_msg=name)
        for f in range(F):
            m = min(int(ref["count"][f]), pout)
            assert not o["xyzs"][f, m:].any() and not o["pscore"][f, m:].any(), f"{name} frame {f}: unused slots must be zero"
            if m:
                msg = f"{name} C={C} kn={kn} {np.dtype(out_dtype).name} frame {f}"
                assert_scores_close(o["xyzs"][f, :m, :, 3], ref["kscore"][f, :m], rtol=rtol, what=msg + " kscore")
                assert_xyz_close(o["xyzs"][f, :m, :, :3], ref["xyz"][f, :m], xyz_tol, score_ref=ref["kscore"][f, :m], what=msg + " xyz")
                assert_scores_close(o["pscore"][f, :m], ref["pscore"][f, :m], rtol=rtol, nterms=kn, what=msg + " pscore")


@pytest.mark.parametrize("out_dtype", [np.float32, np.float64])
def test_keypoint_num_below_J_with_an_active_score_filter(api, out_dtype, knobs):
    """keypoint_num < J with condense_score_tol > 0: the filter of :150-152 needs the mean over the FIRST keypoint_num joints
    before the slots are assigned, which the candidate sums over all J joints (:79) do not give -- a second launch of
    k_candidate_sums over the first keypoint_num joints feeds k_associate.  A tolerance in the middle of the persons' mean
    scores (persons dropped, slots move up) and one exactly on a person's mean (that frame is left to k_frame_recompute)."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(8)
    C, P, J, F = 6, 3, 133, 10
    K, R, t = synth.ring_rig(C, radius=5.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(2.5, 9.0), permute_persons=True, dtype=np.float32)
    prm = dict(PRM, keypoint_num=30, center_point_index=18)
    ref0 = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    f64 = out_dtype == np.float64
    for tol in (float(np.median(ref0["pscore"][ref0["pscore"] > 0])), float(ref0["pscore"][4, 1])):
        prm["condense_score_tol"] = tol
        ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
        assert 0 < ref["count"].sum() < ref0["count"].sum()
        out = _run(api, K, R, t, prm, kp, npers, P + 2, knobs, out_dtype=out_dtype)
        off = _run(api, K, R, t, prm, kp, npers, P + 2, knobs, handover=False, out_dtype=out_dtype)
        assert "k_associate" in out["kernels"] and sum(out["handed"]) > 0
        for o, name in ((out, "route"), (off, "k_frame_recompute")):
            np.testing.assert_array_equal(o["count"], ref["count"], err_msg=name)
            for f in range(F):
                m = min(int(ref["count"][f]), P + 2)
                assert not o["xyzs"][f, m:].any() and not o["pscore"][f, m:].any(), (name, f)
                if m:
                    msg = f"{name} tol={tol} frame {f}"
                    assert_scores_close(o["xyzs"][f, :m, :, 3], ref["kscore"][f, :m], rtol=1e-9 if f64 else 3e-7, what=msg + " kscore")
                    assert_xyz_close(o["xyzs"][f, :m, :, :3], ref["xyz"][f, :m], 1e-8 if f64 else XYZ_F32, score_ref=ref["kscore"][f, :m], what=msg + " xyz")
                    assert_scores_close(o["pscore"][f, :m], ref["pscore"][f, :m], rtol=1e-9 if f64 else 3e-7, nterms=30, what=msg + " pscore")
    # the tolerance that sits exactly on a person's mean score cannot be decided on the fast sums: that frame took the exact route
    assert out["stream_counts"][2] >= 1, out["stream_counts"]


def test_random_rigs_float64_outputs_and_keypoint_num(api, knobs):
    """The randomised sweep of test_random_rigs_with_handover_against_oracle_and_phase3 for the shapes round 4 brought to the
    streaming route: float64 or float32 outputs, keypoint_num anywhere in 1..J, 2..16 cameras -- against the oracle, and against
    k_frame_recompute on the same batch (condense_score_tol > 0 with keypoint_num < J: the second candidate-sum launch)."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(990)
    streamed = kept_off = 0
    for trial in range(36):
        C = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 12, 16]))
        P = int(rng.integers(2, 5))
        J = int(rng.choice([5, 20, 33, 40]))
        F = int(rng.integers(1, 5))
        kn = J if trial % 4 == 0 else int(rng.integers(1, J + 1))
        out_dtype = np.float64 if trial % 3 else np.float32
        K, R, t = synth.ring_rig(C, radius=float(rng.uniform(3.5, 6)))
        X = synth.make_people(rng, F, P, J=J)
        kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=float(rng.choice([0.3, 1.0])), score_range=(2.0, 8.0),
                                         permute_persons=True, dtype=np.float64 if trial % 5 == 0 else np.float32)
        npers = npers.copy()
        for _ in range(int(rng.integers(0, 3))):
            npers[rng.integers(0, F), rng.integers(0, C)] = rng.integers(0, P + 1)
        prm = dict(keypoint_score_threshold=float(rng.choice([0.0, 3.0, 5.0])), average_score_threshold=float(rng.choice([0.0, 0.3, 1.5])),
                   distance_threshold=float(rng.choice([0.02, 0.05, 1.0])), condense_distance_tol=float(rng.choice([0.05, 0.3, 10.0])),
                   condense_person_num_tol=int(rng.choice([0, 1, 2])), condense_score_tol=float(rng.choice([0.0, 0.0, 0.0, 0.5])),
                   center_point_index=int(rng.integers(0, J)), keypoint_num=kn)
        pout = int(rng.choice([1, 4, 16]))
        ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
        out = _run(api, K, R, t, prm, kp, npers, pout, knobs, out_dtype=out_dtype)
        off = _run(api, K, R, t, prm, kp, npers, pout, knobs, handover=False, out_dtype=out_dtype)
        msg = f"trial {trial}: C={C} P={P} J={J} kn={kn} F={F} {np.dtype(out_dtype).name} {prm} pout={pout} n={npers.tolist()}"
        on_route = "k_associate" in out["kernels"]
        assert on_route, msg
        streamed += on_route
        kept_off += kn < J and prm["condense_score_tol"] > 0.0      # (the shapes that need the second candidate-sum launch)
        f64 = out_dtype == np.float64
        xyz_tol, rtol = (1e-8, 1e-9) if f64 else (8e-6 if prm["distance_threshold"] >= 1.0 and C >= 9 else XYZ_F32, 3e-7)
        # (the conditioning budget of assert_scores_close -- a rounding-level error of the distance, 5e-14 m -- is sized for rays of
        # a 4-camera room; the rays of a 9-16 camera ring of up to 6 m radius are twice as long and meet under flatter angles)
        derr = 2e-13 if C >= 9 else 5e-14
        if prm["condense_distance_tol"] >= 10.0:
            # (everything merges: a fused score is the mean of hundreds of member scores, 1/dist-tailed.  8 cameras x 4 persons
            # too -- 448 members: soak round 404 of round 5, 1.32 x the 4-camera budget, with the library of round 4 as well)
            derr = 5e-13 if C >= 9 else 2e-13
        for o, name in ((out, "route"), (off, "k_frame_recompute")):
            np.testing.assert_array_equal(o["count"], ref["count"], err_msg=msg + " " + name)
            for f in range(F):
                m = min(int(ref["count"][f]), pout)
                assert not o["xyzs"][f, m:].any(), f"{msg} {name} frame {f}: unused slots must be zero"
                if m:
                    assert_scores_close(o["xyzs"][f, :m, :, 3], ref["kscore"][f, :m], rtol=rtol, dist_err=derr, what=f"{msg} {name} kscore frame {f}")
                    assert_xyz_close(o["xyzs"][f, :m, :, :3], ref["xyz"][f, :m], xyz_tol, score_ref=ref["kscore"][f, :m], what=f"{msg} {name} xyz frame {f}")
                    assert_scores_close(o["pscore"][f, :m], ref["pscore"][f, :m], rtol=rtol, dist_err=derr, nterms=kn, what=f"{msg} {name} pscore frame {f}")
    assert streamed > 20 and kept_off > 1, (streamed, kept_off)


def test_overlap_mode_with_multi_person_calls_is_bit_identical(api):
    """Overlap mode (BatchTriangulator(streams=n)) on the streaming multi-person route: whole calls rotate over the stream sets
    (each with its own hand-over lists, sums and counters; no split inside a call then); after join() every call's outputs
    equal the ones of the same calls issued one after the other."""
    import torch
    from snowmocap_amd import synth
    rng = np.random.default_rng(31)
    C, P, J, F = 6, 3, 40, 300
    K, R, t = synth.ring_rig(C, radius=5.0)
    dev = torch.device("cuda", 0)
    prm = dict(PRM, keypoint_num=J)
    batches = []
    for b in range(5):
        X = synth.make_people(rng, F, P, J=J)
        kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
        npers = npers.copy()
        npers[b, 1] = P - 1
        batches.append((torch.from_numpy(kp).to(dev), torch.from_numpy(npers).to(dev)))
    seq = api.BatchTriangulator(K, R, t, prm, pout_max=P + 2, out_dtype=np.float32)
    want = [{k: v.clone() for k, v in seq.run_torch(kp, npers).items()} for kp, npers in batches]
    torch.cuda.synchronize(dev)
    assert "k_associate" in seq.ctx.last_kernel_names()
    seq.close()
    ovl = api.BatchTriangulator(K, R, t, prm, pout_max=P + 2, out_dtype=np.float32, streams=3)
    for rep in range(2):
        got = [ovl.run_torch(kp, npers) for kp, npers in batches]
        ovl.join()
        torch.cuda.synchronize(dev)
        for g, w in zip(got, want):
            for k in ("xyzs", "pscore", "count", "flags"):
                assert torch.equal(g[k].view(torch.int32), w[k].view(torch.int32)), (rep, k)
    assert int(want[0]["count"].sum()) >= F * P - 5
    ovl.close()


@pytest.mark.parametrize("in_dtype", [np.float32, np.float64])
def test_records_that_are_not_finite_send_their_frame_to_the_exact_pass(api, in_dtype, knobs):
    """k_candidate_sums gates without compares (a sign trick, p1_tile_sums): a NaN cannot pass through it, so a NaN or infinite
    pixel or a NaN confidence in a LISTED row sends the frame to k_candidate_sums_exact where the record is written -- the
    result must be the reference's -- and the same values in rows a camera does not list must change nothing at all.  A negative
    distance_threshold (every pair gated; the sign of a zero would decide) keeps the batch off the streaming kernels."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(77)
    C, P, J, F = 8, 4, 133, 12
    K, R, t = synth.ring_rig(C, radius=4.5)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=in_dtype)
    npers = npers.copy()
    npers[5, 2] = 2                       # a ragged frame: rows 2 and 3 of camera 2 are not listed
    npers[9, 7] = 3
    prm = dict(PRM, keypoint_num=J)
    clean = _run(api, K, R, t, prm, kp, npers, P + 2, knobs)
    assert "k_candidate_sums<" in clean["kernels"], clean["kernels"]
    # unlisted rows full of NaN / inf: bit-identical outputs
    junk = kp.copy()
    junk[5, 2, 2:] = np.nan
    junk[9, 7, 3, :, 0] = np.inf
    junk[9, 7, 3, :, 2] = np.nan
    out = _run(api, K, R, t, prm, junk, npers, P + 2, knobs)
    for k in ("xyzs", "pscore", "count", "flags"):
        assert np.array_equal(out[k], clean[k], equal_nan=True), k
    # listed rows: one NaN pixel, one infinite pixel, one NaN confidence, one -inf confidence (simply gated), in four frames
    bad = kp.copy()
    bad[1, 3, 1, 17, 0] = np.nan
    bad[2, 0, 2, 130, 1] = np.inf
    bad[3, 6, 0, 60, 2] = np.nan
    bad[4, 5, 3, 5, 2] = -np.inf
    ref = orc.triangulate_condense_batch(K, R, t, bad, npers, orc.make_params(**prm), 64)
    out = _run(api, K, R, t, prm, bad, npers, P + 2, knobs)
    off = _run(api, K, R, t, prm, bad, npers, P + 2, knobs, handover=False)
    np.testing.assert_array_equal(out["count"], ref["count"])
    np.testing.assert_array_equal(off["count"], ref["count"])
    for f in range(F):
        m = min(int(ref["count"][f]), P + 2)
        g, o = out["xyzs"][f, :m].astype(np.float64), np.concatenate([ref["xyz"][f, :m], ref["kscore"][f, :m][..., None]], axis=-1)
        np.testing.assert_array_equal(np.isnan(g), np.isnan(o), err_msg=f"frame {f}")
        fin = np.isfinite(o[..., 3]) & (np.abs(o[..., 3]) < 1e9)
        assert np.abs(g[..., :3] - o[..., :3])[fin].max(initial=0.0) < XYZ_F32 * 4, f
        np.testing.assert_array_equal(np.isnan(out["pscore"][f, :m]), np.isnan(ref["pscore"][f, :m]), err_msg=f"pscore frame {f}")
    # the frames without a bad record are untouched
    for f in (0, 5, 6, 7, 8, 9, 10, 11):
        assert np.array_equal(out["xyzs"][f], clean["xyzs"][f]), f
    # negative distance_threshold: not a streaming batch, and the reference's result (nothing survives the pair gate)
    neg = dict(prm, distance_threshold=-0.05)
    refn = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**neg), 64)
    outn = _run(api, K, R, t, neg, kp, npers, P + 2, knobs)
    assert "k_candidate_sums<" not in outn["kernels"], outn["kernels"]
    np.testing.assert_array_equal(outn["count"], refn["count"])


def test_a_record_that_is_not_finite_in_the_second_candidate_sum_launch(api, knobs):
    """keypoint_num < J with an active condense_score_tol: k_candidate_sums runs a second time over the first keypoint_num joints,
    without an exact list -- a frame with a NaN record there gets NaN sums (which is what they are) and k_associate leaves it to
    k_frame_recompute.  Against the oracle, NaN patterns included."""
    from snowmocap_amd import synth
    from oracle import oracle as orc
    rng = np.random.default_rng(78)
    C, P, J, F, kn = 8, 4, 133, 8, 30
    K, R, t = synth.ring_rig(C, radius=4.5)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
    prm = dict(PRM, keypoint_num=kn, condense_score_tol=0.5)
    kp = kp.copy()
    kp[2, 1, 0, 7, 0] = np.nan        # inside the first keypoint_num joints: both launches see it
    kp[5, 4, 2, 100, 2] = np.nan      # behind them: only the launch over all J joints does
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), 64)
    out = _run(api, K, R, t, prm, kp, npers, P + 2, knobs)
    assert "k_candidate_sums<" in out["kernels"], out["kernels"]
    np.testing.assert_array_equal(out["count"], ref["count"])
    for f in range(F):
        m = min(int(ref["count"][f]), P + 2)
        g, o = out["xyzs"][f, :m].astype(np.float64), np.concatenate([ref["xyz"][f, :m], ref["kscore"][f, :m][..., None]], axis=-1)
        np.testing.assert_array_equal(np.isnan(g), np.isnan(o), err_msg=f"frame {f}")
        fin = np.isfinite(o[..., 3]) & (np.abs(o[..., 3]) < 1e9)
        assert np.abs(g[..., :3] - o[..., :3])[fin].max(initial=0.0) < XYZ_F32 * 4, f
        np.testing.assert_array_equal(np.isnan(out["pscore"][f, :m]), np.isnan(ref["pscore"][f, :m]), err_msg=f"pscore frame {f}")
        ok = np.isfinite(ref["pscore"][f, :m])
        assert_scores_close(out["pscore"][f, :m][ok], ref["pscore"][f, :m][ok], rtol=3e-7, nterms=kn, what=f"pscore frame {f}")


@pytest.mark.parametrize("C,P,kn", [(8, 4, 133), (8, 4, 30), (16, 2, 133), (5, 3, 133)])
def test_equal_rays_in_a_multi_person_batch_are_flagged(api, C, P, kn, knobs):
    """Round-5 advice: the candidate pass formed the determinant FUSED, fma(a, c, -(b b)) -- for equal rays the rounding error of
    b b, positive often enough for a finite (wrong) score and no flag.  Now det comes from separately rounded products (singular as
    the reference sees it <=> det == 0 exactly -> det * rsq(0) = NaN -> the frame goes to k_candidate_sums_exact, which tests
    a c == b b).  Camera 1 is camera 0 moved sideways (the same K and R under a general rig); person 0 of both shows one joint
    at the same pixel in two frames of three.  The flag is on exactly those frames -- also for a joint behind keypoint_num, and
    also when everything runs inside k_frame_recompute (the test build, SNOWTRI_HANDOVER_MODE=0)."""
    from snowmocap_amd import synth, _lib
    rng = np.random.default_rng(31 + C + P + kn)
    J, F = 133, 48
    K, R, t = synth.ring_rig(C, radius=4.5)
    K, R, t = K.copy(), R.copy(), t.copy()
    K[1], R[1] = K[0], R[0]
    t[1] = t[0] + R[0] @ np.array([0.4, 0.1, 0.0])
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=False, dtype=np.float32)
    kp = kp.copy()
    hit = np.arange(F) % 3 != 2
    joint = 100
    px = rng.uniform(200, 1000, size=(F, 2)).astype(np.float32)
    kp[hit, 0, 0, joint, :2] = px[hit]
    kp[hit, 1, 0, joint, :2] = px[hit]
    prm = dict(PRM, keypoint_num=kn, condense_person_num_tol=1)
    out = _run(api, K, R, t, prm, kp, npers, P + 2, knobs)
    assert "k_candidate_sums<" in out["kernels"], out["kernels"]
    sing = (out["flags"] & _lib.FLAG_SINGULAR) != 0
    assert np.array_equal(sing, hit), np.nonzero(sing != hit)[0]
    assert out["stream_counts"][1] >= int(hit.sum()), out["stream_counts"]      # the exact pass took those frames
    off = _run(api, K, R, t, prm, kp, npers, P + 2, knobs, handover=False)
    assert np.array_equal((off["flags"] & _lib.FLAG_SINGULAR) != 0, hit)
    ok = ~hit                                                                   # the other frames: the two routes agree as everywhere
    np.testing.assert_array_equal(out["count"][ok], off["count"][ok])

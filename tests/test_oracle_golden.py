"""Pins the CPU oracle (oracle/snowtri_oracle.c) to the reference's own outputs.

The fixtures under tests/golden/ were produced by running the unmodified reference
(tests/golden/make_golden.py).  Tolerances: <= 1e-11 m on candidate 3D points, <= 1e-10 m on fused joints,
<= 1e-9 relative on scores, identical candidate / person counts and error behaviour.
"""
import numpy as np
import pytest

from conftest import all_scenarios, assert_scores_close, assert_xyz_close, GOLDEN
from oracle import oracle as orc

XYZ_ATOL_FUSED = 1e-10   # fused joints: + (score rounding 2e-9) x (spread of mismatched candidates, metres)
XYZ_ATOL = 1e-11   # opposite ring cameras give near-parallel rays (cond ~1e3): LAPACK vs closed form


def _params(sc):
    return orc.make_params(**sc["params"])


@pytest.mark.parametrize("sc", all_scenarios())
def test_triangulation_candidates(sc):
    """A1 + A3 (camera.py:234-253, triangulation.py:50-93): kept candidates, order, scores."""
    prm = _params(sc)
    F = sc["kpts"].shape[0]
    cand_frames = list(sc["cand_frames"])
    for f in range(F):
        if sc["error"][f] == 1:
            with pytest.raises(np.linalg.LinAlgError):
                orc.human_triangulation_frame(sc["K"], sc["R"], sc["t"], sc["kpts"][f], sc["n_persons"][f], prm)
            continue
        res = orc.human_triangulation_frame(sc["K"], sc["R"], sc["t"], sc["kpts"][f], sc["n_persons"][f], prm)
        n = int(sc["cand_n"][f])
        assert len(res["hrnet_triangulate_points"]) == n
        assert_scores_close(res["hrnet_triangulate_person_scores"], sc["cand_pscore"][f, :n], what="cand pscore", nterms=sc["kpts"].shape[3])
        if f in cand_frames and n:
            i = cand_frames.index(f)
            ks = np.stack(res["hrnet_triangulate_keypoint_scores"])
            assert_scores_close(ks, sc["cand_kscore"][i, :n], what="cand kscore")
            assert_xyz_close(np.stack(res["hrnet_triangulate_points"]), sc["cand_xyz"][i, :n], XYZ_ATOL,
                             what="cand xyz")


@pytest.mark.parametrize("sc", all_scenarios())
def test_condense_on_reference_candidates(sc):
    """A4 alone (triangulation.py:95-162) fed with the REFERENCE's candidates."""
    prm = _params(sc)
    for i, f in enumerate(sc["cand_frames"]):
        n = int(sc["cand_n"][f])
        if sc["error"][f] == 1:
            continue
        res_in = {"hrnet_triangulate_points": [sc["cand_xyz"][i, k] for k in range(n)],
                  "hrnet_triangulate_keypoint_scores": [sc["cand_kscore"][i, k] for k in range(n)]}
        if sc["error"][f] == 2:
            with pytest.raises(IndexError):
                orc.condense_frame(res_in, prm)
            continue
        out = orc.condense_frame(res_in, prm)
        m = int(sc["cond_n"][f])
        assert len(out["hrnet_triangulate_points"]) == m
        if m:
            ks = np.stack(out["hrnet_triangulate_keypoint_scores"])
            assert_scores_close(ks, sc["cond_kscore"][f, :m], what="cond kscore")
            assert_scores_close(out["hrnet_triangulate_person_scores"], sc["cond_pscore"][f, :m], what="cond pscore", nterms=sc["kpts"].shape[3])
            assert_xyz_close(np.stack(out["hrnet_triangulate_points"]), sc["cond_xyz"][f, :m], XYZ_ATOL_FUSED,
                             score_ref=sc["cond_kscore"][f, :m], what="cond xyz")


@pytest.mark.parametrize("sc", all_scenarios())
def test_batch_end_to_end(sc):
    """The whole per-frame sequence of main.py:50-71 through the batched oracle entry."""
    prm = _params(sc)
    if prm.keypoint_num < 0 or prm.keypoint_num > sc["kpts"].shape[3]:
        pytest.skip("IndexError scenario: covered per frame above")
    pout = max(1, sc["cond_xyz"].shape[1])
    out = orc.triangulate_condense_batch(sc["K"], sc["R"], sc["t"], sc["kpts"], sc["n_persons"], prm, pout)
    assert np.array_equal(out["status"] == orc.ORC_SINGULAR, sc["error"] == 1)
    ok = sc["error"] == 0
    assert np.array_equal(out["count"][ok], sc["cond_n"][ok])
    for f in np.nonzero(ok)[0]:
        m = int(sc["cond_n"][f])
        if not m:
            continue
        assert_scores_close(out["kscore"][f, :m], sc["cond_kscore"][f, :m], what="kscore")
        assert_scores_close(out["pscore"][f, :m], sc["cond_pscore"][f, :m], what="pscore", nterms=sc["kpts"].shape[3])
        assert_xyz_close(out["xyz"][f, :m], sc["cond_xyz"][f, :m], XYZ_ATOL_FUSED,
                         score_ref=sc["cond_kscore"][f, :m], what="xyz")


def test_skew_ray_solver_unit_vectors():
    """A2 (triangulation.py:24-31) on 1000 recorded ray pairs, incl. near-parallel ones."""
    z = np.load(f"{GOLDEN}/g5_skew_ray.npz")
    dist, W, sing = orc.skew_ray_solver_batch(z["hm"], z["hs"], z["tm"], z["ts"])
    assert not sing.any()
    well = np.ones(len(dist), bool)
    well[100:120] = False                       # near-parallel pairs: conditioning ~1e12
    np.testing.assert_allclose(dist[well], z["dist"][well], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(W[well], z["W"][well], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(W[~well], z["W"][~well], rtol=0.2, atol=1e-3)
    with pytest.raises(np.linalg.LinAlgError):
        orc.skew_ray_solver([1, 2, 3], [2, 4, 6], [0, 0, 0], [1, 1, 1])


def test_second_order_track():
    """N1: SecondOrderDynamic via Human_Triangulation_Smooth (triangulation.py:4-22,164-186)."""
    z = np.load(f"{GOLDEN}/g6_smooth_blender.npz")
    y = orc.second_order_track(z["track"], float(z["f"]), float(z["z"]), float(z["r"]), float(z["dt"]))
    np.testing.assert_allclose(y, z["smoothed"], rtol=0, atol=1e-12)


def test_blender_restatement_against_reference():
    """N2: oracle/blender.py vs the reference's Human_Triangulation_Blender (fixtures G6, G8) and
    Human_Triangulation_Blender_Smooth over an 80-frame track with invalid points (G8)."""
    from oracle import blender as ob
    z6 = np.load(f"{GOLDEN}/g6_smooth_blender.npz")
    assert [str(n) for n in z6["blender_names"]] == ob.NAMES
    pts, val = ob.control_points_track(z6["blender_persons"])
    assert val.all()
    np.testing.assert_allclose(pts, np.nan_to_num(z6["blender_ctrl"]), rtol=0, atol=1e-15)
    z8 = np.load(f"{GOLDEN}/g8_blender_track.npz")
    pts, val = ob.control_points_track(z8["track"])
    np.testing.assert_array_equal(val, z8["valid"])
    ok = z8["valid"].astype(bool)
    np.testing.assert_allclose(pts[ok], z8["raw"][ok], rtol=0, atol=1e-15)
    assert np.isnan(pts[~ok][:, :3]).any(axis=1).all()
    raw = z8["raw"].copy()
    raw[~ok] = np.nan
    sm = ob.smooth_track(raw, z8["valid"], z8["fzr"], float(z8["dt"]))
    keep = np.ones(sm.shape[:3], bool)
    keep[0] = ok[0]                                   # frame 0 passes through: NaN where invalid
    np.testing.assert_allclose(sm[keep], z8["smoothed"][keep], rtol=0, atol=1e-13)
    assert np.isnan(sm[0][~ok[0]][:, :3]).all()
    # a NaN pelvis matrix: the reference raises inside SciPy's SVD
    bad = z8["track"][5, 0].copy()
    bad[11] = bad[12] = 0.0
    with pytest.raises(np.linalg.LinAlgError):
        ob.control_points(bad)


def test_undistort_restatement_properties():
    """N4 (parity unpinned upstream: no OpenCV here, no reference vector): the inverse model is pinned to the
    forward model by round trip on the shipped rig's four lenses, and to OpenCV's fixed-point scheme run to
    convergence."""
    from oracle import undistort as ou
    from snowmocap_amd import synth
    K, _, _ = synth.load_rig_json()
    D = synth.load_rig_distortion()
    assert D.shape == (4, 5) and np.abs(D[:, 0]).min() > 0.1
    rng = np.random.default_rng(4)
    for c in range(4):
        uv = np.stack([rng.uniform(0, 1280, 5000), rng.uniform(0, 720, 5000)], -1)
        raw = ou.distort_pixels(K[c], D[c], uv)
        assert np.abs(raw - uv).max() > 20                       # tens of pixels at the corners
        back = ou.undistort_pixels(K[c], D[c], raw)
        np.testing.assert_allclose(back, uv, rtol=0, atol=1e-10)
        np.testing.assert_allclose(ou.undistort_opencv5(K[c], D[c], raw, iters=60), back, rtol=0, atol=1e-10)
        assert np.abs(ou.undistort_opencv5(K[c], D[c], raw) - back).max() < 0.5   # cv2's 5 sweeps: sub-pixel, not exact
        np.testing.assert_array_equal(ou.undistort_pixels(K[c], np.zeros(5), raw), ou.distort_pixels(K[c], np.zeros(5), raw))
        np.testing.assert_allclose(ou.undistort_pixels(K[c], np.zeros(5), raw), raw, rtol=0, atol=1e-12)

"""The DEFAULT dispatch past the limits of the streaming kernels (round-4 review, item 7): no SNOWTRI_* knob is set anywhere
in this file -- the reference's loops (triangulation.py:56-65) have no limit on cameras or persons, so whatever the library
routes these shapes to must give the reference's result.

  * 17 and 20 cameras x 2 persons (more than the 16 cameras / 120 pairs of the LDS pair table),
  * 3 cameras x 20 persons (more than the 16 persons per camera of the descriptors' 4-bit person fields),
  * DLT with keypoint_num 256 / 257 (the bound of its multi-detection kernel),
  * F = 0, and an output of more than 4 GiB through the host entry.
"""
import os

import numpy as np
import pytest

from conftest import assert_scores_close, assert_xyz_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import snowmocap_amd as sm
    from snowmocap_amd import _lib
    assert _lib.lib().snowtri_device_count() > 0, "these tests need the HIP device"
    assert not [k for k in os.environ if k.startswith("SNOWTRI_") and k != "SNOWTRI_LIB"], "these tests run the default dispatch"
    return sm


def _against_oracle(api, K, R, t, prm, kp, npers, pout, out_dtype, msg):
    from oracle import oracle as orc
    ref = orc.triangulate_condense_batch(K, R, t, kp, npers, orc.make_params(**prm), pout)
    bt = api.BatchTriangulator(K, R, t, prm, pout_max=pout, out_dtype=out_dtype)
    assert bt.ctx.overrides() == "", bt.ctx.overrides()
    out = bt.run_host(kp, npers)
    names = bt.ctx.last_kernel_names()
    bt.close()
    f32 = np.dtype(out_dtype) == np.float32
    J = prm["keypoint_num"]
    assert np.array_equal(out["count"], ref["count"]), (msg, names, out["count"], ref["count"])
    for f in range(kp.shape[0]):
        m = min(int(ref["count"][f]), pout)
        assert not out["xyzs"][f, m:].any(), f"{msg} frame {f}: unused slots must be zero ({names})"
        if m:
            assert_scores_close(out["xyzs"][f, :m, :, 3], ref["kscore"][f, :m], rtol=3e-7 if f32 else 1e-9, what=f"{msg} kscore frame {f}")
            assert_xyz_close(out["xyzs"][f, :m, :, :3], ref["xyz"][f, :m], 2e-6 if f32 else 1e-8, score_ref=ref["kscore"][f, :m],
                             what=f"{msg} xyz frame {f} ({names})")
            assert_scores_close(out["pscore"][f, :m], ref["pscore"][f, :m], rtol=3e-7 if f32 else 1e-9, nterms=J, what=f"{msg} pscore frame {f}")
    return names, ref


@pytest.mark.parametrize("C", [17, 20])
@pytest.mark.parametrize("out_dtype", [np.float32, np.float64])
def test_rings_of_more_than_sixteen_cameras(api, C, out_dtype):
    """136 / 190 camera pairs: beyond the pair table of the streaming kernels and of k_frame_recompute."""
    from snowmocap_amd import synth
    rng = np.random.default_rng(1700 + C)
    K, R, t = synth.ring_rig(C, radius=5.0)
    F, P = 5, 2
    X = synth.make_people(rng, F, P)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(2.0, 8.0), permute_persons=True)
    npers = npers.copy()
    npers[1, C - 1] = 1
    npers[3, 0] = 0
    prm = dict(synth.default_thresholds(), average_score_threshold=1.0, condense_distance_tol=0.3, condense_person_num_tol=10)
    names, ref = _against_oracle(api, K, R, t, prm, kp, npers, 6, out_dtype, f"C={C}")
    assert (ref["count"] >= 2).all(), ref["count"]
    # one person per camera on such a rig too (the single-detection kernels stop at 8 cameras)
    X1 = synth.make_people(rng, 4, 1)
    kp1, np1 = synth.make_keypoints(rng, K, R, t, X1, pixel_sigma=0.7, score_range=(2.0, 8.0))
    _against_oracle(api, K, R, t, dict(synth.default_thresholds()), kp1, np1, 2, out_dtype, f"C={C} x 1")


@pytest.mark.parametrize("out_dtype", [np.float32, np.float64])
def test_twenty_persons_per_camera(api, out_dtype):
    """3 cameras x 20 persons: 1 200 candidate slots per frame, more persons than a descriptor's 4-bit fields index."""
    from snowmocap_amd import synth
    rng = np.random.default_rng(320)
    K, R, t = synth.ring_rig(3, radius=6.0)
    F, P = 4, 20
    centres = np.stack([np.array([1.6 * (i % 5) - 3.2, 1.6 * (i // 5) - 2.4, 0.0]) for i in range(P)])
    X = synth.make_people(rng, F, P, centres=centres)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.5, score_range=(3.5, 8.0), permute_persons=True)
    npers = npers.copy()
    npers[2, 1] = 17
    prm = dict(synth.default_thresholds(), average_score_threshold=1.0, condense_distance_tol=0.3, condense_person_num_tol=2)
    names, ref = _against_oracle(api, K, R, t, prm, kp, npers, 32, out_dtype, "3 x 20")
    assert ref["count"].max() >= 15, ref["count"]


def test_dlt_keypoint_num_at_and_past_256(api):
    """SNOWTRI_DLT: one detection per camera takes any keypoint_num; several detections per camera take <= 256 (the header says
    so) and refuse 257 with SNOWTRI_ERR_BAD_ARG instead of computing something else."""
    from snowmocap_amd import synth, _lib
    from oracle import dlt, oracle as orc
    rng = np.random.default_rng(256)
    J = 260
    K, R, t = synth.load_rig_json()
    X = synth.make_people(rng, 3, 1, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.5, score_range=(2.0, 8.0))
    for kn in (256, 257, 260):
        prm = dict(synth.default_thresholds(), keypoint_num=kn)
        want, wps, wcnt = dlt.dlt_batch(K, R, t, kp, prm["keypoint_score_threshold"], kn)
        bt = api.BatchTriangulator(K, R, t, prm, pout_max=1, out_dtype=np.float64, method=_lib.DLT)
        out = bt.run_host(kp, npers)
        bt.close()
        assert out["xyzs"].shape == (3, 1, kn, 4) and (out["count"] == 1).all()
        assert np.abs(out["xyzs"][..., :3] - want[..., :3]).max() < 1e-9
        np.testing.assert_allclose(out["pscore"], wps, rtol=1e-6)
    K8, R8, t8 = synth.ring_rig(8)
    X2 = synth.make_people(rng, 2, 2, J=J)
    kp2, np2 = synth.make_keypoints(rng, K8, R8, t8, X2, pixel_sigma=0.5, score_range=(3.5, 8.0), dtype=np.float64)
    prm = dict(synth.default_thresholds(), average_score_threshold=1.0, condense_distance_tol=0.3, keypoint_num=256)
    want, wps, wcnt = dlt.dlt_multi_batch(K8, R8, t8, kp2, np2, orc.make_params(**prm), 4)
    bt = api.BatchTriangulator(K8, R8, t8, prm, pout_max=4, out_dtype=np.float64, method=_lib.DLT)
    out = bt.run_host(kp2, np2)
    bt.close()
    np.testing.assert_array_equal(out["count"], wcnt)
    assert np.abs(out["xyzs"][..., :3] - want[..., :3]).max() < 1e-9
    bt = api.BatchTriangulator(K8, R8, t8, dict(prm, keypoint_num=257), pout_max=4, out_dtype=np.float64, method=_lib.DLT)
    with pytest.raises(_lib.SnowtriError) as e:
        bt.run_host(kp2, np2)
    assert e.value.status == _lib.ERR_BAD_ARG
    bt.close()


def test_empty_batch(api):
    """F = 0 through the host and the device entry: OK, nothing written, nothing launched."""
    import torch
    from snowmocap_amd import synth, _lib
    wl = synth.config_workload(2, 1)
    K, R, t = wl["rig"]
    bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=2, out_dtype=np.float32)
    out = bt.run_host(np.zeros((0, 4, 1, 133, 3), np.float32), np.zeros((0, 4), np.int32))
    assert out["status"] == _lib.OK and out["xyzs"].shape == (0, 2, 133, 4) and out["count"].shape == (0,)
    outd = bt.run_torch(torch.zeros((0, 4, 1, 133, 3), device="cuda"), torch.zeros((0, 4), dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    assert outd["xyzs"].shape == (0, 2, 133, 4)
    # and a batch right behind it still works
    ok = bt.run_host(wl["kpts"], wl["n_persons"])
    assert ok["count"][0] == 1
    bt.close()


def test_output_of_more_than_four_gib_through_the_host_entry(api):
    """130 000 frames x 8 slots x 133 joints x 32 B (float64 records) = 4.43 GB of joints (> 2^32 bytes) for 0.83 GB of
    keypoints: byte offsets past 32 bits in the kernels' stores, the zero-fill of the unused slots and the staged download.
    Checked: the head, the middle and the tail of the batch against the oracle, every unused slot zero, counts and flags."""
    from snowmocap_amd import synth, _lib
    from oracle import oracle as orc
    gen, reps, pout = 1000, 130, 8
    wl = synth.config_workload(2, gen, seed=5)
    K, R, t = wl["rig"]
    kp = np.tile(wl["kpts"], (reps, 1, 1, 1, 1))
    npers = np.tile(wl["n_persons"], (reps, 1))
    F = kp.shape[0]
    bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=np.float64)
    out = bt.run_host(kp, npers)
    bt.close()
    assert out["xyzs"].nbytes > 2 ** 32 and out["status"] == _lib.OK
    assert (out["count"] == 1).all() and ((out["flags"] & _lib.FLAG_FASTPATH) != 0).all()
    ref = orc.triangulate_condense_batch(K, R, t, wl["kpts"], wl["n_persons"], orc.make_params(**wl["params"]), 1)
    for block in (0, reps // 2, reps - 1):            # the same 1 000 frames everywhere: head, middle, tail
        sl = slice(block * gen, (block + 1) * gen)
        assert np.abs(out["xyzs"][sl, 0, :, :3] - ref["xyz"][:, 0]).max() < 1e-8
        assert_scores_close(out["xyzs"][sl, 0, :, 3], ref["kscore"][:, 0], rtol=1e-9, what=f"block {block}")
    assert np.array_equal(out["xyzs"][(reps - 1) * gen:], out["xyzs"][:gen])      # bit for bit: the same frames
    for s in range(1, pout):
        assert not out["xyzs"][:, s].any(), f"slot {s} must be zero-filled"
    assert not out["pscore"][:, 1:].any()

"""The sizes ONE rank processes in the 8-GPU configurations of BASELINE.json, on one GPU:

  configs[3]  4 cameras x 1 person, 1 000 000 frames over 8 GPUs  -> a 125 000-frame shard (k_fused_lean)
  configs[4]  16 cameras x 8 persons, 100 000 frames over 8 GPUs  -> a 12 500-frame shard (k_frame_recompute)

Size-independent properties at the full shard size -- determinism, invariance to how the shard is cut into launches
(frames are independent: bit-identical), every frame resolved -- and an oracle check of a random sample of frames
(copied back from the device, so the oracle sees exactly what the kernel saw).  The shard's frames are generated as a
smaller host batch tiled on the device with fresh sub-pixel jitter per tile (distinct frames, seconds to build).
What stays untested here is only the sharded EXECUTION on several GPUs (the driver's SCALE run).
"""
import numpy as np
import pytest

from conftest import assert_scores_close, assert_xyz_close

pytestmark = pytest.mark.gpu

XYZ_F32 = 2e-6


@pytest.fixture(scope="module")
def api():
    import snowmocap_amd as sm
    from snowmocap_amd import _lib
    assert _lib.lib().snowtri_device_count() > 0, "these tests need the HIP device"
    return sm


def _tiled_shard(cfg, F, base_frames, seed):
    import torch
    from snowmocap_amd import synth
    wl = synth.config_workload(cfg, base_frames, seed=seed)
    dev = torch.device("cuda", 0)
    rep = (F + base_frames - 1) // base_frames
    base = torch.from_numpy(wl["kpts"]).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    tiles = []
    for r in range(rep):
        tile = base.clone()
        if r:
            tile[..., :2] += torch.randn(base.shape[:-1] + (2,), generator=gen, device=dev) * 0.25
        tiles.append(tile)
    kp = torch.cat(tiles, dim=0)[:F].contiguous()
    npers = torch.from_numpy(wl["n_persons"]).to(dev).repeat(rep, 1)[:F].contiguous()
    return wl, kp, npers


def _run_pieces(bt, kp, npers, cuts):
    import torch
    xs, cs = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        o = bt.run_torch(kp[lo:hi].contiguous(), None if npers is None else npers[lo:hi].contiguous())
        torch.cuda.synchronize()
        xs.append(o["xyzs"].cpu().numpy())
        cs.append(o["count"].cpu().numpy())
    return np.concatenate(xs), np.concatenate(cs)


def test_config3_shard_125000_frames(api):
    """One rank's share of configs[3]: 125 000 frames of the 4-camera single-person workload."""
    import torch
    from snowmocap_amd import _lib
    from oracle import oracle as orc
    F = 125000
    wl, kp, npers = _tiled_shard(2, F, 25000, seed=31)
    K, R, t = wl["rig"]
    bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float32)
    full = bt.run_torch(kp)
    torch.cuda.synchronize()
    a = full["xyzs"].cpu().numpy().copy()
    ps = full["pscore"].cpu().numpy().copy()
    again = bt.run_torch(kp)
    torch.cuda.synchronize()
    assert np.array_equal(a, again["xyzs"].cpu().numpy()) and np.array_equal(ps, again["pscore"].cpu().numpy()), "not deterministic"
    assert (full["count"].cpu().numpy() == 1).all()
    assert ((full["flags"].cpu().numpy() & _lib.FLAG_FASTPATH) != 0).all()
    b, cb = _run_pieces(bt, kp, None, [0, 1, 41667, 41668, 100000, F])
    assert np.array_equal(a, b) and (cb == 1).all(), "cutting the shard into launches changed the result"
    idx = np.sort(np.random.default_rng(3).choice(F, 64, replace=False))
    sample = kp[torch.from_numpy(idx).to(kp.device)].cpu().numpy()
    ref = orc.triangulate_condense_batch(K, R, t, sample, np.ones((64, 4), np.int32), orc.make_params(**wl["params"]), 1)
    assert (ref["count"] == 1).all()
    assert_xyz_close(a[idx][..., :3], ref["xyz"], XYZ_F32, score_ref=ref["kscore"])
    assert_scores_close(a[idx][..., 3], ref["kscore"], rtol=3e-7)
    assert_scores_close(ps[idx], ref["pscore"], rtol=3e-7, nterms=133)
    bt.close()


def test_config4_shard_12500_frames(api):
    """One rank's share of configs[4]: 12 500 frames of the 16-camera x 8-person workload (association + fusion)."""
    import torch
    from snowmocap_amd import _lib
    from oracle import oracle as orc
    F = 12500
    wl, kp, npers = _tiled_shard(5, F, 500, seed=41)
    K, R, t = wl["rig"]
    P, pout = wl["X"].shape[1], 32
    bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=np.float32)
    full = bt.run_torch(kp, npers)
    torch.cuda.synchronize()
    a, cnt = full["xyzs"].cpu().numpy().copy(), full["count"].cpu().numpy().copy()
    again = bt.run_torch(kp, npers)
    torch.cuda.synchronize()
    assert np.array_equal(a, again["xyzs"].cpu().numpy()) and np.array_equal(cnt, again["count"].cpu().numpy()), "not deterministic"
    b, cb = _run_pieces(bt, kp, npers, [0, 4167, 4168, F])
    assert np.array_equal(a, b) and np.array_equal(cnt, cb), "cutting the shard into launches changed the result"
    assert (cnt >= P).all() and (cnt <= pout).all()
    assert not ((full["flags"].cpu().numpy() & (_lib.FLAG_SINGULAR | _lib.FLAG_OVERFLOW)) != 0).any()
    idx = np.sort(np.random.default_rng(5).choice(F, 32, replace=False))
    sel = torch.from_numpy(idx).to(kp.device)
    ref = orc.triangulate_condense_batch(K, R, t, kp[sel].cpu().numpy(), npers[sel].cpu().numpy(), orc.make_params(**wl["params"]), pout)
    assert np.array_equal(cnt[idx], ref["count"])
    for i, f in enumerate(idx):
        m = int(ref["count"][i])
        assert_xyz_close(a[f, :m, :, :3], ref["xyz"][i, :m], XYZ_F32, score_ref=ref["kscore"][i, :m])
        assert_scores_close(a[f, :m, :, 3], ref["kscore"][i, :m], rtol=3e-7)
    bt.close()


@pytest.mark.parametrize("method", ["pairwise", "dlt"])
def test_float64_outputs_do_not_depend_on_the_launch_shape(api, method):
    """k_fused_single (float64 outputs): 30 000 frames in one launch == the same frames in uneven pieces, bit for bit.
    (The kernel picks its tile size from the launch size, so a frame lands in a different slot of the prefetch ring:
    implicit FMA contraction used to differ between the ring's code copies.)"""
    import torch
    from snowmocap_amd import _lib
    F = 30000
    wl, kp, npers = _tiled_shard(2, F, 10000, seed=51)
    K, R, t = wl["rig"]
    bt = api.BatchTriangulator(K, R, t, wl["params"], pout_max=1, out_dtype=np.float64,
                               method=_lib.DLT if method == "dlt" else _lib.PAIRWISE)
    full = bt.run_torch(kp)
    torch.cuda.synchronize()
    a = full["xyzs"].cpu().numpy().copy()
    b, cb = _run_pieces(bt, kp, None, [0, 1, 777, 10001, 23456, F])
    assert np.array_equal(a, b) and (cb == 1).all()
    bt.close()

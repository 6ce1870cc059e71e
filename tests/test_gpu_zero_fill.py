"""SNOWTRI_CALL_NO_ZERO_FILL (snowtri_triangulate_condense_ex): the slots behind out_count[f] are left as the caller's buffer
holds them (round-5 review, item 6).  The reference returns LISTS of count persons (triangulation.py:154-160); the [F][Pout_max]
padding is this ABI's artefact and at a generous Pout_max the zeros are most of what a multi-person call writes.  Checked on
every route that pads: the used slots equal the default call's bit for bit, count / flags are the same, the unused slots still
hold the sentinel the test put there -- and the default call zero-fills as before."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SENTINEL = 7.25


def _both(api, K, R, t, prm, kp, npers, pout, out_dtype, knobs=None, env=None):
    import torch
    for k, v in (env or {}).items():
        knobs.set(k, v)
    dev = torch.device("cuda", 0)
    kpd = torch.from_numpy(np.ascontiguousarray(kp)).to(dev)
    npd = torch.from_numpy(np.ascontiguousarray(npers, dtype=np.int32)).to(dev)
    res = []
    for zero_fill in (True, False):
        bt = api.BatchTriangulator(K, R, t, prm, pout_max=pout, out_dtype=out_dtype, zero_fill=zero_fill)
        out = bt.alloc_outputs(kp.shape[0], dev)
        out["xyzs"].fill_(SENTINEL)
        out["pscore"].fill_(SENTINEL)
        bt.run_torch(kpd, npd, out=out)
        torch.cuda.synchronize(dev)
        o = {k: v.cpu().numpy() for k, v in out.items()}
        o["names"] = bt.ctx.last_kernel_names()
        res.append(o)
        bt.close()
    if env:
        knobs.clear(*env)
    return res


def _check(filled, sparse, pout, must_skip):
    assert np.array_equal(filled["count"], sparse["count"]) and np.array_equal(filled["flags"], sparse["flags"])
    F = len(filled["count"])
    used = np.arange(pout)[None, :] < np.minimum(filled["count"], pout)[:, None]        # [F, pout]
    assert np.array_equal(filled["xyzs"][used], sparse["xyzs"][used], equal_nan=True)
    assert np.array_equal(filled["pscore"][used], sparse["pscore"][used], equal_nan=True)
    assert not filled["xyzs"][~used].any() and not filled["pscore"][~used].any()          # the default: zeros, as documented
    left = sparse["xyzs"][~used]
    assert np.all((left == SENTINEL) | (left == 0.0))                                      # unspecified = untouched or zero, never garbage
    if must_skip and (~used).any():
        assert (left == SENTINEL).mean() > 0.99, (left == SENTINEL).mean()                 # this route really skips the stores
    return int((~used).sum()), F


@pytest.mark.parametrize("out_dtype", [np.float32, np.float64])
def test_streaming_multi_person_route_skips_the_padding(out_dtype):
    import snowmocap_amd as api
    from snowmocap_amd import synth
    wl = synth.config_workload(3, 601, seed=5)            # 8 cameras x 4 persons, ~5 persons per frame, ragged counts
    K, R, t = wl["rig"]
    filled, sparse = _both(api, K, R, t, wl["params"], wl["kpts"], wl["n_persons"], 16, out_dtype)
    assert "k_associate<" in sparse["names"], sparse["names"]
    unused, F = _check(filled, sparse, 16, must_skip=True)
    assert unused > 8 * F                                  # more than half of the slots are padding at Pout_max 16


def test_other_routes_honour_the_flag_or_zero_fill(knobs):
    import snowmocap_amd as api
    from snowmocap_amd import synth
    rng = np.random.default_rng(3)
    # one detection per camera, three slots: k_fused_single pads slots 1 and 2 of every frame
    wl = synth.config_workload(2, 300, seed=9)
    K, R, t = wl["rig"]
    prm = dict(wl["params"], keypoint_num=40)
    filled, sparse = _both(api, K, R, t, prm, wl["kpts"], wl["n_persons"], 3, np.float32)
    assert "k_fused_single<" in sparse["names"], sparse["names"]
    _check(filled, sparse, 3, must_skip=True)
    # the lean kernel has one slot and nothing to pad: the flag changes nothing
    filled, sparse = _both(api, K, R, t, wl["params"], wl["kpts"], wl["n_persons"], 1, np.float32)
    assert "k_fused_lean" in sparse["names"] and np.array_equal(filled["xyzs"], sparse["xyzs"])
    # everything inside k_frame_recompute (a forced route: the test build)
    C, P, J, F = 5, 3, 40, 24
    K, R, t = synth.ring_rig(C, radius=5.0)
    X = synth.make_people(rng, F, P, J=J)
    kp, npers = synth.make_keypoints(rng, K, R, t, X, pixel_sigma=0.7, score_range=(3.5, 9.0), permute_persons=True, dtype=np.float32)
    prm = dict(keypoint_score_threshold=3.0, average_score_threshold=0.3, distance_threshold=0.05, condense_distance_tol=0.3,
               condense_person_num_tol=2, condense_score_tol=0.0, center_point_index=0, keypoint_num=J)
    filled, sparse = _both(api, K, R, t, prm, kp, npers, 8, np.float32, knobs=knobs, env={"SNOWTRI_HANDOVER_MODE": "0"})
    assert "k_frame_recompute<" in sparse["names"] and "k_associate" not in sparse["names"], sparse["names"]
    _check(filled, sparse, 8, must_skip=True)
    # the spill kernel (condense_frame)
    filled, sparse = _both(api, K, R, t, prm, kp, npers, 8, np.float64, knobs=knobs, env={"SNOWTRI_GENERAL_MODE": "1"})
    assert "k_frame_general<" in sparse["names"], sparse["names"]
    _check(filled, sparse, 8, must_skip=True)


def test_compact_gather_reads_only_what_the_flag_keeps():
    """The consumer the flag is for: sharded.gather_track_compact packs the persons by their counts, so a triangulator that
    skips the padding gathers the same persons (one rank, RCCL group of one)."""
    import os
    import torch
    import torch.distributed as dist
    import snowmocap_amd as api
    from snowmocap_amd import synth
    from snowmocap_amd.sharded import ShardedTriangulator
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(25000 + os.getpid() % 4000))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        wl = synth.config_workload(3, 257, seed=4)
        K, R, t = wl["rig"]
        kp = torch.from_numpy(wl["kpts"]).to(dev)
        npers = torch.from_numpy(wl["n_persons"]).to(dev)
        outs = []
        for zero_fill in (True, False):
            st = ShardedTriangulator(K, R, t, wl["params"], pout_max=12, device=0, chunks=2, compact=True, zero_fill=zero_fill)
            o = st.run(kp, 257, npers, strict=True)
            torch.cuda.synchronize(dev)
            outs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()})
            st.bt.close()
        for k in ("persons", "pscore", "offsets", "stored", "count", "flags"):
            assert torch.equal(outs[0][k], outs[1][k]), k
    finally:
        dist.destroy_process_group()

"""The sharded path with the REAL kernels and more than one rank, on the one GPU a build box has (round-4 review, item 3).

RCCL refuses two ranks on one device, so the ranks share cuda:0 and the collectives of the group go through gloo, staged
through page-locked host buffers on the stream they were queued on (sharded.all_gather_flat).  Everything else is the
product path: every rank computes its frame block with ShardedTriangulator on the GPU, the gather of a piece runs on the
side stream under the kernels of the next piece (events, two slots, slot reuse), smooth_track_sharded filters a block
with the device-side combine between its all-gather and its fix -- now under genuine multi-process interleaving on the
device.  Each rank compares the gathered track with ITS OWN unsharded launch of the whole batch: bit for bit.

world_size 2 and 4; BASELINE configs[1]-shaped (4 x 1) and configs[2]-shaped (8 x 4, the streaming multi-person route)
batches; uneven blocks and empty trailing blocks; padded and compact gathers.  (e) of SURVEY 8 stays "unmeasured on
hardware" for more than one GPU: this covers the logic, not xGMI.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, case, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from snowmocap_amd import synth
    from snowmocap_amd.batch import BatchTriangulator
    from snowmocap_amd.sharded import ShardedTriangulator, shard_bounds, smooth_track_sharded, compact_to_padded
    import snowmocap_amd as sm
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ok, notes = True, []

    def expect(cond, what):
        nonlocal ok
        if not cond:
            ok = False
            notes.append(what)

    cfg, F, pout, chunks = case
    gen = min(F, 400)
    wl = synth.config_workload(cfg, gen, seed=17)
    K, R, t = wl["rig"]
    reps = (F + gen - 1) // gen
    kp = torch.from_numpy(wl["kpts"]).to(dev).repeat(reps, 1, 1, 1, 1)[:F].contiguous()
    npers = torch.from_numpy(wl["n_persons"]).to(dev).repeat(reps, 1)[:F].contiguous()
    # the unsharded launch of the whole batch on this rank's context
    bt = BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=np.float32)
    full = bt.run_torch(kp, npers)
    torch.cuda.synchronize(dev)
    full = {k: v.clone() for k, v in full.items()}
    bt.close()
    lo, hi, per = shard_bounds(F, world, rank)
    for compact in (False, True):
        st = ShardedTriangulator(K, R, t, wl["params"], pout_max=pout, device=0, chunks=chunks, reuse_buffers=not compact, compact=compact)
        for rep in range(2):       # twice: the second run reuses the workspace (slots, side stream) of the first
            out = st.run(kp[lo:hi], F, npers[lo:hi], strict=True)
            torch.cuda.synchronize(dev)
            if compact:
                pad = compact_to_padded(out, pout)
                expect(torch.equal(pad["xyzs"].view(torch.int32), full["xyzs"].view(torch.int32)), f"compact xyzs rep {rep}")
                expect(torch.equal(pad["pscore"].view(torch.int32), full["pscore"].view(torch.int32)), f"compact pscore rep {rep}")
                expect(int(out["persons"].shape[0]) == int(full["count"].clamp(0, pout).sum()), "compact person rows")
            else:
                expect(torch.equal(out["xyzs"].view(torch.int32), full["xyzs"].view(torch.int32)), f"xyzs rep {rep}")
                expect(torch.equal(out["pscore"].view(torch.int32), full["pscore"].view(torch.int32)), f"pscore rep {rep}")
            expect(torch.equal(out["count"], full["count"]), f"count compact={compact} rep {rep}")
            expect(torch.equal(out["flags"], full["flags"]), f"flags compact={compact} rep {rep}")
            expect(not bool(out["rank_status"].any()), "rank_status")
        if not compact:
            padded_bytes = st.last_gather_bytes
        else:
            if pout >= 6:      # (slots the frames do not fill: 8 x 4 resolves to ~5 persons)
                expect(st.last_gather_bytes < padded_bytes, f"compact gather {st.last_gather_bytes} B vs padded {padded_bytes} B")
        st.bt.close()
    # row N1 on the sharded track: the block of this rank filtered with the carry exchange == the whole track filtered at once
    count1 = bool((full["count"] >= 1).all())
    track = full["xyzs"][:, 0, :, :3].to(torch.float64).contiguous()          # first person of every frame [F, kn, 3]
    want = torch.from_numpy(sm.smooth_track(track.cpu().numpy()[:, None], f=2.5, z=0.75, r=0.5, delta_time=1 / 30))[:, 0].to(dev)
    got = smooth_track_sharded(track[lo:hi].contiguous(), f=2.5, z=0.75, r=0.5, delta_time=1 / 30, F_total=F)
    torch.cuda.synchronize(dev)
    if hi > lo:
        err = float((got - want[lo:hi]).abs().max())
        expect(err < 1e-10 or not count1, f"sharded smoothing differs by {err}")
    # the whole chain on the shard, gather last (ShardedTrackPipeline) == TrackPipeline on the whole batch (fixed person count only)
    if cfg == 2:
        from snowmocap_amd.blender import CONTROL_POINT_NAMES
        from snowmocap_amd.pipeline import ShardedTrackPipeline, TrackPipeline
        th = dict(synth.default_thresholds(), **wl["params"])
        smo = {nm: [2.0 + 0.1 * i, 0.75, 0.1 * (i % 3)] for i, nm in enumerate(CONTROL_POINT_NAMES)}
        one = TrackPipeline(K, R, t, th, smo, n_persons_out=1)
        ref = one.run(kp, npers)
        torch.cuda.synchronize(dev)
        ref = {k: v.clone() for k, v in ref.items()}
        one.close()
        sp = ShardedTrackPipeline(K, R, t, th, smo, n_persons_out=1, device=0)
        got = sp.run(kp[lo:hi], F, npers[lo:hi])
        torch.cuda.synchronize(dev)
        expect(tuple(got["points_smoothed"].shape) == (F, 1, 24, 4), "sharded pipeline: shape of the gathered track")
        expect(torch.equal(got["valid"], ref["valid"]), "sharded pipeline: valid")
        if hi > lo:
            e1 = float((got["smoothed_local"][..., :3] - ref["smoothed"][lo:hi][..., :3]).abs().max())
            expect(e1 < 1e-10, f"sharded pipeline: N1 differs by {e1}")
        ok_rows = torch.isfinite(ref["points_smoothed"])
        e2 = float((got["points_smoothed"][ok_rows] - ref["points_smoothed"][ok_rows]).abs().max())
        expect(e2 < 1e-9, f"sharded pipeline: animation track differs by {e2}")
        expect(got["gather_bytes"] < 0.5 * got["gather_bytes_joint_track"], "sharded pipeline: gathers the control points, not the joints")
        sp.close()
    np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([ok]))
    if notes:
        with open(os.path.join(tmp, f"notes{rank}.txt"), "w") as fh:
            fh.write("\n".join(notes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("case", [
    (2, 1003, 1, 3),      # 4 x 1: blocks of 502 / 251 frames, the last one short; three pieces per block
    (2, 5, 1, 2),         # five frames: with four ranks the blocks are 2 / 2 / 1 / 0 -- an EMPTY trailing block
    (3, 601, 6, 2),       # 8 x 4 through the streaming route, ragged person counts, six slots
    (3, 3, 16, 1),        # three frames of 8 x 4 over four ranks: one frame each, one rank empty
], ids=["cfg2-1003", "cfg2-5", "cfg3-601", "cfg3-3"])
def test_sharded_path_with_real_kernels_on_several_ranks(tmp_path, world, case):
    import torch.multiprocessing as mp
    port = 24000 + (os.getpid() * 13 + world * 101 + case[1]) % 4000
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        note = tmp_path / f"notes{r}.txt"
        assert np.load(tmp_path / f"ok{r}.npy").all(), f"rank {r}: " + (note.read_text() if note.exists() else "?")


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] and configs[4] AT THEIR STATED SIZE (round-5 review, item 1): 1 000 000 frames of 4 x 1 and 100 000
# frames of 16 x 8 over EIGHT ranks.  The build box has one GPU, so the eight processes share cuda:0 (gloo group, collectives
# staged through page-locked host memory: sharded.all_gather_flat) -- the shard SIZES, the piece / slot / side-stream logic,
# the counts and the bytes are the real ones; xGMI is not (SURVEY 8e stays "unmeasured on hardware").  Every rank generates
# only its own block on the device from a per-block seed (synth.config_workload_device: nobody holds the 6.4 / 20.4 GB batch);
# rank 0 regenerates block after block and recomputes it UNSHARDED (its own context, other launch shapes): the gathered
# track must equal that bit for bit; every rank checks frames of its block against the CPU oracle (64 frames per config);
# rank_status and the fall-back counters must be 0.  The wall time of ShardedTriangulator.run per rank is recorded
# (gpurun_out/multiproc_full.jsonl when that directory exists): what the emulated gather costs beside the kernels.

def _full_worker(rank, world, port, case, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import json
    import time
    import torch
    import torch.distributed as dist
    from snowmocap_amd import synth, _lib
    from snowmocap_amd.batch import BatchTriangulator
    from snowmocap_amd.sharded import ShardedTriangulator, shard_bounds
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ok, notes, rec = True, [], {}

    def expect(cond, what):
        nonlocal ok
        if not cond:
            ok = False
            notes.append(what)

    name, cfg, F, pout, chunks, gathers = case
    lo, hi, per = shard_bounds(F, world, rank)
    gen = lambda q: synth.config_workload_device(cfg, shard_bounds(F, world, q)[1] - shard_bounds(F, world, q)[0], 77000 + 100 * cfg + q, dev)
    wl = gen(rank)
    K, R, t = wl["rig"]
    kp, npers = wl["kpts"], wl["n_persons"]
    multi = kp.shape[2] > 1
    results = {}
    for compact in gathers:
        st = ShardedTriangulator(K, R, t, wl["params"], pout_max=pout, device=0, chunks=chunks, compact=compact)
        dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = st.run(kp, F, npers, strict=True)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        expect(not bool(out["rank_status"].any()), f"rank_status compact={compact}")
        if multi:
            # (last segment of the last piece.)  No frame may leave the streaming route -- [0] second association launch,
            # [2] k_frame_recompute.  [1] counts frames with a candidate whose mean lay within 1e-6 of average_score_threshold
            # and was re-done exactly (the numerics contract): on 12 500 FRESH frames x 7 680 candidates a handful do -- the
            # tiled 250-frame batches of the other tests never met one -- so it is recorded and bounded, not required to be 0
            sc = st.bt.ctx.last_stream_counts()
            expect(sc[0] == 0 and sc[2] == 0 and 0 <= sc[1] <= 16, f"fall-back counters {sc} compact={compact}")
            rec.setdefault("exactly_resummed_frames_last_segment", {})["compact" if compact else "padded"] = sc[1]
        # kernels alone, same pieces, no gather: what the rank computes while it waits for the exchange
        t1 = time.perf_counter()
        st.run(kp, F, npers, gather=False)
        torch.cuda.synchronize(dev)
        rec["compact" if compact else "padded"] = dict(run_wall_s=wall, kernels_only_wall_s=time.perf_counter() - t1, pieces=st.last_chunks,
                                                       gather_bytes_received=int(st.last_gather_bytes))
        results[compact] = out
        st.bt.close()
    full = results.get(False)
    comp = results.get(True)
    # every rank: frames of ITS block against the CPU oracle (8 ranks x 8 frames = 64 per config), through whichever gather ran
    if hi > lo:
        rng = np.random.default_rng(rank)
        pick = np.sort(rng.choice(hi - lo, size=min(8, hi - lo), replace=False))
        kph = kp[torch.from_numpy(pick).to(dev)].cpu().numpy()
        ref = orc.triangulate_condense_batch(K, R, t, kph, npers[: len(pick)].cpu().numpy(), orc.make_params(**wl["params"]), pout)
        for i, fl in enumerate(pick):
            f = lo + int(fl)
            m = min(int(ref["count"][i]), pout)
            if full is not None:
                expect(int(full["count"][f]) == int(ref["count"][i]), f"oracle count frame {f}")
                got = full["xyzs"][f, :m].cpu().numpy().astype(np.float64)
            else:
                expect(int(comp["count"][f]) == int(ref["count"][i]), f"oracle count frame {f}")
                o = int(comp["offsets"][f])
                got = comp["persons"][o:o + m].cpu().numpy().astype(np.float64)
            err = np.abs(got[..., :3] - ref["xyz"][i, :m]).max(initial=0.0)
            expect(err < 1e-5, f"oracle frame {f}: {err} m")
            rel = (np.abs(got[..., 3] - ref["kscore"][i, :m]) / np.maximum(1e-30, np.abs(ref["kscore"][i, :m]))).max(initial=0.0)
            expect(rel < 1e-5, f"oracle scores frame {f}: {rel}")
    # rank 0: the whole batch again, UNSHARDED -- block after block regenerated from its seed, one launch per block (the sharded
    # run cut it into `pieces`), on a context of its own: bit for bit what was gathered
    if rank == 0:
        bt = BatchTriangulator(K, R, t, wl["params"], pout_max=pout, out_dtype=np.float32)
        i32 = torch.int32
        slot = torch.arange(pout, device=dev)[None, :]
        for q in range(world):
            qlo, qhi, _ = shard_bounds(F, world, q)
            if qhi == qlo:
                continue
            w = wl if q == 0 else gen(q)
            one = bt.run_torch(w["kpts"], w["n_persons"])
            torch.cuda.synchronize(dev)
            if full is not None:
                for k in ("xyzs", "pscore"):
                    expect(torch.equal(full[k][qlo:qhi].view(i32), one[k].view(i32)), f"padded gather, block {q}: {k}")
                expect(torch.equal(full["count"][qlo:qhi], one["count"]) and torch.equal(full["flags"][qlo:qhi], one["flags"]), f"padded gather, block {q}: count / flags")
            if comp is not None:
                expect(torch.equal(comp["count"][qlo:qhi], one["count"]) and torch.equal(comp["flags"][qlo:qhi], one["flags"]), f"compact gather, block {q}: count / flags")
                mask = slot < one["count"].clamp(0, pout)[:, None]
                rows = (comp["offsets"][qlo:qhi, None] + slot)[mask]
                expect(torch.equal(comp["persons"][rows].view(i32), one["xyzs"][mask].view(i32)), f"compact gather, block {q}: persons")
                expect(torch.equal(comp["pscore"][rows].view(i32), one["pscore"][mask].view(i32)), f"compact gather, block {q}: pscore")
            if not multi:
                expect(bool(((one["flags"] & _lib.FLAG_FASTPATH) != 0).all()), f"block {q}: frames off the fast path")
            del one, w
        bt.close()
        rec["persons_per_frame"] = float((full if full is not None else comp)["count"].float().mean())
    rec.update(rank=rank, world=world, config=name, frames_total=F, frames_of_rank=hi - lo, pout_max=pout)
    with open(os.path.join(tmp, f"rec{rank}.json"), "w") as fh:
        json.dump(rec, fh)
    np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([ok]))
    if notes:
        with open(os.path.join(tmp, f"notes{rank}.txt"), "w") as fh:
            fh.write("\n".join(notes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [
    ("BASELINE configs[3]: 4 x 1, 1 000 000 frames over 8 ranks", 2, 1000000, 1, "auto", (False,)),
    ("BASELINE configs[4]: 16 x 8, 100 000 frames over 8 ranks", 5, 100000, 32, 8, (False, True)),
], ids=["configs3-1M-4x1", "configs4-100k-16x8"])
def test_baseline_configs_3_and_4_at_full_size_over_eight_ranks(tmp_path, case):
    import json
    import torch.multiprocessing as mp
    world = 8
    port = 28000 + (os.getpid() * 7 + case[2]) % 3000
    mp.spawn(_full_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    recs = []
    for r in range(world):
        note = tmp_path / f"notes{r}.txt"
        assert np.load(tmp_path / f"ok{r}.npy").all(), f"rank {r}: " + (note.read_text() if note.exists() else "?")
        recs.append(json.loads((tmp_path / f"rec{r}.json").read_text()))
    line = dict(config=case[0], world=world, ranks=recs)
    print(json.dumps(line))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "multiproc_full.jsonl"), "a") as fh:
            fh.write(json.dumps(line) + "\n")


# ---------------------------------------------------------------------------------------------------------------------
# ShardedTrackPipeline with a person count that varies from frame to frame (round-5 review, item 4): the reference's list-index
# semantics -- banks of frame 0, zip truncation -- on a frame shard: ONE exchange of count[0] with the ranks' error bits, then per
# slot the carry / hold exchanges on the frames that carry the slot.  Four processes on the GPU: (a) fixture G9 -- the reference's
# own main.py loop on an 8-camera sequence with 0 ... 7 persons per frame -- must come out as the reference's JSON; (b) a
# BASELINE configs[2] batch (8 x 4, ~5 persons per frame with ghosts) must equal TrackPipeline.run(ragged="reference") on one GPU.

def _ragged_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import json
    import torch
    import torch.distributed as dist
    from conftest import GOLDEN
    from snowmocap_amd import synth
    from snowmocap_amd.blender import CONTROL_POINT_NAMES
    from snowmocap_amd.pipeline import ShardedTrackPipeline, TrackPipeline
    from snowmocap_amd.sharded import shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ok, notes = True, []

    def expect(cond, what):
        nonlocal ok
        if not cond:
            ok = False
            notes.append(what)

    def both(K, R, t, th, smo, slots, kp, npers, tol):
        F = int(kp.shape[0])
        lo, hi, _ = shard_bounds(F, world, rank)
        one = TrackPipeline(K, R, t, th, smo, n_persons_out=slots)
        ref = one.run(kp, npers)
        torch.cuda.synchronize(dev)
        ref = {k: v.clone() for k, v in ref.items()}
        one.close()
        sp = ShardedTrackPipeline(K, R, t, th, smo, n_persons_out=slots, device=0)
        got = sp.run(kp[lo:hi].contiguous(), F, npers[lo:hi].contiguous())
        torch.cuda.synchronize(dev)
        expect(torch.equal(got["tracked"], ref["tracked"]), "tracked")
        expect(torch.equal(got["valid"], ref["valid"]), "valid")
        expect(got["count0"] == int(ref["count"][0]), "count0")
        live = (torch.arange(slots, device=dev)[None, :] < ref["tracked"][:, None])
        fin = torch.isfinite(ref["points_smoothed"]) & live[:, :, None, None]
        e = float((got["points_smoothed"][fin] - ref["points_smoothed"][fin]).abs().max()) if bool(fin.any()) else 0.0
        expect(e < tol, f"animation track differs by {e}")
        expect(torch.equal(torch.isnan(got["points_smoothed"]), torch.isnan(ref["points_smoothed"])), "NaN pattern of the animation track")
        expect(not bool(got["points_smoothed"][~live].any()), "slots behind tracked[f] must stay zero")
        if hi > lo:
            e1 = float((got["smoothed_local"][..., :3] - ref["smoothed"][lo:hi][..., :3]).abs().max())
            expect(e1 < tol, f"N1 of the block differs by {e1}")
        sp.close()
        return got, ref

    # (a) fixture G9: the reference's JSON
    z = np.load(f"{GOLDEN}/g9_pipeline_multi.npz")
    th, arm, smo = json.loads(str(z["thresholds"])), json.loads(str(z["armature"])), json.loads(str(z["smooth"]))
    want = json.loads(str(z["result"]))
    kp = torch.from_numpy(z["kpts"]).to(dev)
    npers = torch.from_numpy(z["n_persons"].astype(np.int32)).to(dev)
    got, _ = both(z["K"], z["R"], z["t"], th, smo, 8, kp, npers, 1e-9)
    expect(np.array_equal(got["tracked"].cpu().numpy(), z["tracked"]), "G9: tracked")
    res = TrackPipeline.to_blender_result(got["points_smoothed"], got["valid"], arm, got["tracked"])
    expect(len(res) == len(want), "G9: frames")
    worst = 0.0
    for f, (g, w) in enumerate(zip(res, want)):
        expect(len(g["armature"]) == len(w["armature"]) == int(z["tracked"][f]) and g["score"] == w["score"], f"G9 frame {f}: persons / scores")
        for ga, wa in zip(g["armature"], w["armature"]):
            for name, vec in wa.items():
                worst = max(worst, float(np.abs(np.asarray(ga[name]) - np.asarray(vec)).max()))
    expect(worst < 1e-8, f"G9: the reference's JSON differs by {worst}")
    # (b) a BASELINE configs[2] batch: ~5 persons per frame, the count varies
    wl = synth.config_workload(3, 300, seed=11)
    K, R, t = wl["rig"]
    th2 = dict(synth.default_thresholds(), **wl["params"])
    smo2 = {nm: [1.5 + 0.1 * i, 0.75, 0.1 * (i % 3)] for i, nm in enumerate(CONTROL_POINT_NAMES)}
    kp2 = torch.from_numpy(wl["kpts"]).to(dev).repeat(5, 1, 1, 1, 1)[:1403].contiguous()      # 1 403 frames: uneven blocks
    np2 = torch.from_numpy(wl["n_persons"]).to(dev).repeat(5, 1)[:1403].contiguous()
    got2, ref2 = both(K, R, t, th2, smo2, 10, kp2, np2, 1e-9)
    expect(len(set(ref2["count"].cpu().numpy().tolist())) > 1, "configs[2] batch: the count should vary")
    expect(got2["gather_bytes"] < 0.5 * got2["gather_bytes_joint_track"], "gathers the control points, not the joints")
    # ragged="refuse" raises on EVERY rank (the bits travel with count[0]): nobody hangs
    sp = ShardedTrackPipeline(K, R, t, th2, smo2, n_persons_out=10, device=0)
    lo, hi, _ = shard_bounds(1403, world, rank)
    try:
        sp.run(kp2[lo:hi].contiguous(), 1403, np2[lo:hi].contiguous(), ragged="refuse")
        expect(False, "ragged=refuse did not raise")
    except ValueError:
        pass
    sp.close()
    np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([ok]))
    if notes:
        with open(os.path.join(tmp, f"notes{rank}.txt"), "w") as fh:
            fh.write("\n".join(notes))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_pipeline_with_varying_person_counts_on_four_ranks(tmp_path):
    import torch.multiprocessing as mp
    world = 4
    port = 26000 + (os.getpid() * 11) % 1500
    mp.spawn(_ragged_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        note = tmp_path / f"notes{r}.txt"
        assert np.load(tmp_path / f"ok{r}.npy").all(), f"rank {r}: " + (note.read_text() if note.exists() else "?")

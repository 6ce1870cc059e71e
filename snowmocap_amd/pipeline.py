"""Additive whole-recording form of main.py:47-106: every stage after 2D detection on the GPU, device-resident.

    [undistort raw-frame keypoints (N4)] -> triangulate + condense (A1-A4, the hot path) -> temporal smoothing (N1)
    -> Blender control points (N2) -> per-bone smoothing (N2) -> the reference's JSON track

One call per recording instead of one Python iteration per frame.  The per-frame protocol of the reference
identifies persons by list index and lets the list length vary from frame to frame: its filter banks are those of
frame 0, later frames are zip-truncated against them (triangulation.py:169-171, blender.py:152-166).  `run` follows
that: n_persons_out is the number of SLOTS, frame f carries min(count[f], count[0]) persons (fixture G9).  PyTorch is
used for device memory and the stream only.
"""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _lib
from .batch import BatchTriangulator
from .blender import CONTROL_POINT_NAMES, _WIDTH


class TrackPipeline:
    def __init__(self, K, R, t, thresholds, blender_smooth_profile, n_persons_out=1, D=None, device=0,
                 method=_lib.PAIRWISE):
        """thresholds: the dict of configs/snowmocap_default_config.json (8 triangulation keys + smooth_f, smooth_z,
        smooth_r, smooth_delta_time); blender_smooth_profile: {control point: [f, z, r]}."""
        self.th = dict(thresholds)
        self.P = int(n_persons_out)
        self.bt = BatchTriangulator(K, R, t, {k: self.th[k] for k in (
            "keypoint_score_threshold", "average_score_threshold", "distance_threshold", "condense_distance_tol",
            "condense_person_num_tol", "condense_score_tol", "center_point_index", "keypoint_num")},
            pout_max=self.P, out_dtype=np.float64, device=device, method=method, D=D)
        self.kn = self.bt.params.keypoint_num
        self.fzr = np.ascontiguousarray([blender_smooth_profile[n] for n in CONTROL_POINT_NAMES], dtype=np.float64)
        self.device = device

    def close(self):
        self.bt.close()

    def run(self, kpts, n_persons=None, check=True, ragged="reference"):
        """kpts [F, C, Pmax, J, 3] (NumPy or CUDA tensor; raw-frame pixels if D was given) ->
        dict of CUDA tensors: xyzs [F, P, kn, 4] (triangulated), smoothed [F, P, kn, 4], points [F, P, 24, 4],
        valid [F, P, 24], points_smoothed [F, P, 24, 4], count [F], flags [F], tracked [F] (P = n_persons_out slots).

        Person counts that vary from frame to frame follow the reference (ragged="reference"): its filter banks are those of
        frame 0 and `zip` matches a frame's persons to them BY LIST INDEX (triangulation.py:169-171, blender.py:152-166), so
        frame f carries tracked[f] = min(count[f], count[0]) persons, the persons behind that are dropped, and a bank whose
        person is missing in a frame is not stepped in that frame (its time stands still).  Per slot i this is the plain
        filter over the frames with tracked > i: those are gathered, filtered by the same kernels and scattered back;
        slots at or behind tracked[f] hold zeros.  Frame 0 must fit the slots (count[0] <= n_persons_out).
        ragged="refuse": raise unless every frame resolves to exactly n_persons_out persons (round 4's behaviour).
        check=False skips the host read of counts and flags and treats every slot of every frame as tracked."""
        import torch
        dev = torch.device("cuda", self.device)
        if not torch.is_tensor(kpts):
            kpts = torch.from_numpy(np.ascontiguousarray(kpts)).to(dev)
        if n_persons is not None and not torch.is_tensor(n_persons):
            n_persons = torch.from_numpy(np.ascontiguousarray(n_persons, dtype=np.int32)).to(dev)
        F = kpts.shape[0]
        L, h = self.bt.ctx.L, self.bt.ctx.handle
        st = ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        tri = self.bt.run_torch(kpts, n_persons)
        tracked = None                                   # None: every slot of every frame
        if check:
            cnt = tri["count"].cpu().numpy()
            flg = tri["flags"].cpu().numpy()
            if (flg & _lib.FLAG_SINGULAR).any():    # the reference's np.linalg.inv raises (triangulation.py:26)
                raise np.linalg.LinAlgError(f"Singular matrix (frame {int(np.argmax((flg & _lib.FLAG_SINGULAR) != 0))})")
            if not (cnt == self.P).all():
                if ragged == "refuse":
                    bad = int(np.argmax(cnt != self.P))
                    raise ValueError(f"frame {bad} resolved to {int(cnt[bad])} persons, the track is built for {self.P}")
                if F and int(cnt[0]) > self.P:
                    raise ValueError(f"frame 0 resolved to {int(cnt[0])} persons, the track has {self.P} slots (n_persons_out)")
                tracked = np.minimum(cnt, int(cnt[0]) if F else 0).astype(np.int32)
        xyzs = tri["xyzs"]
        th = self.th
        fzrd = (float(th["smooth_f"]), float(th["smooth_z"]), float(th["smooth_r"]), float(th["smooth_delta_time"]))
        pts = torch.empty((F, self.P, 24, 4), dtype=torch.float64, device=dev)
        val = torch.empty((F, self.P, 24), dtype=torch.uint8, device=dev)

        def blender_points(x, n, p_out, v_out):
            _lib.check(L.snowtri_blender_points(h, n, self.kn, ct.c_void_p(x.data_ptr()), _lib.F64, ct.c_void_p(p_out.data_ptr()),
                                                ct.c_void_p(v_out.data_ptr()), _lib.DEVICE, st), "snowtri_blender_points")

        if tracked is None:
            sm = torch.empty_like(xyzs)                    # only the points are filtered, the scores copied (triangulation.py:169-184)
            _lib.check(L.snowtri_smooth_joint_track(h, F, self.P * self.kn, ct.c_void_p(xyzs.data_ptr()), *fzrd, ct.c_void_p(sm.data_ptr()),
                                                    _lib.DEVICE, st), "snowtri_smooth_joint_track")
            blender_points(sm, F * self.P, pts, val)
            pts_s = torch.empty_like(pts)
            _lib.check(L.snowtri_blender_smooth(h, F, self.P, ct.c_void_p(pts.data_ptr()), ct.c_void_p(val.data_ptr()),
                                                _lib.ptr(self.fzr), fzrd[3], ct.c_void_p(pts_s.data_ptr()), _lib.DEVICE, st),
                       "snowtri_blender_smooth")
            trk = torch.full((F,), self.P, dtype=torch.int32, device=dev)
        else:
            # slot by slot over the frames that carry it (frame 0 among them: it seeds the slot's filters)
            trk = torch.from_numpy(tracked).to(dev)
            sm = torch.zeros_like(xyzs)
            pts.zero_()
            val.zero_()
            pts_s = torch.zeros_like(pts)
            for i in range(int(tracked[0]) if F else 0):
                idx = torch.nonzero(trk > i).view(-1)
                T = int(idx.numel())
                xi = xyzs[idx, i].contiguous()                                  # [T, kn, 4]
                si = torch.empty_like(xi)
                _lib.check(L.snowtri_smooth_joint_track(h, T, self.kn, ct.c_void_p(xi.data_ptr()), *fzrd, ct.c_void_p(si.data_ptr()),
                                                        _lib.DEVICE, st), "snowtri_smooth_joint_track")
                pi = torch.empty((T, 1, 24, 4), dtype=torch.float64, device=dev)
                vi = torch.empty((T, 1, 24), dtype=torch.uint8, device=dev)
                blender_points(si, T, pi, vi)
                qi = torch.empty_like(pi)
                _lib.check(L.snowtri_blender_smooth(h, T, 1, ct.c_void_p(pi.data_ptr()), ct.c_void_p(vi.data_ptr()),
                                                    _lib.ptr(self.fzr), fzrd[3], ct.c_void_p(qi.data_ptr()), _lib.DEVICE, st),
                           "snowtri_blender_smooth")
                sm[idx, i] = si
                pts[idx, i] = pi[:, 0]
                val[idx, i] = vi[:, 0]
                pts_s[idx, i] = qi[:, 0]
        return dict(xyzs=xyzs, smoothed=sm, points=pts, valid=val, points_smoothed=pts_s, count=tri["count"],
                    flags=tri["flags"], tracked=trk)

    @staticmethod
    def to_blender_result(points_smoothed, valid, armature_profile=None, tracked=None):
        """Device (or NumPy) track -> the list the reference dumps with save_blender_result (blender.py:180-187):
        one {'armature': [per person {name: list}], 'score': [per person {name: 0/1}]} per frame.  tracked [F]: persons
        frame f carries (run()'s "tracked"; default: every slot)."""
        pts = points_smoothed.cpu().numpy() if hasattr(points_smoothed, "cpu") else np.asarray(points_smoothed)
        val = valid.cpu().numpy() if hasattr(valid, "cpu") else np.asarray(valid)
        if tracked is None:
            trk = np.full(pts.shape[0], pts.shape[1], dtype=np.int64)
        else:
            trk = (tracked.cpu().numpy() if hasattr(tracked, "cpu") else np.asarray(tracked)).astype(np.int64)
        live = np.arange(pts.shape[1])[None, :] < trk[:, None]
        if not val[..., 1][live].all():
            raise np.linalg.LinAlgError("SVD did not converge")     # the reference raises on a NaN pelvis matrix
        names = list(armature_profile.keys()) if armature_profile is not None else list(CONTROL_POINT_NAMES)
        slot = {n: i for i, n in enumerate(CONTROL_POINT_NAMES)}
        frames = []
        for f in range(pts.shape[0]):
            frames.append({
                "armature": [{n: pts[f, p, slot[n], :_WIDTH[n]].tolist() for n in names} for p in range(int(trk[f]))],
                "score": [{n: int(val[f, p, slot[n]]) for n in names} for p in range(int(trk[f]))]})
        return frames


class ShardedTrackPipeline:
    """TrackPipeline on a FRAME SHARD (one instance per rank, one process per GPU): the whole chain after 2D detection runs on
    the rank's contiguous frame block and only the final animation track is gathered (round-4 review, item 4b):

        A1-A4 on the block  ->  N1 with the carry exchange (4n + 1 doubles per rank, sharded.smooth_track_sharded)
        ->  N2 control points (per frame)  ->  N2 per-bone filters with the hold exchange + the carry exchange
        (sharded.blender_smooth_sharded)  ->  ONE all-gather of [F, P, 24, 4] float64 + valid [F, P, 24] + tracked [F]

    instead of gathering the [F, P, kn, 4] joint track first (at roofline speed that gather is ~8x the kernel):
    792 + 24 bytes per person and frame against 2 128 for 133 float32 joints.
    A person count that varies from frame to frame follows the reference, as TrackPipeline.run(ragged="reference") does on one
    GPU (round-5 review, item 4): the filter banks are those of frame 0 and `zip` matches a frame's persons to them by list
    index (triangulation.py:169-171, blender.py:152-166), so frame f carries tracked[f] = min(count[f], count[0]) persons.
    That needs ONE number of another rank -- count[0], handed round with the ranks' error bits (sharded.tracked_counts) --
    and per slot i the frames with tracked > i that a rank holds are a block of the slot's own frame sequence: the N1 carry
    exchange and the N2 hold + carry exchanges run on those blocks unchanged (an empty block is a rank none of whose frames
    carries the slot; the block with frame 0 starts the sequence)."""

    def __init__(self, K, R, t, thresholds, blender_smooth_profile, n_persons_out=1, device=0, group=None, method=_lib.PAIRWISE):
        self.pipe = TrackPipeline(K, R, t, thresholds, blender_smooth_profile, n_persons_out=n_persons_out, device=device, method=method)
        self.group = group
        self.device = device

    def close(self):
        self.pipe.close()

    def run(self, kpts_local, F_total, n_persons_local=None, gather=True, ragged="reference"):
        """kpts_local [T_r, C, Pmax, J, 3] CUDA tensor: the block shard_bounds(F_total, world, rank) gives this rank.  Returns
        points_smoothed / valid / tracked of the WHOLE track on every rank (gather=True) or of the block, plus the block's own
        stages.  ragged="refuse": every frame of every rank must resolve to exactly n_persons_out persons (round 4's contract)."""
        import torch
        import torch.distributed as dist
        from .sharded import (blender_smooth_sharded, gather_track_chunked, shard_bounds, smooth_track_sharded, tracked_counts)
        p = self.pipe
        dev = kpts_local.device
        rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        lo, hi, per = shard_bounds(int(F_total), world, rank)
        T = int(kpts_local.shape[0])
        if T != hi - lo:
            raise ValueError(f"ShardedTrackPipeline: rank {rank} holds {T} frames, shard_bounds gives it [{lo}, {hi})")
        L, h = p.bt.ctx.L, p.bt.ctx.handle
        st = ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        P, kn, th = p.P, p.kn, p.th
        if T:
            tri = p.bt.run_torch(kpts_local, n_persons_local)
        else:
            tri = p.bt.alloc_outputs(0, dev)
        # every rank learns count[0] and whether every block is usable BEFORE the exchanges below (nobody is left waiting in one)
        bits = torch.zeros((), dtype=torch.int32, device=dev)
        if T:
            bits = (((tri["flags"] & _lib.FLAG_SINGULAR) != 0).any().to(torch.int32) + 2 * (tri["count"] != P).any().to(torch.int32)).to(torch.int32)
        tracked, count0, bits = tracked_counts(tri["count"], int(F_total), P, group=self.group, flag_bits=bits)
        if bits & 1:
            raise np.linalg.LinAlgError("Singular matrix (a frame of some rank)")     # the reference's np.linalg.inv raises (triangulation.py:26)
        if (bits & 2) and ragged == "refuse":
            raise ValueError(f"a frame of some rank did not resolve to {P} persons (ragged=\"refuse\")")
        xyzs = tri["xyzs"]
        fkw = dict(f=th["smooth_f"], z=th["smooth_z"], r=th["smooth_r"], delta_time=th["smooth_delta_time"], group=self.group, ctx=p.bt.ctx)
        pts = torch.empty((T, P, 24, 4), dtype=torch.float64, device=dev)
        val = torch.empty((T, P, 24), dtype=torch.uint8, device=dev)

        def blender_points(x, n_, p_out, v_out):
            _lib.check(L.snowtri_blender_points(h, n_, kn, ct.c_void_p(x.data_ptr()), _lib.F64, ct.c_void_p(p_out.data_ptr()),
                                                ct.c_void_p(v_out.data_ptr()), _lib.DEVICE, st), "snowtri_blender_points")

        if not (bits & 2):
            # every frame of every rank carries all P slots: the whole block at once
            sm = smooth_track_sharded(xyzs.view(T, P * kn * 4) if T else xyzs.reshape(0, P * kn * 4), F_total=F_total, **fkw).view(T, P, kn, 4)
            if T:
                sm[..., 3] = xyzs[..., 3]                  # only the points are filtered (triangulation.py:169-184)
                blender_points(sm, T * P, pts, val)
            pts_s = blender_smooth_sharded(pts, val, p.fzr, th["smooth_delta_time"], group=self.group, ctx=p.bt.ctx, F_total=F_total)
        else:
            # slot by slot over the frames that carry it; every rank walks the SAME count0 slots (the exchanges are collectives)
            sm = torch.zeros_like(xyzs)
            pts.zero_()
            val.zero_()
            pts_s = torch.zeros_like(pts)
            holds0 = lo == 0 and hi > 0                    # frame 0 carries every tracked slot: its rank starts every slot's sequence
            for i in range(count0):
                idx = torch.nonzero(tracked > i).view(-1)
                Ti = int(idx.numel())
                xi = xyzs[idx, i].contiguous().view(Ti, kn * 4)
                si = smooth_track_sharded(xi, first=holds0 and Ti > 0, **fkw).view(Ti, kn, 4)
                pi = torch.empty((Ti, 1, 24, 4), dtype=torch.float64, device=dev)
                vi = torch.empty((Ti, 1, 24), dtype=torch.uint8, device=dev)
                if Ti:
                    si[..., 3] = xi.view(Ti, kn, 4)[..., 3]
                    blender_points(si, Ti, pi, vi)
                qi = blender_smooth_sharded(pi, vi, p.fzr, th["smooth_delta_time"], group=self.group, ctx=p.bt.ctx, first=holds0 and Ti > 0)
                if Ti:
                    sm[idx, i] = si
                    pts[idx, i] = pi[:, 0]
                    val[idx, i] = vi[:, 0]
                    pts_s[idx, i] = qi[:, 0]
        out = dict(xyzs_local=xyzs, smoothed_local=sm, points_local=pts, valid_local=val, points_smoothed_local=pts_s,
                   count_local=tri["count"], flags_local=tri["flags"], tracked_local=tracked, count0=count0)
        if gather:
            def copy_block(lo_, hi_, views):
                views["points_smoothed"][: hi_ - lo_] = pts_s[lo_:hi_]
                views["valid"][: hi_ - lo_] = val[lo_:hi_]
                views["tracked"][: hi_ - lo_] = tracked[lo_:hi_]
            g = gather_track_chunked(copy_block, T, int(F_total), {"points_smoothed": ((P, 24, 4), torch.float64), "valid": ((P, 24), torch.uint8),
                                                                   "tracked": ((), torch.int32)},
                                     chunks=1, group=self.group, device=dev)
            out["points_smoothed"], out["valid"], out["tracked"] = g["points_smoothed"], g["valid"], g["tracked"]
            out["gather_bytes"] = world * per * (P * 24 * 4 * 8 + P * 24 + 4) + world * 16
            out["gather_bytes_joint_track"] = world * per * (P * kn * 16 + P * 4 + 8)
        return out

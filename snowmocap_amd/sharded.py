"""Frame sharding across the GPUs of one node + the single all-gather that reassembles the track.

Frames are independent through triangulate + condense (the reference reads only the current
frame's detections: triangulation.py:50-162; state is cleared per frame, main.py:106), so the batch
is split into contiguous frame blocks, one per rank, with NO data-path collective inside the
kernels; an all-gather (RCCL over xGMI when the backend is "nccl") hands every rank the whole
3D track, which is what the next stage -- temporal smoothing, a recurrence over frames
(triangulation.py:164-186) -- needs.  The shard is computed in a few pieces and the gather of a piece
(all of its outputs in one buffer, one collective) runs on a side stream under the next piece's kernel:
the gather moves 16 B/joint over xGMI against 64 B/joint over HBM at ~1/20 of the bandwidth, so it is
what bounds the end-to-end time.  Results are bit-identical to the single-GPU run.

The sharding / gather logic is backend-agnostic (tests run it on CPU with gloo, world_size 2).
"""
from __future__ import annotations


def shard_bounds(F, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; every block has ceil(F/world) frames except that
    trailing blocks are clipped at F (they may be shorter or empty)."""
    per = (F + world - 1) // world
    lo = min(F, rank * per)
    return lo, min(F, lo + per), per


def gather_track(local, F_total, group=None):
    """local: this rank's [n_local, ...] block (n_local <= per).  Returns the full [F_total, ...]
    tensor on every rank with ONE all_gather_into_tensor of equal, zero-padded blocks."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = (F_total + world - 1) // world
    if local.shape[0] != per:
        pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        local = pad
    full = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full[:F_total]


_HOST_STAGE = {}   # (device, world, numel, dtype) -> pinned (send, recv) pair of the host-staged collective
_HOST_STAGE_MAX_BYTES = 1 << 30   # pairs larger than this are not kept (a one-off gather of a whole track would pin its size for good)


def all_gather_flat(recv, send, group=None):
    """recv[world * n] <- send[n] of every rank, in rank order: ONE all_gather_into_tensor.

    CUDA tensors on a group whose backend has no device path (gloo: several ranks sharing ONE GPU -- RCCL refuses two ranks
    on a device -- or a debugging run) are STAGED through page-locked host buffers on the current stream: copy out,
    synchronise that stream, the host collective, copy back.  The host blocks for the exchange, the stream order around it
    is the one of the device path, so the side-stream / event / slot-reuse logic of gather_track_chunked and the device-side
    combine of smooth_track_sharded run unchanged under genuine multi-process interleaving (tests/test_gpu_multiproc.py)."""
    import torch
    import torch.distributed as dist
    if not recv.is_contiguous():
        # (both paths write THROUGH a flat view of recv: reshape(-1) of a non-contiguous tensor would be a temporary copy and the
        # gathered data would be dropped silently -- round-5 advice)
        raise ValueError("all_gather_flat: recv must be contiguous")
    if send.is_cuda and dist.get_backend(group) == "gloo":
        world = dist.get_world_size(group)
        key = (send.device.index, world, send.numel(), send.dtype)
        keep = world * send.numel() * send.element_size() <= _HOST_STAGE_MAX_BYTES
        ent = _HOST_STAGE.get(key)
        if ent is None:
            ent = (torch.empty(send.numel(), dtype=send.dtype).pin_memory(),
                   torch.empty(world * send.numel(), dtype=send.dtype).pin_memory())
            if keep:
                _HOST_STAGE[key] = ent
        hs, hr = ent[:2]
        if len(ent) > 2:
            ent[2].synchronize()  # the previous copy back out of `hr` (possibly on another stream) has finished
        st = torch.cuda.current_stream(send.device)
        hs.copy_(send.reshape(-1), non_blocking=True)
        st.synchronize()
        dist.all_gather_into_tensor(hr, hs, group=group)
        recv.view(-1).copy_(hr, non_blocking=True)
        back = torch.cuda.Event()
        back.record(st)
        if keep:
            _HOST_STAGE[key] = (hs, hr, back)
        else:
            back.synchronize()   # (the one-off pinned pair is freed when this returns)
        return
    dist.all_gather_into_tensor(recv.view(-1), send.contiguous().view(-1), group=group)   # (flat on both sides: every backend takes that)


def auto_chunks(frames_per_rank, min_piece_frames=32768, max_chunks=8):
    """Pieces a rank's shard is cut into for the overlapped gather: as many as give pieces of >= min_piece_frames frames
    (a piece costs a handful of launches and one collective, ~100 us of host time: measured 0.15 ms per 10 000-frame
    step in one piece against 0.48 ms in four), at most max_chunks; a small shard is gathered in one piece."""
    return int(max(1, min(max_chunks, frames_per_rank // max(1, min_piece_frames))))


def _align(n, a=16):
    return (n + a - 1) // a * a


def gather_track_chunked(compute_block, n_local, F_total, regions, chunks=4, group=None, device=None, workspace=None):
    """Frame-sharded results -> the whole track on every rank, gathered WHILE the shard is still being computed.

    The rank's block of `per = ceil(F_total / world)` frame slots is cut into `chunks` pieces.  For piece i,
    `compute_block(lo, hi, views)` fills `views[name][: hi - lo]` for the local frames [lo, hi) (launching
    asynchronously on the current stream); every output of the piece lives in ONE flat buffer (regions back to back,
    16-byte aligned), which is all-gathered with ONE collective on a side stream while piece i + 1 is computed, and
    unpacked into the full tensors there.  Buffers rotate over two slots; a slot is rewritten only after its gather
    has finished.  On CPU tensors (gloo tests) the same steps run in order without streams.

    regions: {name: (shape_tail, torch dtype)} per frame.  Returns {name: tensor [F_total, *shape_tail]} plus
    "rank_status" ([world] int32: non-zero for a rank that claimed more frames than its contiguous block -- that rank
    raises ValueError AFTER the last collective, so no rank is left waiting in one; on CPU tensors every rank raises).
    workspace: a dict the caller keeps between calls -- the gathered tensors, the two send / receive slots and the side
    stream are then allocated once and reused (steady state without allocations; the tensors returned by one call are
    overwritten by the next).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = (F_total + world - 1) // world
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if F_total <= 0:                                       # an empty track: nothing to compute, nothing to gather
        out = {name: torch.empty((0,) + tuple(tail), dtype=dt, device=dev) for name, (tail, dt) in regions.items()}
        out["rank_status"] = torch.zeros(world, dtype=torch.int32, device=dev)
        return out
    # A rank whose block does not fit (n_local > per) must not simply raise: the other ranks would enter the collectives
    # below and wait for it forever.  It takes part with an EMPTY block and a status word of 1 in every piece it sends
    # (the last 16 bytes of a piece's buffer), and raises after the last collective; the other ranks see the word in
    # `rank_status` (CPU tensors: they raise too; CUDA tensors: returned, not read back -- no forced synchronisation).
    bad_local = not 0 <= n_local <= per
    n_claimed = n_local
    if bad_local:
        n_local = 0
    chunks = max(1, min(int(chunks), max(1, per)))
    cs = (per + chunks - 1) // chunks                      # frame slots per piece (same on every rank)
    on_gpu = dev.type == "cuda"
    # flat layout of one piece
    offs, total = {}, 0
    for name, (tail, dt) in regions.items():
        nbytes = cs * int(torch.tensor([], dtype=dt).element_size())
        for d in tail:
            nbytes *= int(d)
        offs[name] = (total, nbytes)
        total = _align(total + nbytes)
    status_off = total
    total += 16
    key = (world, per, cs, total, str(dev), tuple((n, tuple(t), str(d)) for n, (t, d) in regions.items()))
    ws = workspace if workspace is not None else {}
    if ws.get("key") != key:
        ws.clear()
        ws["key"] = key
        ws["full"] = {name: torch.empty((world * per,) + tuple(tail), dtype=dt, device=dev) for name, (tail, dt) in regions.items()}
        ws["send"] = [torch.zeros(total, dtype=torch.uint8, device=dev) for _ in range(2)]
        ws["recv"] = [torch.empty(world * total, dtype=torch.uint8, device=dev) for _ in range(2)]
        ws["side"] = torch.cuda.Stream(device=dev) if on_gpu else None
        ws["status"] = torch.zeros(world, dtype=torch.int32, device=dev)
    full, send, recv, status = ws["full"], ws["send"], ws["recv"], ws["status"]
    status.zero_()

    def views_of(flat):
        return {name: flat[o:o + nb].view(regions[name][1]).view((cs,) + tuple(regions[name][0]))
                for name, (o, nb) in offs.items()}

    if on_gpu:
        main = torch.cuda.current_stream(dev)
        side = ws["side"]
        side.wait_stream(main)                           # a reused workspace: earlier readers of `full` / the slots come first
        gathered = [None, None]
    for i in range((per + cs - 1) // cs):
        slot = i & 1
        lo = min(n_local, i * cs)
        hi = min(n_local, (i + 1) * cs)
        if on_gpu and gathered[slot] is not None:
            main.wait_event(gathered[slot])              # the gather that last read this slot is done
        if hi - lo < cs:
            send[slot].zero_()                           # short or empty piece: deterministic padding
        if bad_local:                                    # (0 otherwise: the slots are allocated zeroed and no region covers the word)
            send[slot][status_off:status_off + 4].view(torch.int32).fill_(1)
        if hi > lo:
            compute_block(lo, hi, views_of(send[slot]))
        width = min(cs, per - i * cs)                    # frame slots of this piece that exist in the block

        def gather_and_unpack():
            all_gather_flat(recv[slot], send[slot], group=group)
            got = recv[slot].view(world, total)
            torch.maximum(status, got[:, status_off:status_off + 4].view(torch.int32).view(world), out=status)
            for name, (o, nb) in offs.items():
                tail, dt = regions[name]
                src = got[:, o:o + nb].view(dt).view((world, cs) + tuple(tail))
                full[name].view((world, per) + tuple(tail))[:, i * cs:i * cs + width] = src[:, :width]

        if on_gpu:
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                gather_and_unpack()
                done = torch.cuda.Event()
                done.record(side)
            gathered[slot] = done
        else:
            gather_and_unpack()
    if on_gpu:
        main.wait_stream(side)                           # results are ready for whatever the caller queues next
    refusal = (f"gather_track_chunked: a rank holds {n_claimed} frames but contiguous blocks of "
               f"ceil({F_total} / {world}) = {per} frames are what is gathered (use shard_bounds)")
    if bad_local:
        for sl in send:                                  # (a caller-kept workspace may be used again)
            sl[status_off:status_off + 4].zero_()
        raise ValueError(refusal)
    if not on_gpu and bool(status.any()):
        bad = [q for q in range(world) if int(status[q])]
        raise ValueError(f"gather_track_chunked: rank(s) {bad} hold more frames than their contiguous block of "
                         f"ceil({F_total} / {world}) = {per} (use shard_bounds); their blocks were gathered as zeros")
    out = {name: t[:F_total] for name, t in full.items()}
    out["rank_status"] = status                          # [world] int32, 0 = that rank's block is valid
    return out


def gather_track_compact(compute_block, n_local, F_total, kn, pout_max, chunks=4, group=None, device=None):
    """Frame-sharded results -> the whole track on every rank as PACKED persons ("or gather compacted persons +
    counts"): per piece ONE all-gather of (count, flags) -- 8 bytes per frame -- then ONE of the persons the frames hold,
    packed by prefix sum into a buffer sized by the largest per-rank total of the piece.  The padded gather moves
    Pout_max slots per frame whatever they hold (BASELINE configs[4] at Pout_max 32 for 8 persons: 851 MB per rank for
    213 MB of persons).

    compute_block(lo, hi, views) as in gather_track_chunked, views = xyzs [n, Pout_max, kn, 4] float32, pscore [n, Pout_max]
    float32, count [n] int32, flags [n] int32 (the kernels' padded outputs of the piece; they never leave the rank).
    The host reads the piece's per-rank totals (one small copy, after the piece's kernels): piece i + 1 is therefore queued
    BEFORE the gather of piece i, so the device computes it under that exchange.

    Returns persons [N, kn, 4] float32 and pscore [N] float32 (rows in gathered order: piece-major, then rank, then frame,
    then slot), offsets [F_total] int64 (row of a frame's first person), stored [F_total] int32 = min(count, Pout_max),
    count [F_total] int32 (the frames' true person counts), flags [F_total] int32, rank_status [world] int32, and
    gather_bytes (int: what this rank received).  compact_to_padded() rebuilds the padded track."""
    import contextlib
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (F_total + world - 1) // world
    dev = torch.device(device) if device is not None else torch.device("cpu")
    P, kn = int(pout_max), int(kn)
    i32, f32 = torch.int32, torch.float32
    if F_total <= 0:
        return dict(persons=torch.empty((0, kn, 4), dtype=f32, device=dev), pscore=torch.empty(0, dtype=f32, device=dev),
                    offsets=torch.empty(0, dtype=torch.int64, device=dev), stored=torch.empty(0, dtype=i32, device=dev),
                    count=torch.empty(0, dtype=i32, device=dev), flags=torch.empty(0, dtype=i32, device=dev),
                    rank_status=torch.zeros(world, dtype=i32, device=dev), gather_bytes=0)
    bad_local = not 0 <= n_local <= per
    n_claimed = n_local
    if bad_local:
        n_local = 0
    chunks = max(1, min(int(chunks), max(1, per)))
    cs = (per + chunks - 1) // chunks
    npieces = (per + cs - 1) // cs
    on_gpu = dev.type == "cuda"
    slots = [dict(xyzs=torch.zeros((cs, P, kn, 4), dtype=f32, device=dev), pscore=torch.zeros((cs, P), dtype=f32, device=dev),
                  count=torch.zeros(cs, dtype=i32, device=dev), flags=torch.zeros(cs, dtype=i32, device=dev)) for _ in range(2)]
    count_full = torch.zeros(world * per, dtype=i32, device=dev)
    flags_full = torch.zeros(world * per, dtype=i32, device=dev)
    stored_full = torch.zeros(world * per, dtype=i32, device=dev)
    offsets_full = torch.zeros(world * per, dtype=torch.int64, device=dev)
    status = torch.zeros(world, dtype=i32, device=dev)
    pieces_x, pieces_s = [], []
    rows_so_far = 0
    gather_bytes = 0
    if on_gpu:
        main = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(main)
        ready, packed = [None, None], [None, None]
    arangeP = torch.arange(P, device=dev, dtype=i32)

    def compute(i):
        slot = i & 1
        lo, hi = min(n_local, i * cs), min(n_local, (i + 1) * cs)
        if on_gpu and packed[slot] is not None:
            main.wait_event(packed[slot])                 # the pack that last read this slot is done
        if hi - lo < cs:
            for t in slots[slot].values():
                t.zero_()                                 # short or empty piece: counts of 0 behind the block
        if hi > lo:
            compute_block(lo, hi, {k: v[: hi - lo] for k, v in slots[slot].items()})
        if on_gpu:
            ready[slot] = torch.cuda.Event()
            ready[slot].record(main)

    compute(0)
    for i in range(npieces):
        slot = i & 1
        if i + 1 < npieces:
            compute(i + 1)                                # queued before the host waits for piece i's totals
        width = min(cs, per - i * cs)
        if on_gpu:
            side.wait_event(ready[slot])
        with (torch.cuda.stream(side) if on_gpu else contextlib.nullcontext()):
            S = slots[slot]
            meta = torch.empty(2 * cs + 4, dtype=i32, device=dev)
            meta[:cs] = S["count"]
            meta[cs:2 * cs] = S["flags"]
            meta[2 * cs:] = 1 if bad_local else 0
            meta_all = torch.empty(world * (2 * cs + 4), dtype=i32, device=dev)
            all_gather_flat(meta_all, meta, group=group)
            meta_all = meta_all.view(world, 2 * cs + 4)
            cnt_all = meta_all[:, :cs]
            sto_all = cnt_all.clamp(0, P)
            torch.maximum(status, meta_all[:, 2 * cs], out=status)
            totals = sto_all.sum(dim=1).cpu()             # the one host read of the piece (synchronises the side stream)
            maxtot = int(totals.max())
            gather_bytes += world * (2 * cs + 4) * 4
            # pack this rank's persons: row = exclusive prefix sum of the stored counts + slot
            mask = arangeP[None, :] < sto_all[rank][:, None]            # [cs, P]
            send = torch.zeros((maxtot, kn * 4 + 1), dtype=f32, device=dev)
            tot_r = int(totals[rank])
            if tot_r:
                send[:tot_r, : kn * 4] = S["xyzs"][mask].reshape(tot_r, kn * 4)
                send[:tot_r, kn * 4] = S["pscore"][mask]
            if on_gpu:
                packed[slot] = torch.cuda.Event()
                packed[slot].record(side)
            if maxtot:
                recv = torch.empty((world, maxtot, kn * 4 + 1), dtype=f32, device=dev)
                all_gather_flat(recv, send, group=group)
                gather_bytes += world * maxtot * (kn * 4 + 1) * 4
                rowmask = torch.arange(maxtot, device=dev)[None, :] < totals.to(dev)[:, None]   # [world, maxtot]
                rows = recv[rowmask]                                     # rank-major, then frame, then slot
                pieces_x.append(rows[:, : kn * 4].reshape(-1, kn, 4))
                pieces_s.append(rows[:, kn * 4].clone())
            # per-frame bookkeeping of the piece: frame (r, j) of the piece is frame r * per + i * cs + j of the track
            flat_sto = sto_all[:, :width].reshape(-1).to(torch.int64)    # rank-major like the rows ...
            full_sto = sto_all.reshape(-1).to(torch.int64)               # ... but the rows also count the slots behind `width`
            excl = torch.cumsum(full_sto, 0) - full_sto                  # (those hold no persons: count 0)
            idx = (torch.arange(world, device=dev)[:, None] * per + i * cs + torch.arange(width, device=dev)[None, :]).reshape(-1)
            offsets_full[idx] = rows_so_far + excl.view(world, cs)[:, :width].reshape(-1)
            stored_full[idx] = flat_sto.to(i32)
            count_full[idx] = cnt_all[:, :width].reshape(-1)
            flags_full[idx] = meta_all[:, cs:cs + width].reshape(-1)
            rows_so_far += int(totals.sum())
    if on_gpu:
        main.wait_stream(side)
        # the pieces were ALLOCATED under the side stream and are read on the main one from here on (the cat below, the
        # caller): tell the allocator, or their blocks go back to the side stream's pool when they are freed and a later
        # side-stream allocation may overwrite them under a main-stream reader (round-5 advice)
        for t_ in pieces_x + pieces_s:
            t_.record_stream(main)
    if bad_local:
        raise ValueError(f"gather_track_compact: a rank holds {n_claimed} frames but contiguous blocks of "
                         f"ceil({F_total} / {world}) = {per} frames are what is gathered (use shard_bounds)")
    if not on_gpu and bool(status.any()):
        bad = [q for q in range(world) if int(status[q])]
        raise ValueError(f"gather_track_compact: rank(s) {bad} hold more frames than their contiguous block of "
                         f"ceil({F_total} / {world}) = {per} (use shard_bounds); their blocks were gathered as empty")
    persons = torch.cat(pieces_x) if pieces_x else torch.empty((0, kn, 4), dtype=f32, device=dev)
    pscore = torch.cat(pieces_s) if pieces_s else torch.empty(0, dtype=f32, device=dev)
    return dict(persons=persons, pscore=pscore, offsets=offsets_full[:F_total], stored=stored_full[:F_total],
                count=count_full[:F_total], flags=flags_full[:F_total], rank_status=status, gather_bytes=gather_bytes)


def compact_to_padded(out, pout_max):
    """The padded track {xyzs [F, Pout_max, kn, 4], pscore [F, Pout_max]} of a gather_track_compact result (zeros in the unused
    slots, as the kernels write them)."""
    import torch
    persons, F = out["persons"], int(out["offsets"].shape[0])
    kn, dev = int(persons.shape[1]), persons.device
    xyzs = torch.zeros((F, pout_max, kn, 4), dtype=persons.dtype, device=dev)
    ps = torch.zeros((F, pout_max), dtype=persons.dtype, device=dev)
    slot = torch.arange(pout_max, device=dev)[None, :]
    mask = slot < out["stored"][:, None]
    rows = (out["offsets"][:, None] + slot)[mask]
    xyzs[mask] = persons[rows]
    ps[mask] = out["pscore"][rows]
    return dict(xyzs=xyzs, pscore=ps)


class ShardedTriangulator:
    """One instance per rank (one process per GPU).  `run(kpts_local, F_total)` triangulates this rank's frame block
    on its GPU, piece by piece, and returns the gathered track: the all-gather of piece i (joints, person scores,
    counts and flags in one buffer, one collective) overlaps the kernel of piece i + 1."""

    def __init__(self, K, R, t, params, pout_max=1, device=0, group=None, chunks="auto", reuse_buffers=False, compact=False,
                 zero_fill=None):
        """chunks: pieces per shard, or "auto" (auto_chunks: pieces of >= 32 768 frames, at most 8).
        reuse_buffers: keep the gathered tensors, the send / receive slots and the side stream between calls (no
        allocation in steady state); the track returned by one run() is then overwritten by the next.
        compact: gather the PERSONS, not the padded slots (gather_track_compact): per piece one all-gather of the counts
        (4 B per frame), then one of the persons packed by prefix sum -- the padded [Pout_max, kn, 4] block of a frame with
        8 persons in 32 slots is four times what its persons take ("or gather compacted persons + counts").
        zero_fill: False = the kernels leave the slots behind count[f] unwritten (SNOWTRI_CALL_NO_ZERO_FILL).  Default: False for
        the compact gather (it packs by the counts and never reads those slots), True for the padded one (the padding IS what
        every rank receives: it must be the documented zeros)."""
        import numpy as np
        from .batch import BatchTriangulator
        if zero_fill is None:
            zero_fill = not compact
        if not zero_fill and not compact:
            raise ValueError("ShardedTriangulator: the padded gather hands every rank the unused slots: they must be zero-filled")
        self.bt = BatchTriangulator(K, R, t, params, pout_max=pout_max, out_dtype=np.float32, device=device, zero_fill=zero_fill)
        self.group = group
        self.chunks = chunks
        self.device = device
        self.compact = bool(compact)
        self._ws = {} if reuse_buffers else None

    def regions(self):
        import torch
        kn, P = self.bt.params.keypoint_num, self.bt.pout_max
        return {"xyzs": ((P, kn, 4), torch.float32), "pscore": ((P,), torch.float32),
                "count": ((), torch.int32), "flags": ((), torch.int32)}

    def verify(self):
        """Reads the status words of the last gathered run (one small device-to-host copy: synchronises) and raises if a
        rank's block was refused -- its frames were gathered as zeros (gather_track_chunked).  run(strict=True) calls it."""
        st = getattr(self, "last_status", None)
        if st is not None and bool(st.any().item()):
            bad = [q for q in range(int(st.numel())) if int(st[q])]
            raise ValueError(f"ShardedTriangulator: rank(s) {bad} held more frames than their contiguous block (use shard_bounds); "
                             f"their frames were gathered as zeros")

    def run(self, kpts_local, F_total, n_persons_local=None, gather=True, chunks=None, strict=False):
        """strict: check the ranks' status words before returning (a host synchronisation; the default leaves them in
        out["rank_status"] / self.last_status for verify() at the caller's next synchronisation point)."""
        if not gather:
            return self.bt.run_torch(kpts_local, n_persons_local)
        if self.compact:
            return self._run_compact(kpts_local, F_total, n_persons_local, chunks, strict)

        def compute_block(lo, hi, views):
            out = {k: v[: hi - lo] for k, v in views.items()}
            self.bt.run_torch(kpts_local[lo:hi], None if n_persons_local is None else n_persons_local[lo:hi], out=out)

        import torch.distributed as dist
        chunks = self.chunks if chunks is None else chunks
        if chunks == "auto":
            world = dist.get_world_size(self.group)
            chunks = auto_chunks((F_total + world - 1) // world)
        self.last_chunks = int(chunks)
        out = gather_track_chunked(compute_block, int(kpts_local.shape[0]), F_total, self.regions(),
                                   chunks=int(chunks), group=self.group,
                                   device=kpts_local.device, workspace=self._ws)
        self.last_status = out["rank_status"]
        self.last_gather_bytes = self._gather_bytes_padded(F_total)
        if strict:
            self.verify()
        return out

    def _run_compact(self, kpts_local, F_total, n_persons_local, chunks, strict):
        def compute_block(lo, hi, views):
            self.bt.run_torch(kpts_local[lo:hi], None if n_persons_local is None else n_persons_local[lo:hi], out=views)

        import torch.distributed as dist
        chunks = self.chunks if chunks is None else chunks
        if chunks == "auto":
            world = dist.get_world_size(self.group)
            chunks = auto_chunks((F_total + world - 1) // world)
        self.last_chunks = int(chunks)
        out = gather_track_compact(compute_block, int(kpts_local.shape[0]), F_total, self.bt.params.keypoint_num, self.bt.pout_max,
                                   chunks=int(chunks), group=self.group, device=kpts_local.device)
        self.last_status = out["rank_status"]
        self.last_gather_bytes = out["gather_bytes"]
        if strict:
            self.verify()
        return out

    def _gather_bytes_padded(self, F_total):
        """bytes ONE rank receives in the padded gather of a whole track (what xGMI moves per rank)"""
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        per = (F_total + world - 1) // world
        kn, P = self.bt.params.keypoint_num, self.bt.pout_max
        return world * per * (P * kn * 16 + P * 4 + 8)


# ---------------------------------------------------------------------------------------------------
# Person counts that vary from frame to frame, on a frame-sharded track.  The reference identifies persons by LIST INDEX and its
# filter banks are those of frame 0: `zip` truncates every later frame against them (triangulation.py:169-171, blender.py:152-166),
# so frame f carries tracked[f] = min(count[f], count[0]) persons and slot i is filtered over the frames with tracked > i -- in frame
# order, frame 0 among them (it seeds the slot).  Nothing of that depends on EARLIER frames except through count[0]: one small
# exchange hands every rank count[0], and per slot the frames a rank holds are a contiguous-in-order piece of the slot's sequence,
# i.e. exactly a "block" of the carry exchanges below (blocks may be empty; the block that holds frame 0 starts the track).

def tracked_counts(count_local, F_total, n_slots, group=None, flag_bits=0):
    """count_local [T_r] int32: the person counts of this rank's shard_bounds block.  ONE all-gather of 3 int32 per rank ->
    (tracked_local [T_r] int32 = min(count, count[0]), count0, bits): count0 = the count of frame 0 of the TRACK (from the
    rank that holds it), bits = OR over the ranks of `flag_bits` (an int or an int32 scalar tensor of this rank's condition bits,
    e.g. 1 = "a singular frame"): every rank learns them before the exchanges that follow, so all raise together and nobody is
    left waiting in a collective.  Raises ValueError on every rank if frame 0 holds more persons than the track has slots."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi, _ = shard_bounds(int(F_total), world, rank)
    if hi - lo != int(count_local.shape[0]):
        raise ValueError(f"tracked_counts: rank {rank} holds {int(count_local.shape[0])} frames, shard_bounds gives it [{lo}, {hi})")
    dev = count_local.device
    mine = torch.zeros(3, dtype=torch.int32, device=dev)
    if lo == 0 and hi > 0:
        mine[0] = 1
        mine[1] = count_local[0]
    mine[2] = flag_bits.to(device=dev, dtype=torch.int32) if torch.is_tensor(flag_bits) else int(flag_bits)
    allr = torch.empty(3 * world, dtype=torch.int32, device=dev)
    all_gather_flat(allr, mine, group=group)
    allr = allr.view(world, 3).cpu()
    holders = [q for q in range(world) if int(allr[q, 0])]
    count0 = int(allr[holders[0], 1]) if holders else 0
    if count0 > int(n_slots):
        raise ValueError(f"frame 0 resolved to {count0} persons, the track has {int(n_slots)} slots (n_persons_out)")
    tracked = torch.clamp(count_local, max=count0).to(torch.int32)
    bits = 0
    for q in range(world):
        bits |= int(allr[q, 2])
    return tracked, count0, bits


# ---------------------------------------------------------------------------------------------------
# Row N1 on the sharded track: temporal smoothing WITHOUT reassembling the track.
# The filter state obeys s_t = A s_{t-1} + b_t (snowtri_smooth.hpp), so each rank filters its own frame
# block from a zero entering state, the ranks exchange ONE small all-gather (4n+1 doubles each: end state,
# first/last input row, block length) and each adds the response of its true entering state.

def smooth_coeffs(f, z, r, dt):
    """{A (2x2), cx, cxd} of the state update, from the library (same arithmetic as the kernels)."""
    import ctypes as ct
    import numpy as np
    from . import _lib
    out = (ct.c_double * 6)()
    _lib.check(_lib.lib().snowtri_smooth_coeffs(float(f), float(z), float(r), float(dt), out), "snowtri_smooth_coeffs")
    A = np.array([[out[0], out[1]], [out[2], out[3]]])
    return A, float(out[4]), float(out[5])


def combine_carries(payloads, rank, A, cxd):
    """payloads[q] = (E_q[n,2], x_first_q[n], x_last_q[n], T_q) for every rank q in frame order.
    Returns start_state[n,2] for `rank`: the filter state entering its first filtered frame, already
    corrected for the input derivative across the shard boundary."""
    import numpy as np
    Ainv = np.linalg.inv(A)
    n = payloads[0][1].shape[0]
    S = None                    # true state entering the current shard
    x_last_prev = None
    for q in range(rank + 1):
        E, x_first, x_last, T = payloads[q]
        T = int(T)
        if T == 0:
            continue
        if S is None:           # the shard that starts the track: seed (x0, 0), frame 0 passes through
            S = np.stack([np.asarray(x_first, dtype=np.float64), np.zeros(n)], axis=1)
            start, m = S, T - 1
        else:
            delta = cxd * (np.asarray(x_first) - x_last_prev)           # missing (x_t - x_{t-1}) term of its first frame
            start = S + np.outer(delta, Ainv[:, 1])                     # S + A^-1 (0, delta)
            m = T
        if q == rank:
            return np.ascontiguousarray(start)
        if m > 0:
            S = start @ np.linalg.matrix_power(A, m).T + np.asarray(E).reshape(n, 2)
        x_last_prev = np.asarray(x_last, dtype=np.float64)
    return np.zeros((n, 2)) if S is None else np.ascontiguousarray(S)


def smooth_exchange(x_local, local_fn, combine_fn, fix_fn, group=None, first=None):
    """The sharded smoothing protocol around three callables (the product binds them to the C ABI on the shard's GPU,
    `smooth_track_sharded`; the gloo tests bind NumPy stand-ins to drive exactly this exchange over 8 CPU ranks):

        local_fn(x_local, first, y, payload)  zero-state response of the block into y, its end state into payload[:2n]
        combine_fn(allp, rank, start)         entering state of `rank` from the gathered payloads [world, 4n + 1]
        fix_fn(y, first, start)               y += response of the entering state

    payload of a rank = (end state [n, 2], first input row [n], last input row [n], block length).  ONE all-gather.
    `first`: does this block start the track (frame 0 passes through and seeds the filters)?  Default: rank 0 -- the
    layout of shard_bounds, where only TRAILING blocks can be empty.  snowtri_smooth_shard_combine treats the first
    NON-EMPTY payload as the start of the track, so a caller with an empty leading block must pass first=True on the
    rank that holds the first frames (include/snowtri.h states the same contract)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    T = int(x_local.shape[0])
    n = 1
    for d in x_local.shape[1:]:
        n *= int(d)
    if first is None:
        first = rank == 0
    y = torch.empty_like(x_local)
    payload = torch.zeros(4 * n + 1, dtype=torch.float64, device=x_local.device)
    payload[4 * n] = T
    if T > 0:
        local_fn(x_local, bool(first), y, payload)
        payload[2 * n:3 * n] = x_local[0].reshape(-1)
        payload[3 * n:4 * n] = x_local[-1].reshape(-1)
    flat = torch.empty(world * (4 * n + 1), dtype=torch.float64, device=x_local.device)
    all_gather_flat(flat, payload, group=group)                         # the one exchange: 4n+1 doubles per rank
    allp = flat.view(world, 4 * n + 1)
    if T == 0:
        return y
    start = torch.empty((n, 2), dtype=torch.float64, device=x_local.device)
    combine_fn(allp, rank, start)
    fix_fn(y, bool(first), start)
    return y


def smooth_exchange2(x_local, reduce_fn, combine_fn, scan_fn, group=None, first=None):
    """The sharded smoothing protocol in TWO passes over the block (round 6; smooth_exchange above is the five-pass form the
    first rounds shipped, kept for its callers and its gloo tests):

        reduce_fn(x_local, first, payload)   payload[:2n] = end state of the block's ZERO-STATE response (x read once, no track written)
        combine_fn(allp, rank, start)        entering state of `rank` from the gathered payloads [world, 4n + 1] (unchanged)
        scan_fn(x_local, first, start, y)    y = the block filtered from its true entering state (x read once, y written once)

    Same payload, same ONE all-gather, same `first` contract as smooth_exchange."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    T = int(x_local.shape[0])
    n = 1
    for d in x_local.shape[1:]:
        n *= int(d)
    if first is None:
        first = rank == 0
    y = torch.empty_like(x_local)
    payload = torch.zeros(4 * n + 1, dtype=torch.float64, device=x_local.device)
    payload[4 * n] = T
    if T > 0:
        reduce_fn(x_local, bool(first), payload)
        payload[2 * n:3 * n] = x_local[0].reshape(-1)
        payload[3 * n:4 * n] = x_local[-1].reshape(-1)
    flat = torch.empty(world * (4 * n + 1), dtype=torch.float64, device=x_local.device)
    all_gather_flat(flat, payload, group=group)
    if T == 0:
        return y
    start = torch.empty((n, 2), dtype=torch.float64, device=x_local.device)
    combine_fn(flat.view(world, 4 * n + 1), rank, start)
    scan_fn(x_local, bool(first), start, y)
    return y


def smooth_track_sharded(x_local, f=2, z=0.75, r=0, delta_time=1 / 30, group=None, ctx=None, F_total=None, first=None):
    """x_local: this rank's frame block [T_r, ...] (CUDA float64 tensor, frame-major) of a track sharded in
    frame order over the ranks of `group`.  Returns the filtered block; the full track is never gathered.
    F_total (optional): the length of the whole track -- the block is then checked against shard_bounds and "this block
    starts the track" is derived from it instead of assumed for rank 0.  first (optional, without F_total): says it outright --
    for blocks that are NOT shard_bounds blocks, e.g. the frames of a block that carry one person slot (ShardedTrackPipeline)."""
    import ctypes as ct
    import torch
    import torch.distributed as dist
    from . import _lib
    assert x_local.is_cuda and x_local.dtype == torch.float64 and x_local.is_contiguous()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if F_total is not None:
        lo, hi, _ = shard_bounds(int(F_total), world, rank)
        if hi - lo != int(x_local.shape[0]):
            raise ValueError(f"smooth_track_sharded: rank {rank} holds {int(x_local.shape[0])} frames, shard_bounds gives it [{lo}, {hi})")
        first = lo == 0 and hi > lo
    n = 1
    for d in x_local.shape[1:]:
        n *= int(d)
    ctx = ctx or _lib.scratch_context(x_local.device.index)     # scratch and kernels on the GPU that holds the shard
    stream = ct.c_void_p(torch.cuda.current_stream(x_local.device).cuda_stream)
    L = ctx.L
    fzrd = (float(f), float(z), float(r), float(delta_time))

    def reduce_fn(x, is_first, payload):
        _lib.check(L.snowtri_smooth_shard_reduce(ctx.handle, int(x.shape[0]), n, ct.c_void_p(x.data_ptr()), 1 if is_first else 0, *fzrd,
                                                 ct.c_void_p(payload.data_ptr()), stream), "snowtri_smooth_shard_reduce")

    def combine_fn(allp, rk, start):
        # the entering state of this shard from the gathered carries, ON the device and the stream (k_smooth_combine: no host
        # round trip between the all-gather and the fix; combine_carries above is its host twin)
        _lib.check(L.snowtri_smooth_shard_combine(ctx.handle, world, rk, n, ct.c_void_p(allp.data_ptr()), *fzrd,
                                                  ct.c_void_p(start.data_ptr()), _lib.DEVICE, stream), "snowtri_smooth_shard_combine")

    def scan_fn(x, is_first, start, y):
        _lib.check(L.snowtri_smooth_shard_scan(ctx.handle, int(x.shape[0]), n, ct.c_void_p(x.data_ptr()), 1 if is_first else 0,
                                               ct.c_void_p(start.data_ptr()), *fzrd, ct.c_void_p(y.data_ptr()), stream), "snowtri_smooth_shard_scan")

    return smooth_exchange2(x_local, reduce_fn, combine_fn, scan_fn, group=group, first=first)


# ---------------------------------------------------------------------------------------------------
# Row N2 on the sharded track: the per-bone filters of Human_Triangulation_Blender_Smooth (blender.py:145-178) WITHOUT
# reassembling the control-point track.  An invalid point feeds its filter the previous input, i.e. the filters see the held
# sequence x_eff[t] = valid[t] ? x[t] : x_eff[t-1]; on x_eff they are the plain linear filters of row N1.  So: one small
# exchange for the hold (last valid input of every block), then the carry exchange of smooth_exchange on x_eff.

def hold_exchange(pts_local, val_local, last_fn, apply_fn, group=None):
    """x_eff of this rank's block.  pts_local [T, P, 24, 4] float64, val_local [T, P, 24] uint8.
        last_fn(pts, val, payload)               payload[2n] = last valid input per lane | found per lane (n = P * 96)
        apply_fn(allp, rank, pts, val, held)     held = x_eff of the block given the gathered payloads [world, 2n]
    (the product binds them to snowtri_blender_hold_shard_last / _apply; the gloo tests bind NumPy stand-ins).  ONE all-gather."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = int(pts_local.shape[1]) * 24 * 4
    payload = torch.zeros(2 * n, dtype=torch.float64, device=pts_local.device)
    if int(pts_local.shape[0]) > 0:
        last_fn(pts_local, val_local, payload)
    flat = torch.empty(world * 2 * n, dtype=torch.float64, device=pts_local.device)
    all_gather_flat(flat, payload, group=group)
    held = torch.empty_like(pts_local)
    if int(pts_local.shape[0]) > 0:
        apply_fn(flat.view(world, 2 * n), rank, pts_local, val_local, held)
    return held


def blender_smooth_sharded(pts_local, val_local, fzr, delta_time=1 / 30, group=None, ctx=None, F_total=None, first=None):
    """pts_local [T_r, P, 24, 4] (CUDA float64), val_local [T_r, P, 24] (CUDA uint8): this rank's frame block of a control-point
    track sharded in frame order.  Returns the smoothed block [T_r, P, 24, 4] = the rows snowtri_blender_smooth returns for
    these frames on the whole track (to rounding: the carries are propagated in closed form).  Two small all-gathers
    (2n and 4n + 1 doubles per rank, n = P * 96); the track itself never moves."""
    import ctypes as ct
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import _lib
    assert pts_local.is_cuda and pts_local.dtype == torch.float64 and pts_local.is_contiguous()
    assert val_local.is_cuda and val_local.dtype == torch.uint8 and val_local.is_contiguous()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    T, P = int(pts_local.shape[0]), int(pts_local.shape[1])
    if F_total is not None:
        lo, hi, _ = shard_bounds(int(F_total), world, rank)
        if hi - lo != T:
            raise ValueError(f"blender_smooth_sharded: rank {rank} holds {T} frames, shard_bounds gives it [{lo}, {hi})")
        first = lo == 0 and hi > lo
    ctx = ctx or _lib.scratch_context(pts_local.device.index)
    stream = ct.c_void_p(torch.cuda.current_stream(pts_local.device).cuda_stream)
    L = ctx.L
    fzr = np.ascontiguousarray(fzr, dtype=np.float64).reshape(24, 3)
    dt = float(delta_time)

    def last_fn(p, v, payload):
        _lib.check(L.snowtri_blender_hold_shard_last(ctx.handle, int(p.shape[0]), P, ct.c_void_p(p.data_ptr()), ct.c_void_p(v.data_ptr()),
                                                     ct.c_void_p(payload.data_ptr()), stream), "snowtri_blender_hold_shard_last")

    def apply_fn(allp, rk, p, v, held):
        _lib.check(L.snowtri_blender_hold_shard_apply(ctx.handle, world, rk, int(p.shape[0]), P, ct.c_void_p(p.data_ptr()),
                                                      ct.c_void_p(v.data_ptr()), ct.c_void_p(allp.data_ptr()), ct.c_void_p(held.data_ptr()),
                                                      stream), "snowtri_blender_hold_shard_apply")

    held = hold_exchange(pts_local, val_local, last_fn, apply_fn, group=group)

    def reduce_fn(x, is_first, payload):
        _lib.check(L.snowtri_blender_smooth_shard_reduce(ctx.handle, int(x.shape[0]), P, ct.c_void_p(x.data_ptr()), 1 if is_first else 0,
                                                         _lib.ptr(fzr), dt, ct.c_void_p(payload.data_ptr()), stream), "snowtri_blender_smooth_shard_reduce")

    def combine_fn(allp, rk, start):
        _lib.check(L.snowtri_blender_smooth_shard_combine(ctx.handle, world, rk, P, ct.c_void_p(allp.data_ptr()), _lib.ptr(fzr), dt,
                                                          ct.c_void_p(start.data_ptr()), stream), "snowtri_blender_smooth_shard_combine")

    def scan_fn(x, is_first, start, y):
        _lib.check(L.snowtri_blender_smooth_shard_scan(ctx.handle, int(x.shape[0]), P, ct.c_void_p(x.data_ptr()), 1 if is_first else 0,
                                                       ct.c_void_p(start.data_ptr()), _lib.ptr(fzr), dt, ct.c_void_p(y.data_ptr()), stream),
                   "snowtri_blender_smooth_shard_scan")

    y = smooth_exchange2(held, reduce_fn, combine_fn, scan_fn, group=group, first=first)
    is_first = (rank == 0) if first is None else first
    if is_first and T > 0:
        y[0].copy_(pts_local[0])       # frame 0 of the track is returned as given, NaNs included (blender.py:176)
    return y

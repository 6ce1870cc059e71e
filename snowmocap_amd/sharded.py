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
        return {name: torch.empty((0,) + tuple(tail), dtype=dt, device=dev) for name, (tail, dt) in regions.items()}
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
            dist.all_gather_into_tensor(recv[slot], send[slot], group=group)
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


class ShardedTriangulator:
    """One instance per rank (one process per GPU).  `run(kpts_local, F_total)` triangulates this rank's frame block
    on its GPU, piece by piece, and returns the gathered track: the all-gather of piece i (joints, person scores,
    counts and flags in one buffer, one collective) overlaps the kernel of piece i + 1."""

    def __init__(self, K, R, t, params, pout_max=1, device=0, group=None, chunks="auto", reuse_buffers=False):
        """chunks: pieces per shard, or "auto" (auto_chunks: pieces of >= 32 768 frames, at most 8).
        reuse_buffers: keep the gathered tensors, the send / receive slots and the side stream between calls (no
        allocation in steady state); the track returned by one run() is then overwritten by the next."""
        import numpy as np
        from .batch import BatchTriangulator
        self.bt = BatchTriangulator(K, R, t, params, pout_max=pout_max, out_dtype=np.float32, device=device)
        self.group = group
        self.chunks = chunks
        self.device = device
        self._ws = {} if reuse_buffers else None

    def regions(self):
        import torch
        kn, P = self.bt.params.keypoint_num, self.bt.pout_max
        return {"xyzs": ((P, kn, 4), torch.float32), "pscore": ((P,), torch.float32),
                "count": ((), torch.int32), "flags": ((), torch.int32)}

    def run(self, kpts_local, F_total, n_persons_local=None, gather=True, chunks=None):
        if not gather:
            return self.bt.run_torch(kpts_local, n_persons_local)

        def compute_block(lo, hi, views):
            out = {k: v[: hi - lo] for k, v in views.items()}
            self.bt.run_torch(kpts_local[lo:hi], None if n_persons_local is None else n_persons_local[lo:hi], out=out)

        import torch.distributed as dist
        chunks = self.chunks if chunks is None else chunks
        if chunks == "auto":
            world = dist.get_world_size(self.group)
            chunks = auto_chunks((F_total + world - 1) // world)
        self.last_chunks = int(chunks)
        return gather_track_chunked(compute_block, int(kpts_local.shape[0]), F_total, self.regions(),
                                    chunks=int(chunks), group=self.group,
                                    device=kpts_local.device, workspace=self._ws)


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
    dist.all_gather_into_tensor(flat, payload, group=group)             # the one exchange: 4n+1 doubles per rank
    allp = flat.view(world, 4 * n + 1)
    if T == 0:
        return y
    start = torch.empty((n, 2), dtype=torch.float64, device=x_local.device)
    combine_fn(allp, rank, start)
    fix_fn(y, bool(first), start)
    return y


def smooth_track_sharded(x_local, f=2, z=0.75, r=0, delta_time=1 / 30, group=None, ctx=None, F_total=None):
    """x_local: this rank's frame block [T_r, ...] (CUDA float64 tensor, frame-major) of a track sharded in
    frame order over the ranks of `group`.  Returns the filtered block; the full track is never gathered.
    F_total (optional): the length of the whole track -- the block is then checked against shard_bounds and "this block
    starts the track" is derived from it instead of assumed for rank 0."""
    import ctypes as ct
    import torch
    import torch.distributed as dist
    from . import _lib
    assert x_local.is_cuda and x_local.dtype == torch.float64 and x_local.is_contiguous()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    first = None
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
    L = _lib.lib()
    fzrd = (float(f), float(z), float(r), float(delta_time))

    def local_fn(x, is_first, y, payload):
        _lib.check(L.snowtri_smooth_shard_local(ctx.handle, int(x.shape[0]), n, ct.c_void_p(x.data_ptr()), 1 if is_first else 0, *fzrd,
                                                ct.c_void_p(y.data_ptr()), ct.c_void_p(payload.data_ptr()), _lib.DEVICE, stream),
                   "snowtri_smooth_shard_local")

    def combine_fn(allp, rk, start):
        # the entering state of this shard from the gathered carries, ON the device and the stream (k_smooth_combine: no host
        # round trip between the all-gather and the fix; combine_carries above is its host twin)
        _lib.check(L.snowtri_smooth_shard_combine(ctx.handle, world, rk, n, ct.c_void_p(allp.data_ptr()), *fzrd,
                                                  ct.c_void_p(start.data_ptr()), _lib.DEVICE, stream), "snowtri_smooth_shard_combine")

    def fix_fn(y, is_first, start):
        _lib.check(L.snowtri_smooth_shard_fix(ctx.handle, int(y.shape[0]), n, 1 if is_first else 0, ct.c_void_p(start.data_ptr()), *fzrd,
                                              ct.c_void_p(y.data_ptr()), _lib.DEVICE, stream), "snowtri_smooth_shard_fix")

    return smooth_exchange(x_local, local_fn, combine_fn, fix_fn, group=group, first=first)

"""Frame sharding across the GPUs of one node + the single all-gather that reassembles the track.

Frames are independent through triangulate + condense (the reference reads only the current
frame's detections: triangulation.py:50-162; state is cleared per frame, main.py:106), so the batch
is split into contiguous frame blocks, one per rank, with NO data-path collective inside the
kernels; one all-gather (RCCL over xGMI when the backend is "nccl") hands every rank the whole
3D track, which is what the next stage -- temporal smoothing, a recurrence over frames
(triangulation.py:164-186) -- needs.  Results are bit-identical to the single-GPU run.

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


class ShardedTriangulator:
    """One instance per rank (one process per GPU).  `run(kpts_local)` triangulates this rank's frame
    block on its GPU and returns the gathered track."""

    def __init__(self, K, R, t, params, pout_max=1, device=0, group=None):
        import numpy as np
        from .batch import BatchTriangulator
        self.bt = BatchTriangulator(K, R, t, params, pout_max=pout_max, out_dtype=np.float32, device=device)
        self.group = group

    def run(self, kpts_local, F_total, n_persons_local=None, gather=True):
        out = self.bt.run_torch(kpts_local, n_persons_local)
        if not gather:
            return out
        return dict(xyzs=gather_track(out["xyzs"], F_total, self.group),
                    count=gather_track(out["count"], F_total, self.group),
                    flags=gather_track(out["flags"], F_total, self.group))


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


def smooth_track_sharded(x_local, f=2, z=0.75, r=0, delta_time=1 / 30, group=None, ctx=None):
    """x_local: this rank's frame block [T_r, ...] (CUDA float64 tensor, frame-major) of a track sharded in
    frame order over the ranks of `group`.  Returns the filtered block; the full track is never gathered."""
    import ctypes as ct
    import numpy as np
    import torch
    import torch.distributed as dist
    from . import _lib
    assert x_local.is_cuda and x_local.dtype == torch.float64 and x_local.is_contiguous()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    T = int(x_local.shape[0])
    lanes = tuple(x_local.shape[1:])
    n = int(np.prod(lanes)) if lanes else 1
    ctx = ctx or _lib.scratch_context(x_local.device.index)     # scratch and kernels on the GPU that holds the shard
    stream = ct.c_void_p(torch.cuda.current_stream(x_local.device).cuda_stream)
    L = _lib.lib()
    y = torch.empty_like(x_local)
    payload = torch.zeros(4 * n + 1, dtype=torch.float64, device=x_local.device)
    payload[4 * n] = T
    if T > 0:
        first = 1 if rank == 0 else 0      # shard_bounds blocks: only trailing blocks can be empty
        _lib.check(L.snowtri_smooth_shard_local(ctx.handle, T, n, ct.c_void_p(x_local.data_ptr()), first, float(f),
                                                float(z), float(r), float(delta_time), ct.c_void_p(y.data_ptr()),
                                                ct.c_void_p(payload.data_ptr()), _lib.DEVICE, stream),
                   "snowtri_smooth_shard_local")
        payload[2 * n:3 * n] = x_local[0].reshape(-1)
        payload[3 * n:4 * n] = x_local[-1].reshape(-1)
    allp = torch.empty((world, 4 * n + 1), dtype=torch.float64, device=x_local.device)
    dist.all_gather_into_tensor(allp, payload, group=group)             # the one exchange: 4n+1 doubles per rank
    if T == 0:
        return y
    host = allp.cpu().numpy()
    payloads = [(host[q, :2 * n].reshape(n, 2), host[q, 2 * n:3 * n], host[q, 3 * n:4 * n], host[q, 4 * n]) for q in range(world)]
    A, _cx, cxd = smooth_coeffs(f, z, r, delta_time)
    start = torch.from_numpy(combine_carries(payloads, rank, A, cxd)).to(x_local.device)
    is_first = rank == 0
    _lib.check(L.snowtri_smooth_shard_fix(ctx.handle, T, n, 1 if is_first else 0, ct.c_void_p(start.data_ptr()), float(f),
                                          float(z), float(r), float(delta_time), ct.c_void_p(y.data_ptr()),
                                          _lib.DEVICE, stream), "snowtri_smooth_shard_fix")
    return y

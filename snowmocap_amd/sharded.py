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

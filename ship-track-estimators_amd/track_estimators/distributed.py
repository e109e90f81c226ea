"""
Track-sharded multi-GPU execution (one process per GPU, ``torch.distributed``; backend "nccl" is RCCL on ROCm).

Tracks are independent (no cross-track term in kalman_filter.py:61-117 or unscented.py:285-351), so the batch is cut
into contiguous blocks of tracks, one per rank, with no communication while filtering.  The only exchange is the final
all-gather of the smoothed longitude/latitude histories (BASELINE.json configs[2]); the full means/covariances are 20x
larger and stay on the rank that produced them.
"""
from __future__ import annotations

from typing import Tuple


def shard_bounds(ntracks: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank ``rank``; the first ``ntracks % world`` ranks get one extra track."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    q, r = divmod(int(ntracks), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_smoothed_positions(sm_mean, group=None, out=None, pad_to=None):
    """
    All-gather rows 0-1 (lon, lat) of a smoothed-mean tensor laid out [N+1][4][B_local] (include/ste.h).

    Every rank must pass the same N and B_local; with an uneven split pass ``pad_to`` = the largest shard and cut the
    result with ``assemble_tracks``.  Returns a tensor [world][N+1][2][B_local] on every rank; rank r's block is ``out[r]``.  One collective, issued on the current stream
    after the smoother kernel; with RCCL over xGMI each peer's shard travels on its own link.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    local = sm_mean[:, :2, :].contiguous()
    if pad_to is not None and pad_to != local.shape[2]:
        padded = torch.zeros(local.shape[:2] + (int(pad_to),), dtype=local.dtype, device=local.device)
        padded[:, :, : local.shape[2]] = local
        local = padded
    if out is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    # concatenation along dim 0 is the layout both RCCL and gloo accept for all_gather_into_tensor
    dist.all_gather_into_tensor(out.view((world * local.shape[0],) + tuple(local.shape[1:])), local, group=group)
    return out


def assemble_tracks(gathered, total: int):
    """[world][N+1][2][bmax] from an all-gather of (padded) shards -> [N+1][2][total] in global track order
    (``shard_bounds`` layout: the first ``total % world`` ranks hold one track more than the others)."""
    import torch

    world = gathered.shape[0]
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        parts.append(gathered[r][:, :, : hi - lo])
    return torch.cat(parts, dim=2)


class OverlappedGather:
    """
    Double-buffered, asynchronous form of ``gather_smoothed_positions`` for a stream of batches.

    ``launch(sm_mean)`` snapshots the lon/lat rows into a send buffer on the current stream and starts the all-gather
    with ``async_op=True``: RCCL runs it on its own stream behind the kernels already queued, while the caller goes on
    to queue the next batch's filter kernels.  Two slots alternate; a slot is waited for just before it is reused, and
    ``finish()`` drains what is still in flight.  ``result(i)`` is the gathered tensor of the i-th launch (valid after
    that launch was waited for).
    """

    def __init__(self, nrows: int, ntracks: int, device, dtype=None, group=None):
        import torch
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        dtype = dtype or torch.float64
        self.send = [torch.empty((nrows, 2, ntracks), dtype=dtype, device=device) for _ in range(2)]
        self.recv = [torch.empty((self.world, nrows, 2, ntracks), dtype=dtype, device=device) for _ in range(2)]
        self.work = [None, None]
        self.count = 0

    def launch(self, sm_mean):
        """``sm_mean``: [nrows][4][b] with b <= ntracks (a short last shard is zero-padded to the common width)."""
        slot = self.count & 1
        if self.work[slot] is not None:
            self.work[slot].wait()  # the current stream now waits for the collective that last used this slot
        b = sm_mean.shape[2]
        if b == self.send[slot].shape[2]:
            self.send[slot].copy_(sm_mean[:, :2, :])
        else:
            self.send[slot][:, :, :b].copy_(sm_mean[:, :2, :])
            self.send[slot][:, :, b:].zero_()
        out = self.recv[slot]
        self.work[slot] = self.dist.all_gather_into_tensor(
            out.view((self.world * out.shape[1],) + tuple(out.shape[2:])), self.send[slot], group=self.group,
            async_op=True)
        self.count += 1
        return slot

    def finish(self):
        for slot in (0, 1):
            if self.work[slot] is not None:
                self.work[slot].wait()
                self.work[slot] = None

    def result(self, slot: int):
        return self.recv[slot]

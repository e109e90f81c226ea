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

    ``launch(pos)`` starts the all-gather of one batch's smoothed lon / lat with ``async_op=True``: RCCL runs it on its own
    stream behind the kernels already queued on the current stream, while the caller goes on to queue the next batch's
    filter kernels.  ``pos`` is either the smoother's own ``sm_pos`` output ([nrows][2][ntracks], include/ste.h) -- then
    the collective sends that tensor as it is, no copy -- or a smoothed-mean tensor [nrows][4][b], whose lon / lat rows are
    first snapshotted into a send buffer (also the route for a short last shard, which is zero-padded to the common width).
    Two receive slots alternate; a slot is waited for just before it is reused, and ``finish()`` drains what is still in
    flight.  ``result(i)`` is the gathered tensor of slot i (valid after that launch was waited for).

    ``launch`` returns the receive slot.  After a launch that sent the caller's tensor itself, ``reader_done`` is an event
    that marks the END of that collective (recorded on a side stream, so the launching stream does not wait): whoever
    rewrites ``pos`` next -- the batch's next forward pass / smoother -- has to wait for it; ``SmootherPipeline.submit``
    does, through the value its ``after_smoother`` hook returns (``launch_for_pipeline``).
    """

    def __init__(self, nrows: int, ntracks: int, device, dtype=None, group=None):
        import torch
        import torch.distributed as dist

        self.torch = torch
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        dtype = dtype or torch.float64
        self.shape = (nrows, 2, ntracks)
        self.send = [None, None]  # snapshot buffers, allocated at first need
        self.recv = [torch.empty((self.world, nrows, 2, ntracks), dtype=dtype, device=device) for _ in range(2)]
        self.work = [None, None]
        self.count = 0
        self.direct = 0  # launches that sent the smoother's own output
        self.reader_done = None
        self._on_gpu = torch.device(device).type == "cuda"
        self._side = torch.cuda.Stream(device) if self._on_gpu else None
        self._device, self._dtype = device, dtype

    def _snapshot(self, slot, pos):
        torch = self.torch
        if self.send[slot] is None:
            self.send[slot] = torch.empty(self.shape, dtype=self._dtype, device=self._device)
        b = pos.shape[2]
        if b == self.shape[2]:
            self.send[slot].copy_(pos[:, :2, :])
        else:
            self.send[slot][:, :, :b].copy_(pos[:, :2, :])
            self.send[slot][:, :, b:].zero_()
        return self.send[slot]

    def launch(self, pos):
        torch = self.torch
        slot = self.count & 1
        if self.work[slot] is not None:
            self.work[slot].wait()  # the current stream now waits for the collective that last used this slot
        if tuple(pos.shape) == self.shape and pos.is_contiguous():
            src = pos  # sm_pos: the smoother's own output
            self.direct += 1
        else:
            src = self._snapshot(slot, pos)
        out = self.recv[slot]
        work = self.dist.all_gather_into_tensor(
            out.view((self.world * out.shape[1],) + tuple(out.shape[2:])), src, group=self.group, async_op=True)
        self.work[slot] = work
        self.count += 1
        self.reader_done = None
        if src is pos and self._on_gpu:
            # end of the collective, without making the launching stream wait for it: a side stream waits, an event marks it
            # (a snapshot is this object's own buffer: nothing of the caller's is still being read then)
            with torch.cuda.stream(self._side):
                work.wait()
                self.reader_done = torch.cuda.Event()
                self.reader_done.record(self._side)
        return slot

    def launch_for_pipeline(self, pos):
        """``after_smoother`` hook body for ``SmootherPipeline.submit``: start the gather, hand back the event the next
        use of ``pos``'s buffers must wait for (None when a snapshot was sent)."""
        self.launch(pos)
        return self.reader_done

    def finish(self):
        for slot in (0, 1):
            if self.work[slot] is not None:
                self.work[slot].wait()
                self.work[slot] = None

    def result(self, slot: int):
        return self.recv[slot]

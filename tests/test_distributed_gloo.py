"""
The N>1 path on CPU: two ``gloo`` ranks shard a synthetic batch by track and all-gather the (stand-in) smoothed
positions.  No GPU: the filter output is replaced by a deterministic function of the inputs, which is enough to check
the shard boundaries, the independence of a track from its shard, and the gather layout.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_smoothed(hb):
    """Stand-in for sm_mean [N+1][4][B]: depends on the track's own inputs only."""
    N, B = hb.Nmax, hb.B
    out = np.zeros((N + 1, 4, B))
    out[0] = hb.x0
    out[1:, 0] = hb.x0[0] + np.cumsum(hb.dt * hb.sog_rate, axis=0)
    out[1:, 1] = hb.x0[1] + np.cumsum(hb.dt * hb.cog_rate, axis=0)
    return out


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from track_estimators import batch, distributed, synthetic

        lo, hi = distributed.shard_bounds(total, rank, world)
        H, Q, R, P0 = synthetic.example_matrices()
        sb = synthetic.make_batch(hi - lo, nobs=9, gap_h=1.0, seed0=lo)
        hb = batch.pack_uniform(sb, 2, H, Q, R, P0)
        local = torch.from_numpy(_fake_smoothed(hb))
        g = distributed.gather_smoothed_positions(local)
        # the overlapped, double-buffered form must deliver the same tensors for a stream of three batches
        og = distributed.OverlappedGather(local.shape[0], local.shape[2], "cpu")
        slots = [og.launch(local + float(i)) for i in range(3)]
        og.finish()
        assert torch.equal(og.result(slots[2]), g + 2.0) and torch.equal(og.result(slots[1]), g + 1.0)
        q.put((rank, lo, hi, g.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from track_estimators.distributed import shard_bounds

    for n in (1, 7, 8, 100000, 12501):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


@pytest.mark.timeout(120)
def test_two_rank_gather_matches_single_process():
    from track_estimators import batch, synthetic

    total, world = 12, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(total, nobs=9, gap_h=1.0, seed0=0)
    ref = _fake_smoothed(batch.pack_uniform(sb, 2, H, Q, R, P0))  # the whole batch in one process
    for rank, lo, hi, g in res:
        assert g.shape == (world, ref.shape[0], 2, total // world)
        for r in range(world):
            rlo, rhi = r * (total // world), (r + 1) * (total // world)
            assert np.array_equal(g[r], ref[:, :2, rlo:rhi])  # every rank sees every shard, in rank order

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
        assert og.direct == 0  # [N+1][4][B] tensors: lon / lat snapshotted into a send buffer
        # the smoother's own sm_pos output ([N+1][2][B], include/ste.h) is sent as it is: same result, no snapshot
        pos = local[:, :2, :].contiguous()
        slot = og.launch(pos)
        og.finish()
        assert og.direct == 1 and og.send[slot] is not pos and torch.equal(og.result(slot), g)
        assert og.launch_for_pipeline(pos) is None  # no GPU stream to order against on the CPU backend
        og.finish()
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


def _oracle_smoothed(sb, substeps):
    """The real filter + smoother arithmetic on a CPU-sized shard, through the oracle: sm_mean laid out [N+1][4][B]."""
    from oracle import ukf_oracle as orc
    from track_estimators import batch, synthetic

    H, Q, R, P0 = synthetic.example_matrices()
    hb = batch.pack_uniform(sb, substeps, H, Q, R, P0)
    fires = hb.upd_idx.T >= 0
    zidx = np.where(fires, hb.upd_idx.T, 0)
    ridx = np.cumsum(fires, axis=1) - fires
    T = sb.lon.shape[1]
    m, P = orc.forward_batch(hb.x0.T, P0, H, Q, R, hb.dt.T, fires, zidx, ridx, sb.z, sb.sog_rate, sb.cog_rate)
    rr = np.broadcast_to(batch.rts_rate_index(hb.Nmax + 1, T - 1, T), (hb.B, hb.Nmax))
    sm, _ = orc.backward_batch(m, P, Q, hb.dt.T, rr, sb.sog_rate, sb.cog_rate)
    return np.ascontiguousarray(sm.transpose(1, 2, 0))


def _worker_uneven(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from track_estimators import distributed, synthetic

        lo, hi = distributed.shard_bounds(total, rank, world)
        bmax = -(-total // world)
        sb = synthetic.make_batch(hi - lo, nobs=6, gap_h=1.0, seed0=lo)  # seed = global track index
        local = torch.from_numpy(_oracle_smoothed(sb, 2))
        g = distributed.gather_smoothed_positions(local, pad_to=bmax)
        og = distributed.OverlappedGather(local.shape[0], bmax, "cpu")
        slot = og.launch(local)
        og.finish()
        assert torch.equal(og.result(slot), g)
        q.put((rank, distributed.assemble_tracks(g, total).numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_uneven_split_real_arithmetic():
    """13 tracks over 2 ranks (7 + 6): each rank filters and smooths its shard (the oracle stands in for the GPU), the
    shards are padded to the larger one for the all-gather and cut back by assemble_tracks; every rank ends up with the
    positions one process computes for the whole batch, bit for bit (a track does not depend on its shard)."""
    from track_estimators import synthetic

    total, world = 13, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    ref = _oracle_smoothed(synthetic.make_batch(total, nobs=6, gap_h=1.0, seed0=0), 2)[:, :2, :]
    for rank, full in res:
        assert full.shape == ref.shape
        assert np.array_equal(full, ref)

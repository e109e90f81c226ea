"""batch.SmootherPipeline: forward passes and smoothers of consecutive batches side by side (sharing every CU, or on
disjoint CU partitions) give the same bits as running each batch alone, in any interleaving, and a batch is not
resubmitted before its smoother has drained."""
import numpy as np
import pytest

from track_estimators import batch, synthetic

pytestmark = pytest.mark.gpu


def _batch(B, seed0, nobs=40):
    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(B, nobs=nobs, gap_h=1.0, seed0=seed0)
    return batch.pack_uniform(sb, 2, H, Q, R, P0)


def test_pipeline_matches_serial_runs():
    import torch

    hbs = [_batch(300, 0), _batch(300, 1000), _batch(300, 2000)]
    want = []
    for hb in hbs:
        hb.lanes = 1  # the pipeline's forward passes run with one lane per track; same mapping, same bits
        db = batch.DeviceBatch(hb)
        db.run()
        torch.cuda.synchronize()
        want.append((db.sm_mean.clone(), db.sm_cov.clone(), db.fwd_mean.clone()))
    dbs = [batch.DeviceBatch(hb) for hb in hbs]
    pipe = batch.SmootherPipeline("cuda:0", ntracks=300)
    assert pipe.shared and pipe.forward_cus == pipe.smoother_cus  # default: every stream may use every CU
    order = [0, 1, 2, 1, 0, 2, 2, 0, 1]
    for i, k in enumerate(order):
        pipe.submit(dbs[k], final=(i == len(order) - 1))
    pipe.synchronize()
    for db, (sm, sc, fm) in zip(dbs, want):
        assert torch.equal(db.sm_mean, sm) and torch.equal(db.sm_cov, sc) and torch.equal(db.fwd_mean, fm)
        assert not db.status_host().any()
    pipe.close()
    # compute units left to a collective's kernels (multi-GPU runs): fewer CUs for the same work, the same bits
    with batch.SmootherPipeline("cuda:0", ntracks=300, reserve_cus=32) as part:
        assert part.shared and part.reserve_cus == 32
        for db in dbs:
            db.sm_mean.zero_()
        for i, k in enumerate(order):
            part.submit(dbs[k], final=(i == len(order) - 1))
        part.synchronize()
    for db, (sm, sc, fm) in zip(dbs, want):
        assert torch.equal(db.sm_mean, sm) and torch.equal(db.sm_cov, sc)
    with pytest.raises(ValueError):
        batch.SmootherPipeline("cuda:0", ntracks=300, reserve_cus=10_000)


def test_pipeline_hooks_and_timing_events():
    import torch

    hb = _batch(64, 7)
    dbs = [batch.DeviceBatch(hb), batch.DeviceBatch(hb)]
    pipe = batch.SmootherPipeline("cuda:0", forward_cus=160)
    snaps = []
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    pipe.submit(dbs[0], after_smoother=lambda s: snaps.append(dbs[0].sm_mean[:, :2, :].clone()), timing=evs)
    pipe.submit(dbs[1], final=True)
    pipe.synchronize()
    assert evs[0].elapsed_time(evs[1]) > 0 and evs[2].elapsed_time(evs[3]) > 0
    # the hook ran on the smoother stream after the smoother: it saw the finished positions
    assert torch.equal(snaps[0], dbs[0].sm_mean[:, :2, :]) and torch.equal(dbs[0].sm_mean, dbs[1].sm_mean)
    with pytest.raises(ValueError):
        batch.SmootherPipeline("cuda:0", forward_cus=10_000)
    pipe.close()
    # closing is idempotent, a closed pipeline refuses work, and the context manager closes on the way out
    pipe.close()
    assert pipe.closed
    with pytest.raises(RuntimeError):
        pipe.submit(dbs[0])
    ref = dbs[0].sm_mean.clone()
    with batch.SmootherPipeline("cuda:0", forward_cus=160) as pipe2:
        pipe2.submit(dbs[0])  # the event the closed pipeline left on the batch is gone: no wait on a dead stream
        pipe2.submit(dbs[1], final=True)
        pipe2.synchronize()
    assert pipe2.closed and torch.equal(dbs[0].sm_mean, ref)


def test_pipeline_full_size_matches_serial():
    """BASELINE.json configs[1] at full size (10 000 tracks x 500 steps): six pipelined steps over two sets of buffers
    leave exactly the bits one batch run on its own leaves."""
    import torch

    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(10_000, nobs=126, gap_h=1.0, seed0=31)
    hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
    quad = batch.DeviceBatch(hb)  # a batch on its own: the library picks the quad mapping at this size
    quad.run()
    hb.lanes = 1
    ref = batch.DeviceBatch(hb)
    ref.run()
    torch.cuda.synchronize()
    hb.lanes = None
    err = (quad.sm_mean - ref.sm_mean).abs() / ref.sm_mean.abs().clamp_min(1e-12)
    assert float(err.max()) < 1e-7  # the two lane mappings round differently, nothing more (tolerance: 1e-6)
    pipe = batch.SmootherPipeline("cuda:0", ntracks=hb.B)
    assert (pipe.shared, pipe.forward_cus, pipe.smoother_cus, len(pipe.fwd_streams), len(pipe.bwd_streams)) == (True, 256, 256, 7, 6)
    assert pipe.buffers_needed == 14
    dbs = [batch.DeviceBatch(hb) for _ in range(3)]  # fewer sets than streams: resubmission waits for the smoother
    for k in range(8):
        pipe.submit(dbs[k % 3], final=(k == 7))
    pipe.synchronize()
    for db in dbs:
        assert torch.equal(db.fwd_mean, ref.fwd_mean) and torch.equal(db.fwd_cov, ref.fwd_cov)
        assert torch.equal(db.sm_mean, ref.sm_mean) and torch.equal(db.sm_cov, ref.sm_cov)
        assert not db.status_host().any()
    pipe.close()
    # the round-2 split (forward passes and smoothers on disjoint CU partitions) is still there, and gives the same bits
    with batch.SmootherPipeline("cuda:0", ntracks=hb.B, shared=False) as part:
        assert (part.shared, part.forward_cus, part.smoother_cus, len(part.bwd_streams)) == (False, 160, 96, 2)
        for k in range(5):
            part.submit(dbs[k % 3], final=(k == 4))
        part.synchronize()
    for db in dbs:
        assert torch.equal(db.sm_mean, ref.sm_mean) and torch.equal(db.sm_cov, ref.sm_cov)
    # full size against the oracle: filtered AND smoothed, means AND covariances, on 256 of the 10 000 tracks
    from oracle import ukf_oracle as orc

    n = 256
    fires = hb.upd_idx.T[:n] >= 0
    zidx = np.where(fires, hb.upd_idx.T[:n], 0)
    ridx = np.cumsum(fires, axis=1) - fires
    m, P = orc.forward_batch(hb.x0.T[:n], P0, H, Q, R, hb.dt.T[:n], fires, zidx, ridx, sb.z[:n], sb.sog_rate[:n], sb.cog_rate[:n])
    rr = np.broadcast_to(batch.rts_rate_index(501, 125, 126), (n, 500))
    sm, sP = orc.backward_batch(m, P, Q, hb.dt.T[:n], rr, sb.sog_rate[:n], sb.cog_rate[:n])
    for db in (ref, quad):
        res = db.download(track_index=torch.arange(n, device=db.device))
        for name, want, tol in (("means", m, 1e-6), ("means_smoothed", sm, 1e-6)):
            assert float(np.max(np.abs(res[name] - want) / np.maximum(np.abs(want), 1e-12))) < tol, name
        for name, want, tol in (("covs", P, 1e-5), ("covs_smoothed", sP, 1e-5)):
            assert float(np.max(np.abs(res[name] - want) / np.max(np.abs(want), axis=(-1, -2), keepdims=True))) < tol, name


def test_pipeline_config2_shard_size():
    """BASELINE.json configs[2]'s shard (100 000 tracks / 8 GPUs = 12 500 tracks x 500 steps) through the default
    pipeline for that size (six lane-per-track forward passes and six smoothers in flight, sharing the chip): seven
    pipelined steps leave exactly the bits of a batch run on its own, and a sample of tracks matches the oracle."""
    import torch
    from oracle import ukf_oracle as orc

    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(12_500, nobs=126, gap_h=1.0, seed0=87_500)  # the last shard of the 100 000-track job
    hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
    hb.lanes = 1
    ref = batch.DeviceBatch(hb)
    ref.run()
    torch.cuda.synchronize()
    hb.lanes = None
    with batch.SmootherPipeline("cuda:0", ntracks=hb.B) as pipe:
        assert (pipe.shared, len(pipe.fwd_streams), len(pipe.bwd_streams)) == (True, 6, 6)
        dbs = [batch.DeviceBatch(hb) for _ in range(4)]
        for k in range(7):
            pipe.submit(dbs[k % len(dbs)], final=(k == 6))
        pipe.synchronize()
    for db in dbs:
        assert torch.equal(db.sm_mean, ref.sm_mean) and torch.equal(db.sm_cov, ref.sm_cov)
        assert torch.equal(db.fwd_mean, ref.fwd_mean) and not db.status_host().any()
    n = 24
    fires = hb.upd_idx.T[:n] >= 0
    zidx = np.where(fires, hb.upd_idx.T[:n], 0)
    ridx = np.cumsum(fires, axis=1) - fires
    m, P = orc.forward_batch(hb.x0.T[:n], P0, H, Q, R, hb.dt.T[:n], fires, zidx, ridx, sb.z[:n], sb.sog_rate[:n], sb.cog_rate[:n])
    rr = np.broadcast_to(batch.rts_rate_index(501, 125, 126), (n, 500))
    sm, sP = orc.backward_batch(m, P, Q, hb.dt.T[:n], rr, sb.sog_rate[:n], sb.cog_rate[:n])
    res = ref.download(("means_smoothed", "covs_smoothed"), torch.arange(n, device=ref.device))
    got, gotP = res["means_smoothed"], res["covs_smoothed"]
    assert float(np.max(np.abs(got - sm) / np.maximum(np.abs(sm), 1e-12))) < 1e-6
    assert float(np.max(np.abs(gotP - sP) / np.max(np.abs(sP), axis=(-1, -2), keepdims=True))) < 1e-5


def test_pipeline_with_general_matrices_and_ragged_tracks():
    """The shared pipeline with the kernels' general routes: a dense H / R (the 4x4 update instead of the closed form; that
    forward kernel holds more registers than fit beside a smoother wave, so the smoothers have to find SIMDs of their own)
    and tracks of different lengths.  It must finish and leave the bits of the same batches run one after the other."""
    import types

    import torch

    rng = np.random.default_rng(11)
    H = np.array([[1.0, 0.1, 0.0, 0.0], [0.0, 1.0, 0.0, 0.05], [0.0, 0.0, 0.5, 0.0], [0.0, 0.0, 0.0, 0.0]])
    A = rng.normal(size=(4, 4)) * 0.1
    R = A @ A.T + np.diag([0.2, 0.2, 0.5, 30.0])
    Q = np.diag([1e-4, 1e-4, 1e-6, 1e-6])
    P0 = np.eye(4)
    hbs = []
    for seed in (1, 2):
        sb = synthetic.make_batch(200, nobs=40, gap_h=1.0, seed0=1000 * seed)
        tracks, dts, x0s = [], [], []
        for b in range(200):
            T = int(rng.integers(12, 41))
            tracks.append(types.SimpleNamespace(z=sb.z[b][:, :T], dts=sb.dts[b][: T - 1], sog_rate=sb.sog_rate[b][:T],
                                                cog_rate=sb.cog_rate[b][:T]))
            dts.append(np.repeat(sb.dts[b][: T - 1] / 2, 2))
            x0s.append(sb.z[b][:, 0])
        hb = batch.pack_tracks(tracks, dts, x0s, H, Q, R, P0)
        hb.lanes = 1
        hbs.append(hb)
    want = []
    for hb in hbs:
        db = batch.DeviceBatch(hb)
        db.run()
        torch.cuda.synchronize()
        want.append((db.download(), db.status_host()))
    with batch.SmootherPipeline("cuda:0", ntracks=200) as pipe:
        dbs = [batch.DeviceBatch(hb) for hb in hbs]
        for k in range(8):
            pipe.submit(dbs[k % 2], final=(k == 7))
        pipe.synchronize()
    for hb, db, (ref, st) in zip(hbs, dbs, want):
        got = db.download()
        for b in range(hb.B):  # rows past a track's last step are padding (never written)
            n1 = int(hb.nsteps[b]) + 1
            for key in ("means", "covs", "means_smoothed", "covs_smoothed"):
                assert np.array_equal(got[key][b, :n1], ref[key][b, :n1]), (b, key)
        assert np.array_equal(db.status_host(), st) and not (st & 0x1).any()

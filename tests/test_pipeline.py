"""batch.SmootherPipeline: forward passes and smoothers of consecutive batches on disjoint CU partitions give the same
bits as running each batch alone, in any interleaving, and a batch is not resubmitted before its smoother has drained."""
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
        db = batch.DeviceBatch(hb)
        db.run()
        torch.cuda.synchronize()
        want.append((db.sm_mean.clone(), db.sm_cov.clone(), db.fwd_mean.clone()))
    dbs = [batch.DeviceBatch(hb) for hb in hbs]
    pipe = batch.SmootherPipeline("cuda:0", ntracks=300)
    assert pipe.forward_cus % 8 == 0 and pipe.smoother_cus > 0
    order = [0, 1, 2, 1, 0, 2, 2, 0, 1]
    for i, k in enumerate(order):
        pipe.submit(dbs[k], final=(i == len(order) - 1))
    pipe.synchronize()
    for db, (sm, sc, fm) in zip(dbs, want):
        assert torch.equal(db.sm_mean, sm) and torch.equal(db.sm_cov, sc) and torch.equal(db.fwd_mean, fm)
        assert not db.status_host().any()
    pipe.close()


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
    # opt-in: lane-per-track recurrence on the smoother partition (different rounding, same answer)
    ref = dbs[0].sm_mean.clone()
    pipe = batch.SmootherPipeline("cuda:0", forward_cus=160, smoother_lane_per_track=True)
    pipe.submit(dbs[0])
    pipe.submit(dbs[1], final=True)
    pipe.synchronize()
    assert float((dbs[0].sm_mean - ref).abs().max() / ref.abs().max()) < 1e-10
    assert dbs[0].struct.flags & 0x8 == 0  # the flag is not left behind on the batch
    pipe.close()


def test_pipeline_full_size_matches_serial():
    """BASELINE.json configs[1] at full size (10 000 tracks x 500 steps): six pipelined steps over two sets of buffers
    leave exactly the bits one batch run on its own leaves; with the lane-per-track recurrence on the smoother partition
    (what bench.py uses) the smoothed states agree to rounding."""
    import torch

    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(10_000, nobs=126, gap_h=1.0, seed0=31)
    hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
    ref = batch.DeviceBatch(hb)
    ref.run()
    torch.cuda.synchronize()
    pipe = batch.SmootherPipeline("cuda:0", ntracks=hb.B)
    assert (pipe.forward_cus, pipe.smoother_cus) == (160, 96) and pipe.buffers_needed == 5
    dbs = [batch.DeviceBatch(hb) for _ in range(3)]  # fewer sets than streams: resubmission waits for the smoother
    for k in range(8):
        pipe.submit(dbs[k % 3], final=(k == 7))
    pipe.synchronize()
    for db in dbs:
        assert torch.equal(db.fwd_mean, ref.fwd_mean) and torch.equal(db.fwd_cov, ref.fwd_cov)
        assert torch.equal(db.sm_mean, ref.sm_mean) and torch.equal(db.sm_cov, ref.sm_cov)
        assert not db.status_host().any()
    pipe.close()
    pipe = batch.SmootherPipeline("cuda:0", ntracks=hb.B, smoother_lane_per_track=True)
    for k in range(6):
        pipe.submit(dbs[k % 3], final=(k == 5))
    pipe.synchronize()
    for db in dbs:
        assert torch.equal(db.fwd_mean, ref.fwd_mean)
        err = (db.sm_mean - ref.sm_mean).abs() / ref.sm_mean.abs().clamp_min(1e-12)
        assert float(err.max()) < 1e-9
    pipe.close()

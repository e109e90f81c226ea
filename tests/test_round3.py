"""Round-3 behaviour through the C ABI and the host classes: guards around the compact smoother work rows, the
pipeline's stream handling, full-size parity on covariances as well as means."""
import numpy as np
import pytest

MEAN_TOL = 1e-6
COV_TOL = 1e-5


def mean_err(a, ref):
    return float(np.max(np.abs(a - ref) / np.maximum(np.abs(ref), 1e-12)))


def cov_err(a, ref):
    scale = np.max(np.abs(ref), axis=(-1, -2), keepdims=True)
    return float(np.max(np.abs(a - ref) / scale))


def _uniform(B, nobs, s, seed0):
    from track_estimators import batch, synthetic

    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(B, nobs=nobs, gap_h=1.0, seed0=seed0)
    return sb, batch.pack_uniform(sb, s, H, Q, R, P0), (H, Q, R, P0)


@pytest.mark.gpu
def test_fan_constants_off_the_identities_take_the_standalone_smoother():
    """The compact work rows assume weights that sum to one and 2 wi fan_scale = 1.  A fan drawn with scale = n
    (weights_computed = False: fan_scale 4, wi 1/6) breaks the second, so the library must not use the rows: the fused
    call has to return the bits of the stand-alone smoother, and differ from the run with the regular constants."""
    import torch
    from track_estimators import batch

    _, hb, _ = _uniform(130, 20, 2, 77)
    hb.weights_computed = False
    res = []
    for fuse in (True, False):
        db = batch.DeviceBatch(hb, fuse_gains=fuse)
        db.run()
        torch.cuda.synchronize()
        res.append((db.sm_mean.clone(), db.sm_cov.clone(), db.fwd_mean.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    hb.weights_computed = True
    db = batch.DeviceBatch(hb)
    db.run()
    torch.cuda.synchronize()
    assert float((db.fwd_mean - res[0][2]).abs().max()) > 1e-6  # the other scale really is another filter


@pytest.mark.gpu
def test_pipeline_first_use_of_a_buffer_set_is_no_device_barrier():
    """submit() on a DeviceBatch that has never run must not put anything on the legacy default stream (round 2 did, via
    wait_stream(current stream), and every first use drained the whole pipeline).  Observable without a profiler: a
    long job queued on another of the pipeline's streams is still running when submit() of a fresh batch has returned and its own
    forward kernel has finished."""
    import torch
    from track_estimators import batch

    _, hb, _ = _uniform(640, 41, 4, 3)
    dev = torch.device("cuda:0")
    with batch.SmootherPipeline(dev, ntracks=hb.B) as pipe:
        dbs = [batch.DeviceBatch(hb, device=dev) for _ in range(3)]
        torch.cuda.synchronize()
        # one of the pipeline's own CU-masked streams: a blocking stream, i.e. one a default-stream marker waits for
        side = pipe.bwd_streams[1]
        big = torch.zeros(1 << 28, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ev_side = torch.cuda.Event()
        with torch.cuda.stream(side):
            for _ in range(40):
                big.add_(1.0)
            ev_side.record(side)
        done = pipe.submit(dbs[0])
        done.synchronize()
        side_still_running = not ev_side.query()
        pipe.synchronize()
        side.synchronize()
    assert side_still_running, "submit() of a fresh batch waited for unrelated work on another stream"

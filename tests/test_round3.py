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
def test_fan_constants_off_their_usual_relation_still_smooth_right():
    """A fan drawn with scale = n (weights_computed = False: fan_scale 4, wi 1/6, so 2 wi fan_scale = 4/3) is another
    filter than the regular one; the work rows of round 2 silently assumed 2 wi fan_scale = 1 (columns 2-3 of D taken
    from the filtered covariance) and smoothed such a batch wrongly.  The forward pass now forms D from the fan itself:
    the smoother from its rows must agree with the stand-alone smoother, which recomputes everything."""
    import torch
    from track_estimators import batch

    _, hb, _ = _uniform(130, 20, 2, 77)
    hb.weights_computed = False
    res = []
    for fuse in (True, False):
        db = batch.DeviceBatch(hb, fuse_gains=fuse)
        db.run()
        torch.cuda.synchronize()
        assert not db.status_host().any()
        res.append(db.smoothed() + (db.fwd_mean.clone(),))
    assert mean_err(res[0][0], res[1][0]) < 1e-9 and cov_err(res[0][1], res[1][1]) < 1e-9
    hb.weights_computed = True
    db = batch.DeviceBatch(hb)
    db.run()
    torch.cuda.synchronize()
    assert float((db.fwd_mean - res[0][2]).abs().max()) > 1e-6  # the other scale really is another filter


def test_abi_refuses_weights_that_do_not_sum_to_one():
    """w0 + 2 n wi = 1 for every weight0 the reference accepts (unscented.py:125-132); the streamed moments rely on it, so
    anything else is an argument error, not a silently different filter.  (Argument validation only: runs without a GPU.)"""
    import ctypes as C

    from track_estimators._hip import binding

    lib = binding.load()
    s = binding.SteUkfBatchF64()
    s.B, s.Nmax, s.Tmax, s.n = 1, 0, 1, 4
    s.fan_scale, s.w0, s.wi = 3.0, -1.0 / 3.0, 0.2
    eye = (C.c_double * 16)(*([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0]))
    s.H = s.Q = s.R = C.addressof(eye)
    fake = 4096  # never dereferenced: the call must fail before any launch
    for f in ("x0", "P0", "z", "fwd_mean", "fwd_cov", "status"):
        setattr(s, f, fake)
    rc = lib.ste_ukf_forward_f64(C.byref(s), None)
    assert rc == -1 and b"sum to one" in lib.ste_last_error()


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


@pytest.mark.gpu
@pytest.mark.parametrize("B,nobs,s,lanes,robust", [(130, 20, 3, 1, False), (130, 20, 3, 4, False), (70, 12, 6, 0, False),
                                                   (200, 15, 3, 4, True), (96, 30, 5, 1, False)])
def test_work_rows_serve_a_smoother_with_rates_of_its_own(B, nobs, s, lanes, robust):
    """The reference's smoother indexes the repeated rate arrays by step (unscented.py:287-292), which for most sub-step
    counts is not the rate the forward step used.  Speed and heading enter the process model as x + rate * dt, so the
    work rows of the forward pass still serve: x_b[2:4] moves by (rate_rts - rate) dt, P_b by the matching rank-two
    term, D not at all.  Must agree with the stand-alone smoother, which propagates its own fan with its own rates."""
    import torch
    from track_estimators import batch

    sb, hb, _ = _uniform(B, nobs, s, 500 + B)
    assert hb.sog_rate_rts is not None or hb.cog_rate_rts is not None, "this packing shares the rates: nothing to test"
    # make the difference count: rates that really change from observation to observation
    assert float(np.abs(hb.sog_rate_rts - hb.sog_rate).max()) > 1e-3 or float(np.abs(hb.cog_rate_rts - hb.cog_rate).max()) > 1e-3
    hb.lanes, hb.robust = lanes, robust
    res = []
    for fuse in (True, False):
        db = batch.DeviceBatch(hb, fuse_gains=fuse)
        assert (db.rts_work is not None) == fuse
        db.run()
        torch.cuda.synchronize()
        assert not db.status_host().any()
        res.append(db.smoothed())
    assert mean_err(res[0][0], res[1][0]) < 1e-9 and cov_err(res[0][1], res[1][1]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [1, 4])
def test_work_rows_with_recorded_noise_and_rates_of_its_own(lanes):
    """The CLI's default: noise drawn like the reference draws it AND a smoother that reads other rates than the forward
    pass.  x_b carries the recorded noise, P_b does not (unscented.py:319-325), so the b of P_b = C + b b^T is x_b - x_k
    less that noise; the work-row smoother must land on the stand-alone smoother's result."""
    import types

    import torch
    from track_estimators import batch, synthetic

    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(40, nobs=18, gap_h=1.0, seed0=4242)
    rng = np.random.default_rng(9)
    np.random.seed(77)
    tracks, dts, x0s, noise = [], [], [], []
    for b in range(sb.ntracks):
        T = int(rng.integers(8, 19))
        st = types.SimpleNamespace(z=sb.z[b][:, :T], dts=sb.dts[b][: T - 1], sog_rate=sb.sog_rate[b][:T],
                                   cog_rate=sb.cog_rate[b][:T])
        d = np.repeat(st.dts / 3, 3)
        tracks.append(st)
        dts.append(d)
        x0s.append(st.z[:, 0])
        noise.append(batch.draw_reference_noise(Q, R, d, st.dts))
    hb = batch.pack_tracks(tracks, dts, x0s, H, Q, R, P0, noise=noise)
    assert hb.noise_rts is not None and (hb.sog_rate_rts is not None or hb.cog_rate_rts is not None)
    hb.lanes = lanes
    res = []
    for fuse in (True, False):
        db = batch.DeviceBatch(hb, fuse_gains=fuse)
        assert (db.rts_work is not None) == fuse
        db.run()
        torch.cuda.synchronize()
        assert not db.status_host().any()
        res.append(db.download())
    for b in range(hb.B):
        n1 = int(hb.nsteps[b]) + 1
        assert mean_err(res[0]["means_smoothed"][b, :n1], res[1]["means_smoothed"][b, :n1]) < 1e-9
        assert cov_err(res[0]["covs_smoothed"][b, :n1], res[1]["covs_smoothed"][b, :n1]) < 1e-9


@pytest.mark.gpu
def test_one_batch_of_a_hundred_thousand_tracks():
    """BASELINE.json configs[2]'s whole job (100 000 tracks x 500 steps) as ONE batch on one GPU: 1.5e9 work-row doubles = 12 GB,
    i.e. byte offsets three times past 2^32 in every kernel (indices are size_t throughout), 25 GB of HBM in all.  A sample of
    tracks from both ends, the middle and either side of 2^16 against the oracle: filtered and smoothed, means and
    covariances."""
    import torch
    from oracle import ukf_oracle as orc
    from track_estimators import batch, synthetic

    H, Q, R, P0 = synthetic.example_matrices()
    B = 100_000
    sb = synthetic.make_batch(B, nobs=126, gap_h=1.0, seed0=5)
    hb = batch.pack_uniform(sb, 4, H, Q, R, P0)
    db = batch.DeviceBatch(hb)
    assert db.rts_work is not None and db.rts_work.numel() * 8 > 2**32
    db.run()
    torch.cuda.synchronize()
    assert not db.status_host().any()
    sel = np.array([0, 1, 63, 64, 4999, 50_000, 65_535, 65_536, 99_998, 99_999])
    res = db.download(track_index=torch.from_numpy(sel).to(db.device))
    del db
    torch.cuda.empty_cache()
    fires = hb.upd_idx.T[sel] >= 0
    zidx = np.where(fires, hb.upd_idx.T[sel], 0)
    ridx = np.cumsum(fires, axis=1) - fires
    m, P = orc.forward_batch(hb.x0.T[sel], P0, H, Q, R, hb.dt.T[sel], fires, zidx, ridx, sb.z[sel], sb.sog_rate[sel],
                             sb.cog_rate[sel])
    rr = np.broadcast_to(batch.rts_rate_index(501, 125, 126), (len(sel), 500))
    sm, sP = orc.backward_batch(m, P, Q, hb.dt.T[sel], rr, sb.sog_rate[sel], sb.cog_rate[sel])
    assert mean_err(res["means"], m) < MEAN_TOL and mean_err(res["means_smoothed"], sm) < MEAN_TOL
    assert cov_err(res["covs"], P) < COV_TOL and cov_err(res["covs_smoothed"], sP) < COV_TOL

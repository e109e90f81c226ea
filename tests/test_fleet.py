"""Round 4: fleets (windows of one resident batch), the forward pass in time slices, the smoother's sm_pos output.

The reference's batch dimension is its per-ship loop (/root/reference/examples/example_ukf_rts_smoother_batch.py:19-90);
``batch.run_fleet`` is that loop for a fleet of any size.  Everything here is "the same bits as the plain launch":
windows and slices only move launch boundaries, they do not change arithmetic (include/ste.h 0.3.1)."""
import numpy as np
import pytest

from track_estimators import batch, synthetic
from track_estimators._hip import binding

pytestmark = pytest.mark.gpu

HIST = ("means", "covs", "means_smoothed", "covs_smoothed")


def _uniform(B, seed0, nobs=33, substeps=4):
    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(B, nobs=nobs, gap_h=1.0, seed0=seed0)
    return sb, batch.pack_uniform(sb, substeps, H, Q, R, P0)


def _oracle(sb, hb, n):
    from oracle import ukf_oracle as orc

    H, Q, R, P0 = synthetic.example_matrices()
    fires = hb.upd_idx.T[:n] >= 0
    zidx = np.where(fires, hb.upd_idx.T[:n], 0)
    ridx = np.cumsum(fires, axis=1) - fires
    m, P = orc.forward_batch(hb.x0.T[:n], P0, H, Q, R, hb.dt.T[:n], fires, zidx, ridx, sb.z[:n], sb.sog_rate[:n],
                             sb.cog_rate[:n])
    rr = np.broadcast_to(batch.rts_rate_index(hb.Nmax + 1, sb.nobs - 1, sb.nobs), (n, hb.Nmax))
    sm, sP = orc.backward_batch(m, P, Q, hb.dt.T[:n], rr, sb.sog_rate[:n], sb.cog_rate[:n])
    return {"means": m, "covs": P, "means_smoothed": sm, "covs_smoothed": sP}


def test_run_fleet_is_run_batch_bit_for_bit_on_35000_distinct_tracks():
    """35 000 distinct tracks: four windows through the pipelined kernels (uploads and downloads overlapped) give the bits
    of one lane-per-track launch over the whole batch; a 48-track sample agrees with the oracle."""
    sb, hb = _uniform(35_000, 5_000_000)
    hb.lanes = 1
    want = batch.run_batch(hb)
    got = batch.run_fleet(hb, chunk=10_000)
    assert len(batch.fleet_windows(hb.B, 10_000)) == 4
    for k in HIST:
        assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["status"], want["status"]) and not got["status"].any()
    assert np.array_equal(got["nsteps"], want["nsteps"])
    ref = _oracle(sb, hb, 48)
    for k in ("means", "means_smoothed"):
        assert np.max(np.abs(got[k][:48] - ref[k]) / np.maximum(np.abs(ref[k]), 1e-12)) < 1e-6, k  # north_star: 1e-6 relative
    for k in ("covs", "covs_smoothed"):
        err = np.max(np.abs(got[k][:48] - ref[k]), axis=(-1, -2)) / np.max(np.abs(ref[k]), axis=(-1, -2))
        assert err.max() < 1e-5, k  # north_star: 1e-5 per matrix
    # the same fleet, resident: nothing moves, results stay in the DeviceBatch's tensors; time slices change nothing
    import torch

    db = batch.DeviceBatch(hb)
    res = batch.run_fleet(db, chunk=10_000, slices=2)
    assert res["device_batch"] is db and set(res) == {"status", "nsteps", "device_batch"}
    torch.cuda.synchronize()
    d = db.download(HIST)
    for k in HIST:
        assert np.array_equal(d[k], want[k]), k
    part = batch.run_fleet(db, chunk=12_000, outputs=("means_smoothed",))
    assert np.array_equal(part["means_smoothed"], want["means_smoothed"])


def _ragged(B=150, seed=3, substeps=4):
    """Tracks of 5 .. 60 observations with irregular gaps (ragged, length-bucketed by pack_tracks)."""
    rng = np.random.default_rng(seed)
    H, Q, R, P0 = synthetic.example_matrices()
    tracks, dts, x0s = [], [], []
    for b in range(B):
        nobs = int(rng.integers(5, 61))
        sb = synthetic.make_batch(1, nobs=nobs, gap_h=float(rng.choice([0.5, 1.0, 2.0])), seed0=9000 + b)
        st = _Track(sb)
        tracks.append(st)
        dts.append(np.repeat(st.dts / substeps, substeps))
        x0s.append(st.z[:, 0])
    return batch.pack_tracks(tracks, dts, x0s, H, Q, R, P0)


def _zero_outputs(db):
    for t in (db.fwd_mean, db.fwd_cov, db.sm_mean, db.sm_cov):
        t.zero_()


class _Track:
    def __init__(self, sb):
        self.z, self.dts, self.sog_rate, self.cog_rate = sb.z[0], sb.dts[0], sb.sog_rate[0], sb.cog_rate[0]


def test_fleet_of_ragged_tracks_in_small_windows():
    """Length-bucketed ragged tracks, windows of ~64: each window only runs as long as its longest track, results come back
    in the caller's order and equal the single launch's."""
    hb = _ragged()
    assert hb.order is not None
    hb.lanes = 1
    want = batch.run_batch(hb)
    got = batch.run_fleet(hb, chunk=64, slices=2)
    for k in HIST:
        for b in range(hb.B):
            n1 = want["nsteps"][b] + 1
            assert np.array_equal(got[k][b, :n1], want[k][b, :n1]), (k, b)
    assert np.array_equal(got["status"], want["status"]) and np.array_equal(got["nsteps"], want["nsteps"])


@pytest.mark.parametrize("lanes,packed", [(1, True), (1, False), (4, False)])
def test_time_slices_are_bit_identical(lanes, packed):
    """include/ste.h step_begin / step_end: a forward pass issued as 2 .. 8 launches over consecutive step ranges leaves the
    histories, the smoother's work rows and the status of the single launch, bit for bit -- also for ragged tracks, recorded
    noise, and a track whose prior is indefinite (its square root is clamped at step 0: the `first bad step` word and the
    extra work-row columns have to survive the slice boundaries)."""
    import torch

    sb, hb = _uniform(200, 77, nobs=76, substeps=4)  # 300 steps
    hb.lanes = lanes
    hb.nsteps = hb.nsteps.copy()
    hb.nsteps[5], hb.nsteps[6], hb.nsteps[70], hb.nsteps[71] = 64, 63, 130, 0
    P0 = np.repeat(np.eye(4)[None], hb.B, 0)
    P0[9] = np.diag([1.0, -0.3, 1.0, 1.0])  # indefinite: clamped
    hb.P0 = np.ascontiguousarray(P0.reshape(hb.B, 16).T)
    rng = np.random.default_rng(0)
    hb.noise_pred = 1e-3 * rng.standard_normal((hb.Nmax, 4, hb.B))
    hb.noise_upd = 1e-3 * rng.standard_normal((hb.Nmax + 1, 4, hb.B))
    hb.noise_rts = 1e-3 * rng.standard_normal((hb.Nmax, 4, hb.B))
    for noise in (True, False):
        if not noise:
            hb.noise_pred = hb.noise_upd = hb.noise_rts = None
        one = batch.DeviceBatch(hb, packed_cov=packed)
        _zero_outputs(one)  # rows past a short track's end are never written
        one.run()
        torch.cuda.synchronize()
        assert one.status_host()[9] & binding.STE_STATUS_CLAMPED
        for slices in (2, 3, 8):
            db = batch.DeviceBatch(hb, packed_cov=packed)
            _zero_outputs(db)
            db.rts_work.fill_(float("nan"))
            one_work = one.rts_work
            db.forward(slices=slices)
            db.backward()
            torch.cuda.synchronize()
            assert len(batch.DeviceBatch.slice_bounds(hb.Nmax, slices)) > 1
            for name in ("fwd_mean", "fwd_cov", "sm_mean", "sm_cov", "status"):
                a, b = getattr(db, name), getattr(one, name)
                live = torch.ones_like(a, dtype=torch.bool)
                assert torch.equal(torch.where(torch.isnan(a), torch.zeros_like(a), a),
                                   torch.where(torch.isnan(b), torch.zeros_like(b), b)) and \
                    torch.equal(torch.isnan(a) & live, torch.isnan(b) & live), (name, slices, noise)
            # the last row of the work buffer: first bad step per track
            assert torch.equal(db.rts_work[-1], one_work[-1])


def test_quad_slices_with_packed_covariances_are_refused():
    _, hb = _uniform(64, 1, nobs=40, substeps=4)
    hb.lanes = 4
    db = batch.DeviceBatch(hb, packed_cov=True)
    with pytest.raises(binding.SteError, match="lane-per-track"):
        db.forward(slices=2)
    assert db.struct.step_begin == 0 and db.struct.step_end == 0  # the struct is left as it was


def test_windows_write_the_fleets_tensors_in_place():
    """DeviceBatch.window: same tensors, offset pointers, track_stride = the fleet's width.  Windows run in any order and
    mapping; tracks outside a window are not touched."""
    import torch

    _, hb = _uniform(500, 21, nobs=30, substeps=2)
    hb.lanes = 1
    whole = batch.DeviceBatch(hb, sm_pos=True)
    whole.run()
    torch.cuda.synchronize()
    fleet = batch.DeviceBatch(hb, sm_pos=True)
    for t in (fleet.fwd_mean, fleet.fwd_cov, fleet.sm_mean, fleet.sm_cov):
        t.fill_(-7.0)
    w = fleet.window(128, 333)
    assert w.struct.track_stride == 500 and w.struct.B == 205 and w.parent is fleet
    w.run()
    torch.cuda.synchronize()
    for name in ("fwd_mean", "fwd_cov", "sm_mean", "sm_cov", "sm_pos"):
        a, b = getattr(fleet, name), getattr(whole, name)
        assert torch.equal(a[..., 128:333], b[..., 128:333]), name
        if name != "sm_pos":
            assert bool((a[..., :128] == -7.0).all()) and bool((a[..., 333:] == -7.0).all()), name
    assert bool((fleet.sm_pos[..., :128] == 0).all())
    # a window of a window, and the rest in the quad mapping's windows
    fleet.window(0, 128).window(64, 128).run()
    fleet.window(0, 64).run()
    fleet.window(333, 500).run()
    torch.cuda.synchronize()
    for name in ("fwd_mean", "fwd_cov", "sm_mean", "sm_cov", "sm_pos", "status"):
        assert torch.equal(getattr(fleet, name), getattr(whole, name)), name
    with pytest.raises(ValueError):
        fleet.window(10, 501)


def test_sm_pos_is_the_position_block_of_sm_mean():
    """The optional [N+1][2][B] output the multi-GPU exchange sends: rows 0 .. nsteps of every track equal sm_mean[:, :2];
    rows past a short track's end stay zero.  Both smoothers write it."""
    import torch

    hb = _ragged(B=70, seed=5)
    for fuse in (True, False):
        db = batch.DeviceBatch(hb, sm_pos=True, fuse_gains=fuse)
        db.run()
        torch.cuda.synchronize()
        sm, pos, ns = db.sm_mean.cpu().numpy(), db.sm_pos.cpu().numpy(), hb.nsteps
        for b in range(hb.B):
            assert np.array_equal(pos[: ns[b] + 1, :, b], sm[: ns[b] + 1, :2, b]), (fuse, b)
            assert not pos[ns[b] + 1:, :, b].any()


def test_pipeline_orders_itself_behind_work_issued_outside_it():
    """a batch touched on a stream of the caller's (run / forward / backward) and then submitted to the
    pipeline -- the pipelined forward pass waits for that work instead of overwriting histories it still reads."""
    import torch

    _, hb = _uniform(2048, 400, nobs=60, substeps=4)
    hb.lanes = 1
    ref = batch.DeviceBatch(hb)
    ref.run()
    torch.cuda.synchronize()
    db = batch.DeviceBatch(hb)
    side = torch.cuda.Stream()
    with batch.SmootherPipeline("cuda:0", ntracks=2048, slices=2) as pipe:
        for _ in range(3):
            with torch.cuda.stream(side):
                db.run()  # recorded as the batch's last use
                snap = db.sm_mean.clone()
            assert db._last_use is not None
            pipe.submit(db, final=True)
            assert db._last_use is None
            pipe.synchronize()
            side.synchronize()
            assert torch.equal(snap, ref.sm_mean) and torch.equal(db.sm_mean, ref.sm_mean)


@pytest.mark.parametrize("kind", ["uniform", "ragged-real-rates", "noise", "clamped-prior-all-eig"])
def test_two_kernel_smoother_is_the_one_kernel_smoother_bit_for_bit(kind):
    """Small batches smooth in two kernels -- x_b, P_b and the gain K = D pinv(P_b) of EVERY step at once (they depend on the
    forward pass alone, unscented.py:297-333), then a recurrence that only loads them -- because a few waves of the one-kernel
    smoother are a latency chain (include/ste.h ``tuning`` bits 9 / 10, csrc/ste_kernels.hip launch_backward).  Same device
    functions, same bits: histories, sm_pos, status; and the call stays repeatable although the work rows hold gains
    afterwards."""
    import torch

    tun = 0
    if kind == "uniform":
        _, hb = _uniform(300, 31, nobs=40, substeps=4)
    elif kind == "ragged-real-rates":
        # thirds of the gaps: most float-equality triggers miss (kalman_filter.py:98-101), so the smoother's rate indexing
        # (unscented.py:287-292) differs from the forward pass's -> sog_rate_rts, the kShift kernels
        hb = _ragged(B=90, seed=8, substeps=3)
        assert hb.sog_rate_rts is not None
    elif kind == "noise":
        _, hb = _uniform(130, 5, nobs=30, substeps=2)
        rng = np.random.default_rng(1)
        hb.noise_pred = 1e-3 * rng.standard_normal((hb.Nmax, 4, hb.B))
        hb.noise_upd = 1e-3 * rng.standard_normal((hb.Nmax + 1, 4, hb.B))
        hb.noise_rts = 1e-3 * rng.standard_normal((hb.Nmax, 4, hb.B))
    else:
        _, hb = _uniform(70, 9, nobs=30, substeps=4)
        P0 = np.repeat(np.eye(4)[None], hb.B, 0)
        P0[3] = np.diag([1.0, -0.3, 1.0, 1.0])
        hb.P0 = np.ascontiguousarray(P0.reshape(hb.B, 16).T)
        tun = 0x100
    outs = {}
    for form in (0x400, 0x200, 0x200 | 0x800):  # one kernel; gains + quad-per-track recurrence; gains + lane-per-track recurrence
        db = batch.DeviceBatch(hb, tuning=tun | form, sm_pos=True)
        _zero_outputs(db)
        db.run()
        torch.cuda.synchronize()
        outs[form] = [t.clone() for t in (db.sm_mean, db.sm_cov, db.sm_pos, db.status, db.fwd_mean)]
        if form & 0x200:
            db.sm_mean.zero_()
            db.backward()  # again, on work rows that now hold gains
            torch.cuda.synchronize()
            assert torch.equal(db.sm_mean, outs[form][0])
            assert bool((db.rts_work[-1] < 0).all())  # the marker: first-bad word stored negated
    for form in (0x200, 0x200 | 0x800):
        for a, b in zip(outs[0x400], outs[form]):
            assert torch.equal(torch.nan_to_num(a.double(), nan=-1.0), torch.nan_to_num(b.double(), nan=-1.0))


def test_run_fleet_forward_only_and_selected_outputs():
    """``smooth=False`` (the filter alone, no smoother buffers) and ``outputs`` (bring back only what is asked for) through
    windows, against the single launch."""
    _, hb = _uniform(1000, 77_000, nobs=20, substeps=4)
    hb.lanes = 1
    want = batch.run_batch(hb, smooth=False)
    got = batch.run_fleet(hb, chunk=256, smooth=False, outputs=("means",))
    assert set(got) == {"means", "status", "nsteps", "device_batch"} and got["device_batch"].sm_mean is None
    assert np.array_equal(got["means"], want["means"]) and np.array_equal(got["status"], want["status"])
    both = batch.run_fleet(hb, chunk=256, outputs=("covs_smoothed", "means"))
    full = batch.run_batch(hb)
    assert np.array_equal(both["covs_smoothed"], full["covs_smoothed"]) and np.array_equal(both["means"], full["means"])
    with pytest.raises(ValueError):
        batch.run_fleet(got["device_batch"], chunk=256)  # built without smoothed outputs


def test_windows_offset_every_per_track_array():
    """Windows of a fleet with per-track priors, recorded noise, the robust update and a ragged tail: every per-track pointer of
    the batch struct has to move with the window (P0 [16][B], the three noise arrays, nsteps, status), not only the histories."""
    _, hb = _uniform(700, 123, nobs=24, substeps=4)
    hb.lanes = 1
    rng = np.random.default_rng(2)
    P0 = np.repeat(np.eye(4)[None], hb.B, 0) * rng.uniform(0.5, 2.0, (hb.B, 1, 1))
    hb.P0 = np.ascontiguousarray(P0.reshape(hb.B, 16).T)
    hb.noise_pred = 1e-3 * rng.standard_normal((hb.Nmax, 4, hb.B))
    hb.noise_upd = 1e-3 * rng.standard_normal((hb.Nmax + 1, 4, hb.B))
    hb.noise_rts = 1e-3 * rng.standard_normal((hb.Nmax, 4, hb.B))
    hb.nsteps = hb.nsteps.copy()
    hb.nsteps[rng.integers(0, hb.B, 60)] = rng.integers(0, hb.Nmax, 60)
    hb.robust = True
    want = batch.run_batch(hb)
    got = batch.run_fleet(hb, chunk=192, slices=2)
    for k in HIST:
        for b in range(hb.B):
            n1 = hb.nsteps[b] + 1
            assert np.array_equal(got[k][b, :n1], want[k][b, :n1]), (k, b)
    assert np.array_equal(got["status"], want["status"])

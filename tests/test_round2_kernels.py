"""Round-2 kernel behaviour through the C ABI: the smoother from the forward pass's work rows (repeatable, both gain routes,
ragged batches), the robust update pinned to reference-run fixtures in both lane mappings, per-call lane flags and the
upd_idx precondition."""
import os
import types

import numpy as np
import pytest
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

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


def test_backward_is_repeatable():
    """ste_urtss_backward_f64 only reads the forward pass's work rows: forward(); backward(); backward() leaves the bits
    of forward(); backward() (round 1 turned D into K in place, so a second call returned wrong states)."""
    import torch
    from track_estimators import batch

    _, hb, _ = _uniform(200, 30, 2, 5)
    db = batch.DeviceBatch(hb)
    db.forward()
    db.backward()
    torch.cuda.synchronize()
    m1, c1 = db.sm_mean.clone(), db.sm_cov.clone()
    db.sm_mean.zero_()
    db.sm_cov.zero_()
    db.backward()
    torch.cuda.synchronize()
    assert torch.equal(db.sm_mean, m1) and torch.equal(db.sm_cov, c1)
    assert not db.status_host().any()


@pytest.mark.parametrize("tuning", [0, 0x100], ids=lambda t: f"tuning={t:#x}")
def test_smoother_gain_routes(tuning):
    """Both routes to the smoother gain K = D pinv(P_b) (by factorisation with the eigenvalue route as fallback, or all by
    the eigenvalue route: ste_ukf_batch_f64.tuning bit 8) against the oracle on a batch that does not fill its last wave,
    and against each other to rounding."""
    import torch
    from oracle import ukf_oracle as orc
    from track_estimators import batch

    sb, hb, (H, Q, R, P0) = _uniform(150, 27, 4, 900)  # N = 104: not a multiple of 3
    db = batch.DeviceBatch(hb, tuning=tuning)
    db.run()
    torch.cuda.synchronize()
    assert not db.status_host().any()
    sm, sP = db.smoothed()
    ref = batch.DeviceBatch(hb)
    ref.run()
    torch.cuda.synchronize()
    rm, rP = ref.smoothed()
    assert mean_err(sm, rm) < 1e-9 and cov_err(sP, rP) < 1e-9
    n = 6
    fires = hb.upd_idx.T[:n] >= 0
    zidx = np.where(fires, hb.upd_idx.T[:n], 0)
    ridx = np.cumsum(fires, axis=1) - fires
    m, P = orc.forward_batch(hb.x0.T[:n], P0, H, Q, R, hb.dt.T[:n], fires, zidx, ridx, sb.z[:n], sb.sog_rate[:n], sb.cog_rate[:n])
    rr = np.broadcast_to(batch.rts_rate_index(hb.Nmax + 1, 26, 27), (n, hb.Nmax))
    om, oP = orc.backward_batch(m, P, Q, hb.dt.T[:n], rr, sb.sog_rate[:n], sb.cog_rate[:n])
    assert mean_err(sm[:n], om) < MEAN_TOL and cov_err(sP[:n], oP) < COV_TOL


def test_smoother_singular_pb_falls_back_to_pinv():
    """Q = 0 and a prior that knows speed and heading exactly: P_b has rank 2, the factorisation's pivots vanish and every
    gain must come from the eigenvalue route with NumPy's rank cutoff (np.linalg.pinv, unscented.py:333) -- checked
    against the per-track oracle, which calls np.linalg.pinv itself, and against the all-eigenvalue shape."""
    from oracle import ukf_oracle as orc
    from track_estimators import batch, synthetic

    H, _, R, _ = synthetic.example_matrices()
    Q = np.zeros((4, 4))
    P0 = np.diag([1.0, 1.0, 0.0, 0.0])
    sb = synthetic.make_batch(5, nobs=8, gap_h=1.0, seed0=321)
    tracks = [types.SimpleNamespace(z=sb.z[b], dts=sb.dts[b], sog_rate=sb.sog_rate[b], cog_rate=sb.cog_rate[b]) for b in range(5)]
    dts = [np.repeat(sb.dts[b] / 2, 2) for b in range(5)]
    hb = batch.pack_tracks(tracks, dts, [sb.z[b][:, 0] for b in range(5)], H, Q, R, P0)
    out = batch.run_batch(hb)
    db = batch.DeviceBatch(hb, tuning=0x100)
    db.run()
    em, eP = db.smoothed()
    assert not (out["status"] & 0x1).any()
    assert np.array_equal(out["means_smoothed"], em) and np.array_equal(out["covs_smoothed"], eP)
    for b in range(5):
        m, P = orc.forward_track(sb.z[b][:, 0], P0, H, Q, R, dts[b], sb.dts[b], sb.z[b], sb.sog_rate[b], sb.cog_rate[b])
        sm, sP = orc.backward_track(m, P, Q, dts[b], len(sb.dts[b]), sb.sog_rate[b], sb.cog_rate[b])
        assert mean_err(out["means_smoothed"][b], sm) < MEAN_TOL
        assert np.max(np.abs(out["covs_smoothed"][b] - sP)) < COV_TOL * max(np.max(np.abs(sP)), 1.0)


def test_smoother_ragged_lengths_in_one_workgroup():
    """Tracks of 0 .. 40 steps inside the same 64-track workgroup (no length bucketing): each one's smoothed history is
    the one it gets in a batch of its own."""
    from track_estimators import batch, synthetic

    H, Q, R, P0 = synthetic.example_matrices()
    sb = synthetic.make_batch(70, nobs=21, gap_h=1.0, seed0=4242)
    rng = np.random.default_rng(1)
    cut = rng.integers(1, 21, size=70)  # observations kept per track
    cut[:3] = [1, 2, 21]
    tracks, dts, x0s = [], [], []
    for b in range(70):
        T = int(cut[b])
        tracks.append(types.SimpleNamespace(z=sb.z[b][:, :T], dts=sb.dts[b][: T - 1], sog_rate=sb.sog_rate[b][:T],
                                            cog_rate=sb.cog_rate[b][:T]))
        dts.append(np.repeat(sb.dts[b][: T - 1] / 2, 2))
        x0s.append(sb.z[b][:, 0])
    hb = batch.pack_tracks(tracks, dts, x0s, H, Q, R, P0, bucket_by_length=False)
    out = batch.run_batch(hb)
    assert not out["status"].any()
    for b in (0, 1, 2, 17, 40, 69):
        one = batch.run_batch(batch.pack_tracks([tracks[b]], [dts[b]], [x0s[b]], H, Q, R, P0))
        n1 = len(dts[b]) + 1
        assert np.array_equal(out["means_smoothed"][b, :n1], one["means_smoothed"][0])
        assert np.array_equal(out["covs_smoothed"][b, :n1], one["covs_smoothed"][0])


@pytest.mark.parametrize("lanes", [1, 4], ids=["lane-per-track", "quad-per-track"])
def test_robust_update_vs_reference_runs(lanes):
    """STE_FLAG_ROBUST in both lane mappings against whole tracks produced by the reference with its check_robustness call
    site enabled (tests/golden/robust.npz: gross outliers injected, 1 and 2 sub-steps), forward and smoothed."""
    from track_estimators import batch

    g = np.load(os.path.join(GOLDEN, "robust.npz"))
    for ci in range(int(g["nruns"])):
        z = g[f"run{ci}_z"]
        tr = types.SimpleNamespace(z=z, dts=g[f"run{ci}_dts"], sog_rate=g[f"run{ci}_sog_rate"], cog_rate=g[f"run{ci}_cog_rate"])
        hb = batch.pack_tracks([tr], [g[f"run{ci}_dt"]], [z[:, 0]], g["H"], g["Q"], g["R"], g["P0"])
        hb.robust, hb.lanes = True, lanes
        out = batch.run_batch(hb)
        assert not (out["status"][0] & ~0x8)
        assert mean_err(out["means"][0], g[f"run{ci}_means"]) < MEAN_TOL
        assert cov_err(out["covs"][0], g[f"run{ci}_covs"]) < COV_TOL
        assert mean_err(out["means_smoothed"][0], g[f"run{ci}_means_smoothed"]) < MEAN_TOL
        assert cov_err(out["covs_smoothed"][0], g[f"run{ci}_covs_smoothed"]) < COV_TOL
        # and the robust path really was taken: the plain filter is degrees away on these tracks
        assert np.abs(out["means"][0][:, 0] - g[f"run{ci}_plain_means"][:, 0]).max() > 1.0


def test_check_robustness_method_vs_reference(monkeypatch, capsys):
    """UnscentedKalmanFilter.check_robustness (device terms, host loop, the reference's prints) against direct calls of
    the reference's method with its noise draws zeroed."""
    from track_estimators.kalman_filters.unscented import UnscentedKalmanFilter

    monkeypatch.setattr(np.random, "normal", lambda loc=0.0, scale=1.0, size=None: np.zeros(size))
    g = np.load(os.path.join(GOLDEN, "robust.npz"))
    for i in range(int(g["cr_dense_from"])):
        u = UnscentedKalmanFilter(H=g["H"], Q=g["Q"], R=g["cr_R"][i], P=g["cr_P"][i], x0=g["cr_x"][i])
        Rr = u.check_robustness(g["cr_z"][i].reshape(-1, 1), g["cr_P"][i], g["cr_R"][i])
        np.testing.assert_allclose(Rr, g["cr_Rout"][i], rtol=1e-9, atol=0)
        assert len(capsys.readouterr().out.splitlines()) == int(g["cr_iters"][i]) + 1  # one print per evaluation


def test_lane_flags_and_bad_update_index():
    """STE_FLAG_LANES_1 / _4 are per call (no process-global knob) and exclude each other; an upd_idx >= Tmax skips that
    update and sets STE_STATUS_BAD_INDEX instead of reading past z."""
    import ctypes as C

    import torch
    from track_estimators import batch
    from track_estimators._hip import binding

    _, hb, _ = _uniform(40, 9, 1, 77)
    res = {}
    for lanes in (1, 4):
        hb.lanes = lanes
        db = batch.DeviceBatch(hb)
        assert bool(db.struct.flags & binding.STE_FLAG_LANES_1) == (lanes == 1)
        assert bool(db.struct.flags & binding.STE_FLAG_LANES_4) == (lanes == 4)
        db.run()
        torch.cuda.synchronize()
        res[lanes] = db.sm_mean.clone()
    assert float(((res[1] - res[4]).abs() / res[1].abs().clamp_min(1e-12)).max()) < 1e-8
    db.struct.flags |= binding.STE_FLAG_LANES_1 | binding.STE_FLAG_LANES_4
    lib = binding.load()
    assert lib.ste_ukf_forward_f64(C.byref(db.struct), None) == -1 and b"exclude" in lib.ste_last_error()
    for lanes in (1, 4):
        hb.lanes = lanes
        db = batch.DeviceBatch(hb)
        db.t["upd_idx"][3, 5] = hb.Tmax  # one step of track 5 points past the last observation column
        db.t["upd_idx"][6, 9] = hb.Tmax + 1000
        db.run()
        torch.cuda.synchronize()
        st = db.status_host()
        assert st[5] == binding.STE_STATUS_BAD_INDEX and st[9] == binding.STE_STATUS_BAD_INDEX
        assert not np.delete(st, [5, 9]).any()
        assert torch.isfinite(db.sm_mean).all()

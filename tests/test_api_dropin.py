"""
The drop-in Python surface (same import paths / names as the reference package) on the GPU.
The first three tests restate the reference's own unit tests (tests/test_unscented_kf.py:24-87, a 2-state filter);
the rest pin ``run`` / ``run_rts_smoother`` / ``predict`` / ``update`` / ``geodetic_dynamics`` to the golden vectors.
"""
import copy
import os

import numpy as np
import pytest
from conftest import GOLDEN, load_cases

pytestmark = pytest.mark.gpu

CSV = os.path.join(GOLDEN, "ship_01203823.csv")


@pytest.fixture
def ukf2():
    from track_estimators.kalman_filters.unscented import UnscentedKalmanFilter

    rng = np.random.default_rng(5)
    P = np.diag(rng.uniform(0, 1, 2))
    x = rng.uniform(0, 1, 2).reshape(-1, 1)
    return UnscentedKalmanFilter(H=np.diag([1, 1]), P=P, x0=x)


def test_sigma_points_mean_and_cov(ukf2):
    ukf2.compute_sigma_points()
    assert np.allclose(ukf2.x[:, 0], np.mean(ukf2.sigma_points, axis=1))
    assert np.allclose(ukf2.P, np.cov(ukf2.sigma_points))


def test_weights(ukf2):
    ukf2.compute_weights()
    assert np.isclose(np.trace(ukf2.weights), 1.0)
    assert -1.0 < ukf2.weights[0, 0] < 1.0
    assert np.count_nonzero(ukf2.weights - np.diag(np.diagonal(ukf2.weights))) == 0


def test_weighted_sigma_points(ukf2):
    ukf2.compute_weights()
    ukf2.compute_sigma_points()
    assert np.allclose(ukf2.x[:, 0], np.sum(np.dot(ukf2.sigma_points, ukf2.weights), axis=1))
    res = ukf2.sigma_points - ukf2.x
    assert np.allclose(ukf2.P, np.dot(np.dot(res, ukf2.weights), res.T))


def _ship_track(c):
    from track_estimators.ship_track import ShipTrack

    st = ShipTrack()
    st.lon, st.lat, st.dts = c["z"][0].copy(), c["z"][1].copy(), c["dts"].copy()
    st.sog, st.cog = c["sog"].copy(), c["cog"].copy()
    st.sog_rate, st.cog_rate = c["sog_rate"].copy(), c["cog_rate"].copy()
    st.z = c["z"].copy()
    return st


def mean_err(a, ref):
    return float(np.max(np.abs(a - ref) / np.maximum(np.abs(ref), 1e-12)))


def cov_err(a, ref):
    return float(np.max(np.abs(a - ref) / np.max(np.abs(ref), axis=(-1, -2), keepdims=True)))


@pytest.mark.parametrize("name,i,seed", [("ukf_ship_01203823.npz", 0, None), ("ukf_ship_01203823.npz", 1, 2024),
                                         ("ukf_synthetic.npz", 0, None), ("ukf_synthetic.npz", 5, 1005),
                                         ("ukf_synthetic.npz", 8, 1008)])
def test_run_and_smoother_like_the_reference(name, i, seed):
    """ukf.run + ukf.run_rts_smoother through the class API.  For the replay cases the global NumPy generator is
    seeded like the golden script seeded its RandomState: the class draws in the reference's call order, so the noise
    stream -- and therefore the whole history -- matches."""
    from track_estimators.kalman_filters.non_linear_process import geodetic_dynamics
    from track_estimators.kalman_filters.unscented import UnscentedKalmanFilter

    c = load_cases(name)[i]
    st = _ship_track(c)
    ukf = UnscentedKalmanFilter(H=c["H"], Q=c["Q"], R=c["R"], P=c["P0"], x0=st.z[:, 0].reshape(-1, 1).copy(),
                                non_linear_process=geodetic_dynamics)
    if seed is None:
        ukf.inject_noise = False
    else:
        np.random.seed(seed)
    N = len(c["dt"])
    means, covs = ukf.run(nsteps=N, dt=c["dt"], ship_track=st)
    assert means.shape == (N + 1, 4) and covs.shape == (N + 1, 4, 4)
    assert mean_err(means, c["means"]) < 1e-6 and cov_err(covs, c["covs"]) < 1e-5
    assert ukf.x.shape == (4, 1) and len(ukf.means) == N + 1 and ukf.time == pytest.approx(np.sum(c["dt"]))
    st2 = copy.deepcopy(st)
    sm, sP = ukf.run_rts_smoother(ship_track=st2)
    assert mean_err(sm, c["means_smoothed"]) < 1e-6 and cov_err(sP, c["covs_smoothed"]) < 1e-5
    # the reference expands the ShipTrack's rates in place (unscented.py:287-292)
    rep = int((N + 1) / len(st.dts))
    assert len(st2.sog_rate) == rep * len(st.sog_rate)


def test_error_behaviour():
    from track_estimators.kalman_filters.non_linear_process import geodetic_dynamics
    from track_estimators.kalman_filters.unscented import UnscentedKalmanFilter

    with pytest.raises(ValueError):
        UnscentedKalmanFilter()
    c = load_cases("ukf_synthetic.npz")[7]
    st = _ship_track(c)
    ukf = UnscentedKalmanFilter(H=c["H"], Q=c["Q"], R=c["R"], P=c["P0"], x0=c["x0"], non_linear_process=geodetic_dynamics)
    with pytest.raises(AssertionError):
        ukf.run(nsteps=3, dt=c["dt"], ship_track=st)
    with pytest.raises(AssertionError):
        ukf.compute_weights(weight0=1.5)
    other = UnscentedKalmanFilter(H=c["H"], x0=c["x0"], non_linear_process=lambda x, **kw: x)
    with pytest.raises(NotImplementedError):
        other.predict(dt=1.0, c=None)
    with pytest.raises(AssertionError):
        UnscentedKalmanFilter(H=c["H"]).predict(dt=1.0)


def test_predict_update_single_steps():
    from track_estimators.kalman_filters.non_linear_process import geodetic_dynamics
    from track_estimators.kalman_filters.unscented import UnscentedKalmanFilter

    k = np.load(os.path.join(GOLDEN, "kats.npz"))
    for i in range(16):
        u = UnscentedKalmanFilter(H=k["H"], Q=k["Q"], R=k["R"], P=k["pr_P"][i], x0=k["pr_x"][i],
                                  non_linear_process=geodetic_dynamics)
        u.inject_noise = False
        u.predict(dt=k["pr_dt"][i], c=None, sog_rate=k["pr_sr"][i], cog_rate=k["pr_cr"][i])
        assert mean_err(u.x[:, 0], k["pr_xo"][i]) < 1e-9 and cov_err(u.P, k["pr_Po"][i]) < 1e-9
        u.update(k["up_z"][i].copy())
        assert mean_err(u.x[:, 0], k["up_xo"][i]) < 1e-9 and cov_err(u.P, k["up_Po"][i]) < 1e-9


def test_geodetic_dynamics_function():
    from track_estimators.kalman_filters.non_linear_process import geodetic_dynamics

    k = np.load(os.path.join(GOLDEN, "kats.npz"))
    for i in range(0, 64, 7):
        y = geodetic_dynamics(k["gd_x"][i], None, k["gd_dt"][i], k["gd_sr"][i], k["gd_cr"][i])
        np.testing.assert_allclose(y, k["gd_y"][i], rtol=1e-13, atol=1e-12)
    # 2-entry state: the reference truncates the result to x.shape[0] and needs c to supply speed/heading
    y2 = geodetic_dynamics(k["gd_x"][0][:2], k["gd_x"][0][2:], k["gd_dt"][0], k["gd_sr"][0], k["gd_cr"][0])
    np.testing.assert_allclose(y2, k["gd_y"][0][:2], rtol=1e-13)


def test_second_run_appends_and_keeps_time():
    """Like the reference, a second ``run`` on the same object appends to the history, restarts the update index at 0
    and keeps accumulating ``self.time`` (so its float-equality trigger sees offset times)."""
    from track_estimators.kalman_filters.non_linear_process import geodetic_dynamics
    from track_estimators.kalman_filters.unscented import UnscentedKalmanFilter

    g = np.load(os.path.join(GOLDEN, "two_runs.npz"))
    st = _ship_track({k: g[k] for k in ("z", "dts", "sog", "cog", "sog_rate", "cog_rate")})
    ukf = UnscentedKalmanFilter(H=np.diag([1, 1, 0, 0]), Q=g["Q"], R=g["R"], P=g["P0"],
                                x0=st.z[:, 0].reshape(-1, 1).copy(), non_linear_process=geodetic_dynamics)
    ukf.inject_noise = False
    m1, c1 = ukf.run(nsteps=len(g["dt1"]), dt=g["dt1"], ship_track=st)
    assert mean_err(m1, g["means1"]) < 1e-9 and cov_err(c1, g["covs1"]) < 1e-9
    m2, c2 = ukf.run(nsteps=len(g["dt2"]), dt=g["dt2"], ship_track=st)
    assert m2.shape == g["means2"].shape == (len(g["dt1"]) + len(g["dt2"]) + 2, 4)
    assert mean_err(m2, g["means2"]) < 1e-9 and cov_err(c2, g["covs2"]) < 1e-9
    assert ukf.time == float(g["time_end"])

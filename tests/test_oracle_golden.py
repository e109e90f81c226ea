"""
Pins the oracle (oracle/ukf_oracle.py) against vectors produced by running the reference itself
(tests/golden/make_golden.py).  CPU only.
"""
import numpy as np
import pytest
from conftest import load_cases

from oracle import ukf_oracle as orc

ALL = [("ukf_synthetic.npz", i) for i in range(10)] + [("ukf_edge.npz", i) for i in range(3)] + [
    ("ukf_ship_01203823.npz", i) for i in range(2)
]
# reference runs with cond(P_b) from 1e2 to 3e13 (make_golden.illcond_cases): the same-call-order restatement stays bit
# exact on all of them; a restatement with another order of arithmetic cannot promise 1e-6 past cond ~ 1e11 (last case)
ILLCOND = [("ukf_illcond.npz", i) for i in range(5)]


def _case(name, i):
    return load_cases(name)[i]


def _noise(c):
    if c["mode"] == "zero":
        return None, None, None
    return c["noise_pred"], c["noise_upd"], c["noise_rts"]


def mean_err(a, ref):
    return np.max(np.abs(a - ref) / np.maximum(np.abs(ref), 1e-12))


def cov_err(a, ref):
    scale = np.max(np.abs(ref), axis=(-1, -2), keepdims=True)
    return np.max(np.abs(a - ref) / scale)


@pytest.mark.parametrize("name,i", ALL + ILLCOND)
def test_track_restatement_bit_exact(name, i):
    """Same NumPy/SciPy calls in the same order => identical bits to the reference (zero and replayed noise)."""
    c = _case(name, i)
    npred, nupd, nrts = _noise(c)
    H = c["H"]
    m, P = orc.forward_track(c["x0"], c["P0"], H, c["Q"], c["R"], c["dt"], c["dts"], c["z"], c["sog_rate"],
                             c["cog_rate"], npred, nupd)
    assert np.array_equal(m, c["means"])
    assert np.array_equal(P, c["covs"])
    sm, sP = orc.backward_track(m, P, c["Q"], c["dt"], len(c["dts"]), c["sog_rate"], c["cog_rate"], nrts)
    assert np.array_equal(sm, c["means_smoothed"])
    assert np.array_equal(sP, c["covs_smoothed"])


@pytest.mark.parametrize("name,i", ALL + ILLCOND[:4])
def test_batch_restatement_close(name, i):
    """Vectorised form (eigh square root, stacked pinv) vs the reference: well inside the 1e-6 / 1e-5 parity bar."""
    c = _case(name, i)
    npred, nupd, nrts = _noise(c)
    fires, zidx = orc.update_schedule(c["dt"], c["dts"])
    assert np.array_equal(fires, c["fires"])
    rate_idx = zidx - fires
    N = len(c["dt"])
    T = c["z"].shape[1]
    b = lambda a: None if a is None else np.asarray(a)[None]
    m, P = orc.forward_batch(c["x0"][None], c["P0"], c["H"], c["Q"], c["R"], c["dt"][None], fires[None], zidx[None],
                             rate_idx[None], c["z"][None], c["sog_rate"][None], c["cog_rate"][None],
                             noise_pred=b(npred), noise_upd=b(nupd))
    assert mean_err(m[0], c["means"]) < 1e-7
    assert cov_err(P[0], c["covs"]) < 1e-8
    ri = orc.rts_rate_index(N + 1, len(c["dts"]), T)[:N]
    sm, sP = orc.backward_batch(m, P, c["Q"], c["dt"][None], ri[None], c["sog_rate"][None], c["cog_rate"][None],
                                noise_rts=b(nrts))
    # worst case is the real ship (12 h steps): 4e-9, from the square-root algorithm alone
    assert mean_err(sm[0], c["means_smoothed"]) < 1e-7
    assert cov_err(sP[0], c["covs_smoothed"]) < 1e-8


def test_kats():
    import os

    from conftest import GOLDEN

    k = np.load(os.path.join(GOLDEN, "kats.npz"))
    y = np.array([orc.geodetic_dynamics(k["gd_x"][i], k["gd_dt"][i], k["gd_sr"][i], k["gd_cr"][i]) for i in range(64)])
    assert np.array_equal(y, k["gd_y"])
    yv = orc.geodetic_dynamics(k["gd_x"], k["gd_dt"], k["gd_sr"], k["gd_cr"])
    np.testing.assert_allclose(yv, k["gd_y"], rtol=1e-14, atol=1e-13)
    assert np.array_equal(orc.weight_matrix(4), k["weights4"])
    assert np.array_equal(orc.weight_matrix(2), k["weights2"])
    for i in range(16):
        x = k["sp_x"][i].reshape(-1, 1)
        assert np.array_equal(orc._sigma_points_track(x, k["sp_P"][i], 4, 0.0), k["sp_sig_unweighted"][i])
        assert np.array_equal(orc._sigma_points_track(x, k["sp_P"][i], 4, orc.sigma_weights(4)[0]),
                              k["sp_sig_weighted"][i])
        fan = orc._fan(k["sp_x"][i][None], k["sp_P"][i][None], 4)[0].T  # (n, m)
        np.testing.assert_allclose(fan, k["sp_sig_weighted"][i], rtol=0, atol=1e-11)
    W = orc.weight_matrix(4)
    for i in range(16):
        xp, Pp = orc.predict_track(k["pr_x"][i].reshape(-1, 1), k["pr_P"][i], k["Q"], W, k["pr_dt"][i], k["pr_sr"][i],
                                   k["pr_cr"][i])
        assert np.array_equal(xp[:, 0], k["pr_xo"][i]) and np.array_equal(Pp, k["pr_Po"][i])
        xu, Pu = orc.update_track(xp, Pp, k["H"], k["R"], k["up_z"][i])
        assert np.array_equal(xu[:, 0], k["up_xo"][i]) and np.array_equal(Pu, k["up_Po"][i])
        xb, Pb = orc.predict_batch(k["pr_x"][i][None], k["pr_P"][i][None], k["Q"], k["pr_dt"][i][None],
                                   k["pr_sr"][i][None], k["pr_cr"][i][None])
        np.testing.assert_allclose(xb[0], k["pr_xo"][i], rtol=1e-12)
        assert cov_err(Pb[0], k["pr_Po"][i]) < 1e-10


def test_schedule_non_dyadic_misses_updates():
    """SURVEY headline 5: 6.0 h / 10 sub-steps -> float-equality trigger fires once in 20 gaps."""
    c = _case("ukf_synthetic.npz", 7)
    fires, _ = orc.update_schedule(c["dt"], c["dts"])
    assert fires.sum() == 1 and len(fires) == 200


def test_robust_helpers_vs_reference():
    """criterion_index / update_lambda_factor restated in the oracle vs direct calls of the reference's methods."""
    import os

    from conftest import GOLDEN

    k = np.load(os.path.join(GOLDEN, "kats.npz"))
    for i in range(8):
        c = orc.criterion_index(k["rb_x"][i], k["H"], k["rb_z"][i], k["rb_P"][i], k["R"])
        assert np.isclose(c, k["rb_ci"][i], rtol=1e-12)
        lam = orc.update_lambda_factor(k["rb_x"][i], k["H"], 1.0, c, 50.0, k["rb_z"][i], k["rb_P"][i], k["R"])
        assert np.isclose(lam, k["rb_lambda"][i], rtol=1e-12)


def test_check_robustness_vs_reference():
    """oracle.check_robustness against direct calls of the reference's check_robustness (noise zeroed): the rescaled R
    bit for bit and the same number of loop iterations, for outliers needing 0 .. 6 rescalings, diagonal and dense R."""
    import os

    from conftest import GOLDEN

    g = np.load(os.path.join(GOLDEN, "robust.npz"))
    assert g["cr_iters"].max() >= 5 and (g["cr_iters"] == 0).sum() >= 3
    for i in range(len(g["cr_x"])):
        H = g["H"] if i < int(g["cr_dense_from"]) else np.eye(4)
        Rr, it = orc.check_robustness(g["cr_x"][i], H, g["cr_z"][i], g["cr_P"][i], g["cr_R"][i], return_iters=True)
        assert it == int(g["cr_iters"][i])
        assert np.array_equal(Rr, g["cr_Rout"][i]), i


def test_robust_runs_vs_reference():
    """Whole robust tracks (reference run with its check_robustness call site enabled, tests/golden/make_golden.py
    ``_RobustUKF``): the oracle's forward_track(robust=True) + backward_track reproduce them bit for bit."""
    import os

    from conftest import GOLDEN

    g = np.load(os.path.join(GOLDEN, "robust.npz"))
    for ci in range(int(g["nruns"])):
        z, dts = g[f"run{ci}_z"], g[f"run{ci}_dts"]
        m, P = orc.forward_track(z[:, 0], g["P0"], g["H"], g["Q"], g["R"], g[f"run{ci}_dt"], dts, z,
                                 g[f"run{ci}_sog_rate"], g[f"run{ci}_cog_rate"], robust=True)
        assert np.array_equal(m, g[f"run{ci}_means"]) and np.array_equal(P, g[f"run{ci}_covs"])
        assert np.abs(m[:, 0] - g[f"run{ci}_plain_means"][:, 0]).max() > 1.0  # the robust path really differs
        sm, sP = orc.backward_track(m, P, g["Q"], g[f"run{ci}_dt"], len(dts), g[f"run{ci}_sog_rate"],
                                    g[f"run{ci}_cog_rate"])
        assert np.array_equal(sm, g[f"run{ci}_means_smoothed"]) and np.array_equal(sP, g[f"run{ci}_covs_smoothed"])

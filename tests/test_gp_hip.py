"""GP kernel set (csrc/ste_gp.hip) through the C ABI vs the goldens from the reference's GPRegression and the oracle."""
import os

import numpy as np
import pytest
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

NAMES = ["ship", "syn130", "syn300"]


def _data(g, name):
    x = np.insert(np.cumsum(g[f"{name}_dts"]), 0, 0)
    y = np.column_stack([g[f"{name}_lon"], g[f"{name}_lat"]])
    return x, y


def test_kmatrix_and_cholesky_vs_oracle():
    from oracle import gp_oracle as gpo
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    data = [_data(g, n) for n in NAMES]
    batch = GpDeviceBatch([d[0] for d in data], [d[1] for d in data])
    th = g["thetas"][1]
    K = batch.kmatrix(np.tile(th, (3, 1)))
    L, status = batch.cholesky()
    assert not status.any()
    for b, (x, y) in enumerate(data):
        n = len(x)
        Kref, _ = gpo.kernel_matrix(th, x)
        Kref[np.diag_indices_from(Kref)] += gpo.JITTER
        np.testing.assert_allclose(np.tril(K[b, :n, :n]), np.tril(Kref), rtol=1e-13, atol=1e-13)
        Lref = np.linalg.cholesky(Kref)
        np.testing.assert_allclose(L[b, :n, :n], Lref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("order", [1, 2], ids=["row-ordered-inverse", "column-ordered-inverse"])
def test_objective_factors_the_kernel_function_in_place_with_the_stand_alone_bits(order):
    """``ste_gp_lml_f64`` never stores K: the factorisation evaluates the kernel function where ``ste_gp_potrf_f64`` reads the
    matrix ``ste_gp_rbf_kmatrix_f64`` built.  Same expression, same blocked algorithm: the L it leaves is the stand-alone L bit
    for bit; and the alpha that rides along with factorisation and inversion is K^-1 y."""
    from oracle import gp_oracle as gpo
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    data = [_data(g, n) for n in NAMES]
    th = g["thetas"][1]
    batch = GpDeviceBatch([d[0] for d in data], [d[1] for d in data], inverse_order=order)
    thetas = np.tile(th, (3, 1))
    batch.kmatrix(thetas)
    L_alone, status = batch.cholesky()
    assert not status.any()
    _, _, status = batch.objective(thetas)
    assert not status.any()
    L_fused = np.tril(batch.t_K.cpu().numpy())
    alpha = batch.alpha()
    for b, (x, y) in enumerate(data):
        n = len(x)
        assert np.array_equal(L_fused[b, :n, :n], L_alone[b, :n, :n])
        Kref, _ = gpo.kernel_matrix(th, x)
        Kref[np.diag_indices_from(Kref)] += gpo.JITTER
        np.testing.assert_allclose(alpha[b], np.linalg.solve(Kref, y), rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("nout", [1, 3, 4])
@pytest.mark.parametrize("order", [1, 2], ids=["row-ordered-inverse", "column-ordered-inverse"])
def test_objective_with_one_to_four_outputs_vs_oracle(nout, order):
    """The right-hand sides (w = L^-1 y inside the factorisation, alpha = U w inside the inversion) and the quadratic forms
    (inside the K^-1 reduction) run over ``nout`` = 1..4 output columns; the reference only ever passes two (lon, lat)."""
    from oracle import gp_oracle as gpo
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    rng = np.random.default_rng(nout)
    data = []
    for name in NAMES:
        x, y2 = _data(g, name)
        y = np.column_stack([y2, y2[:, ::-1] + 0.1 * rng.standard_normal(y2.shape)])[:, :nout]
        data.append((x, np.ascontiguousarray(y)))
    th = g["thetas"][2]
    batch = GpDeviceBatch([d[0] for d in data], [d[1] for d in data], inverse_order=order)
    lml, grad, status = batch.objective(np.tile(th, (3, 1)))
    assert not status.any()
    alpha = batch.alpha()
    for b, (x, y) in enumerate(data):
        want_lml, want_grad, _, want_alpha = gpo.lml_and_grad(th, x, y)
        assert np.isclose(lml[b], want_lml, rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(grad[b], want_grad, rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(alpha[b], want_alpha, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("order", [1, 2], ids=["row-ordered-inverse", "column-ordered-inverse"])
def test_objective_at_tile_boundaries_vs_oracle(order):
    """n = 2, 16, 63, 64, 65, 128, 129, 192 in one ragged batch: whole tiles, one row over, one strip, almost nothing -- the
    strip, padding and short-pass logic of the panel kernels at every boundary."""
    from oracle import gp_oracle as gpo
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    x_all, y_all = _data(g, "syn300")
    sizes = [2, 16, 63, 64, 65, 128, 129, 192]
    data = [(x_all[:n].copy(), np.ascontiguousarray(y_all[:n])) for n in sizes]
    th = g["thetas"][1]
    batch = GpDeviceBatch([d[0] for d in data], [d[1] for d in data], inverse_order=order)
    lml, grad, status = batch.objective(np.tile(th, (len(sizes), 1)))
    assert not status.any()
    alpha = batch.alpha()
    for b, (x, y) in enumerate(data):
        want_lml, want_grad, _, want_alpha = gpo.lml_and_grad(th, x, y)
        assert np.isclose(lml[b], want_lml, rtol=1e-9, atol=1e-7), sizes[b]
        np.testing.assert_allclose(grad[b], want_grad, rtol=1e-6, atol=1e-5, err_msg=str(sizes[b]))
        np.testing.assert_allclose(alpha[b], want_alpha, rtol=1e-6, atol=1e-8, err_msg=str(sizes[b]))


@pytest.mark.parametrize("batched", [False, True])
def test_lml_and_gradient_vs_reference(batched):
    """Ragged batch (n = 52, 130, 300 -> 1, 3, 5 tiles) at four thetas, vs scikit-learn through the reference wrapper."""
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    data = [_data(g, n) for n in NAMES]
    groups = [[0, 1, 2]] if batched else [[0], [1], [2]]
    for grp in groups:
        batch = GpDeviceBatch([data[i][0] for i in grp], [data[i][1] for i in grp])
        for t, th in enumerate(g["thetas"]):
            lml, grad, status = batch.objective(np.tile(th, (len(grp), 1)))
            assert not status.any()
            for k, i in enumerate(grp):
                assert np.isclose(lml[k], g[f"{NAMES[i]}_lml"][t], rtol=1e-9, atol=1e-7)
                np.testing.assert_allclose(grad[k], g[f"{NAMES[i]}_grad"][t], rtol=1e-6, atol=1e-5)


def test_predict_vs_reference():
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    data = [_data(g, n) for n in NAMES]
    batch = GpDeviceBatch([d[0] for d in data], [d[1] for d in data])
    for t, th in enumerate(g["thetas"][:3]):
        out = batch.predict(np.tile(th, (3, 1)), [g[f"{n}_tq"] for n in NAMES])
        for b, n in enumerate(NAMES):
            np.testing.assert_allclose(out[b][0], g[f"{n}_pred"][t], rtol=1e-8, atol=1e-7)
            np.testing.assert_allclose(out[b][1], g[f"{n}_std"][t], rtol=1e-5, atol=1e-6)


def test_not_positive_definite_is_reported():
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    x = np.array([0.0, 1.0, 1.0, 2.0])  # duplicate input + no noise -> singular K
    batch = GpDeviceBatch([x], [np.zeros((4, 2))], jitter=0.0)
    lml, grad, status = batch.objective(np.log([[1.0, 1.0, 1e-300]]))
    assert status[0] == 1 and lml[0] == -np.inf and not grad.any()


def test_gpregression_api_fit_and_predict():
    """GPRegression.fit / predict (drop-in) on ship 01203823 with a seeded fit, vs the reference's seeded fit."""
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel
    from track_estimators.gaussian_processes.gaussian_process import GPRegression
    from track_estimators.ship_track import ShipTrack

    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    st = ShipTrack()
    st.dts, st.lon, st.lat = g["ship_dts"], g["ship_lon"], g["ship_lat"]
    gp = GPRegression(kernel=1.0 * RBF() + WhiteKernel(noise_level=0.5))
    model = gp.fit(st, gpr_kwargs={"n_restarts_optimizer": 3, "random_state": 0})
    assert np.isclose(model.log_marginal_likelihood_value_, g["ship_fit_lml"], rtol=1e-6)
    np.testing.assert_allclose(model.kernel_.theta, g["ship_fit_theta"], rtol=1e-3, atol=1e-3)
    pred, std = gp.predict(g["ship_tq"])
    assert pred.shape == (len(g["ship_tq"]), 2) and std.shape == pred.shape
    with pytest.raises(NotImplementedError):
        GPRegression(kernel=RBF()).fit(st, gpr_kwargs={"optimizer": None})
    with pytest.raises(AssertionError):
        GPRegression(kernel=RBF()).predict(np.zeros(3))


def test_fit_batch_lockstep_matches_single_fits():
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel
    from track_estimators.gaussian_processes.gaussian_process import GPRegression
    from track_estimators.ship_track import ShipTrack

    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    tracks = []
    for n in NAMES:
        st = ShipTrack()
        st.dts, st.lon, st.lat = g[f"{n}_dts"], g[f"{n}_lon"], g[f"{n}_lat"]
        tracks.append(st)
    gp = GPRegression(kernel=1.0 * RBF() + WhiteKernel(noise_level=0.5))
    thetas, lml = gp.fit_batch(tracks, gpr_kwargs={"n_restarts_optimizer": 0})
    for b, n in enumerate(NAMES):
        single = GPRegression(kernel=1.0 * RBF() + WhiteKernel(noise_level=0.5))
        m = single.fit(tracks[b], gpr_kwargs={"n_restarts_optimizer": 0})
        assert np.isclose(lml[b], m.log_marginal_likelihood_value_, rtol=1e-8)
        # same objective bits in a batch and alone (deterministic reductions) -> same optimiser path
        np.testing.assert_allclose(thetas[b], m.kernel_.theta, rtol=1e-9, atol=1e-9)
    preds = gp.predict_batch([g[f"{n}_tq"] for n in NAMES])
    assert all(p[0].shape == (len(g[f"{n}_tq"]), 2) for p, n in zip(preds, NAMES))


def test_subset_evaluation_matches_full_batch_and_leaves_others_alone():
    """ste_gp_lml_subset_f64: listed tracks get the bits a full launch gives them, the others keep their last outputs."""
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    data = [_data(g, n) for n in NAMES] * 2  # six tracks, ragged
    batch = GpDeviceBatch([d[0] for d in data], [d[1] for d in data])
    th_a, th_b = g["thetas"][0], g["thetas"][2]
    full_a = batch.objective(np.tile(th_a, (6, 1)))
    full_b = batch.objective(np.tile(th_b, (6, 1)))
    batch.objective(np.tile(th_a, (6, 1)))  # device outputs now hold the theta_a results
    theta = np.tile(th_a, (6, 1))
    theta[[1, 4, 5]] = th_b
    lml, grad, status = batch.objective(theta, active=[5, 1, 4])
    assert not status.any()
    for b in range(6):
        want = full_b if b in (1, 4, 5) else full_a
        assert lml[b] == want[0][b]
        np.testing.assert_array_equal(grad[b], want[1][b])
    with pytest.raises(IndexError):
        batch.objective(theta, active=[6])


def test_large_batch_kernels_match_small_batch_kernels_and_oracle():
    """Batches of >= 128 matrices take the column-ordered inverse kernel (gp_trtri_cols); smaller ones the row-ordered
    one.  Same objective either way, and both agree with the oracle."""
    from oracle import gp_oracle as gpo
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    rng = np.random.default_rng(3)
    B = 136
    xs, ys = [], []
    for b in range(B):
        n = int(rng.integers(40, 331))
        x = np.insert(np.cumsum(rng.choice([1.0, 2.0, 6.0], n - 1)), 0, 0)
        f = np.column_stack([np.sin(x / 40.0) + 0.01 * x, np.cos(x / 55.0)])
        xs.append(x)
        ys.append(f + rng.normal(0, 0.05, f.shape))
    theta = np.tile(np.log([2.0, 30.0, 0.01]), (B, 1)) + rng.normal(0, 0.2, (B, 3))
    big = GpDeviceBatch(xs, ys)
    lml, grad, status = big.objective(theta)
    assert not status.any()
    for lo in range(0, B, 50):
        hi = min(B, lo + 50)
        small = GpDeviceBatch(xs[lo:hi], ys[lo:hi])
        l2, g2, s2 = small.objective(theta[lo:hi])
        assert not s2.any()
        np.testing.assert_allclose(lml[lo:hi], l2, rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(grad[lo:hi], g2, rtol=1e-8, atol=1e-7)
    for b in (0, 57, 135):
        l, g, _, _ = gpo.lml_and_grad(theta[b], xs[b], ys[b])
        assert np.isclose(lml[b], l, rtol=1e-10, atol=1e-8)
        np.testing.assert_allclose(grad[b], g, rtol=1e-7, atol=1e-6)


def _gp_tracks(rng, B, n):
    xs, ys = [], []
    for b in range(B):
        x = np.insert(np.cumsum(rng.choice([0.5, 1.0, 2.0], n - 1)), 0, 0)
        f = np.column_stack([np.sin(x / 90.0) + 0.002 * x, np.cos(x / 140.0)])
        xs.append(x)
        ys.append(f + rng.normal(0, 0.05, f.shape))
    return xs, ys


@pytest.mark.timeout(600)
def test_objective_at_n2000_vs_oracle():
    """BASELINE configs[4]'s matrix size (n = 2000: 32 tile columns) in a small batch, every track against the oracle
    (SciPy LAPACK cholesky / cho_solve, the calls scikit-learn makes): LML, gradient and alpha."""
    from oracle import gp_oracle as gpo
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    rng = np.random.default_rng(2000)
    B, n = 5, 2000
    xs, ys = _gp_tracks(rng, B, n)
    xs[3], ys[3] = xs[3][:1937], ys[3][:1937]  # one track that does not fill its last tile
    theta = np.tile(np.log([2.0, 60.0, 0.01]), (B, 1)) + rng.normal(0, 0.15, (B, 3))
    batch = GpDeviceBatch(xs, ys)
    lml, grad, status = batch.objective(theta)
    assert not status.any()
    alphas = batch.alpha()
    for b in range(B):
        l, g, _, a = gpo.lml_and_grad(theta[b], xs[b], ys[b])
        assert np.isclose(lml[b], l, rtol=1e-9, atol=1e-7), (b, lml[b], l)
        np.testing.assert_allclose(grad[b], g, rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(alphas[b], a, rtol=1e-6, atol=1e-6 * np.abs(a).max())


@pytest.mark.timeout(900)
def test_objective_at_config4_full_size_properties():
    """BASELINE configs[4] at full size (1000 tracks x 2000 observations, 68 GB of matrices) through size-independent
    properties: no status flags; a track gives the same bits wherever it sits in the batch (40 distinct tracks, 25
    copies each, shuffled); a subset launch reproduces the full launch bit for bit; two tracks match the oracle."""
    from oracle import gp_oracle as gpo
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    rng = np.random.default_rng(4)
    nuniq, B, n = 40, 1000, 2000
    uxs, uys = _gp_tracks(rng, nuniq, n)
    utheta = np.tile(np.log([2.0, 60.0, 0.01]), (nuniq, 1)) + rng.normal(0, 0.15, (nuniq, 3))
    idx = rng.permutation(np.arange(nuniq).repeat(B // nuniq))
    batch = GpDeviceBatch([uxs[i] for i in idx], [uys[i] for i in idx])
    theta = utheta[idx]
    lml, grad, status = batch.objective(theta)
    assert not status.any() and np.isfinite(lml).all() and np.isfinite(grad).all()
    first = {}
    for slot, src in enumerate(idx):
        first.setdefault(int(src), slot)
    ref_slot = np.array([first[int(src)] for src in idx])
    assert np.array_equal(lml, lml[ref_slot]) and np.array_equal(grad, grad[ref_slot])
    # subset launch: 37 scattered entries, after poisoning the outputs
    sub = np.sort(rng.choice(B, 37, replace=False))
    batch.t_lml.fill_(-7.0)
    batch.t_grad.fill_(-7.0)
    l2, g2, s2 = batch.objective(theta, active=sub.tolist())
    assert np.array_equal(l2[sub], lml[sub]) and np.array_equal(g2[sub], grad[sub])
    others = np.setdiff1d(np.arange(B), sub)
    assert (l2[others] == -7.0).all()
    for u in (0, 23):
        l, g, _, _ = gpo.lml_and_grad(utheta[u], uxs[u], uys[u])
        assert np.isclose(lml[first[u]], l, rtol=1e-9, atol=1e-7)
        np.testing.assert_allclose(grad[first[u]], g, rtol=1e-6, atol=1e-5)


def test_fit_restarts_run_as_batch_entries():
    """fit_thetas with restarts: all (restarts + 1) x B optimisers in one lock-step group give the optimum the one-at-a
    -time loop over restarts finds (same starts, same objective bits), for a shared stream and for per-track streams;
    an integer random_state in fit_batch means one stream per track, like a loop of single seeded fits."""
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel
    from track_estimators.gaussian_processes import gaussian_process as gpm
    from track_estimators.gaussian_processes.device import GpDeviceBatch
    from track_estimators.ship_track import ShipTrack

    g = np.load(os.path.join(GOLDEN, "gp.npz"))
    tracks = []
    for nme in NAMES:
        st = ShipTrack()
        st.dts, st.lon, st.lat = g[f"{nme}_dts"], g[f"{nme}_lon"], g[f"{nme}_lat"]
        tracks.append(st)
    kernel = 1.0 * RBF() + WhiteKernel(noise_level=0.5)
    theta0, bounds = gpm._kernel_spec(kernel)
    data = [gpm.GPRegression._training_data(st) for st in tracks]
    batch = GpDeviceBatch([X[:, 0] for X, _ in data], [y for _, y in data])
    grouped = gpm.fit_thetas(batch, theta0, bounds, 3, np.random.RandomState(5))
    saved = gpm.MAX_LOCKSTEP_ENTRIES
    try:
        gpm.MAX_LOCKSTEP_ENTRIES = 1  # forces one restart per group: the sequential loop
        serial = gpm.fit_thetas(batch, theta0, bounds, 3, np.random.RandomState(5))
    finally:
        gpm.MAX_LOCKSTEP_ENTRIES = saved
    np.testing.assert_allclose(grouped[1], serial[1], rtol=1e-12)
    np.testing.assert_allclose(grouped[0], serial[0], rtol=1e-9, atol=1e-9)
    gp = gpm.GPRegression(kernel=kernel)
    thetas, lml = gp.fit_batch(tracks, gpr_kwargs={"n_restarts_optimizer": 2, "random_state": 0})
    for b in range(len(tracks)):
        m = gpm.GPRegression(kernel=kernel).fit(tracks[b], gpr_kwargs={"n_restarts_optimizer": 2, "random_state": 0})
        assert np.isclose(lml[b], m.log_marginal_likelihood_value_, rtol=1e-10)
        np.testing.assert_allclose(thetas[b], m.kernel_.theta, rtol=1e-8, atol=1e-8)


def test_replicated_batch_sees_the_same_objective_bits():
    """A fit with restarts evaluates a track's first start in the original batch and its restarts in a replicated one whose
    size is a multiple of the original's -- possibly across the library's size threshold for the inverse kernel (row order
    below 128 matrices, column order from there).  The two kernels sum in different orders, so the choice is inherited by
    the replica (ste_gp_batch_f64.inverse_order): same track, same theta -> the same bits."""
    from track_estimators._hip import binding
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    rng = np.random.default_rng(5)
    xs = [np.sort(rng.uniform(0, 200, size=150 + 17 * b)) for b in range(3)]
    ys = [np.stack([np.sin(x / 20) + 0.1 * rng.standard_normal(len(x)), np.cos(x / 30)], axis=1) for x in xs]
    theta = np.log([[1.3, 25.0, 0.05], [0.7, 40.0, 0.1], [2.0, 15.0, 0.02]])
    small = GpDeviceBatch(xs, ys)
    assert small.inverse_order == binding.STE_GP_INVERSE_ROWS
    big = small.replicated(50)  # 150 matrices: on its own the library would pick the column-ordered kernel
    assert big.B == 150 and big.inverse_order == binding.STE_GP_INVERSE_ROWS
    l0, g0, s0 = small.objective(theta)
    l1, g1, s1 = big.objective(np.tile(theta, (50, 1)))
    assert not s0.any() and not s1.any()
    for c in (0, 17, 49):
        assert np.array_equal(l1[3 * c: 3 * c + 3], l0) and np.array_equal(g1[3 * c: 3 * c + 3], g0)
    # and the other order differs by rounding only
    cols = GpDeviceBatch(xs, ys, inverse_order=binding.STE_GP_INVERSE_COLS)
    l2, g2, _ = cols.objective(theta)
    np.testing.assert_allclose(l2, l0, rtol=1e-10)
    np.testing.assert_allclose(g2, g0, rtol=1e-7, atol=1e-8)

"""
Generate the golden vectors under tests/golden/ by importing and RUNNING the reference implementation.

Runs only in the build container (needs /root/reference); never on the GPU box.  The committed ``*.npz`` files hold
data only (inputs + the reference's outputs); nothing of the reference's source travels.

    python tests/golden/make_golden.py

Reference entry points exercised (paths relative to /root/reference/src/track_estimators):
  kalman_filters/non_linear_process.py:6   geodetic_dynamics
  kalman_filters/unscented.py:76,109,144,209,267   compute_sigma_points / compute_weights / predict / update / rts_step
  kalman_filters/unscented.py:389,430,485   robustification helpers (direct calls; their call site is commented out)
  kalman_filters/kalman_filter.py:36,119   run / run_rts_smoother
  ship_track.py:107-338   ShipTrack.read_csv / calculate_* / get_measurements (with utils.haversine_formula / heading)
  utils.py:175            generate_dts

``geographiclib`` is not installed in this image; a stub module is registered before import (the real-data golden
uses the reference's own pure-NumPy ``haversine_formula``/``heading`` injection instead).

Noise handling: the reference draws from the global unseeded ``np.random.normal`` (unscented.py:198,232,320).
``zero`` cases patch it to return zeros; ``replay`` cases patch it to draw from a seeded ``RandomState`` and record
every draw in call order, then re-index the draws per step (the layout the oracle and the HIP kernels consume).
"""
import contextlib
import copy
import io
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "ship-track-estimators_amd"))

# --- import the reference --------------------------------------------------------------------------------------
_m = types.ModuleType("geographiclib")
_g = types.ModuleType("geographiclib.geodesic")


class _Geodesic:  # never called: the goldens inject haversine/heading
    WGS84 = None


_g.Geodesic = _Geodesic
_m.geodesic = _g
sys.modules["geographiclib"] = _m
sys.modules["geographiclib.geodesic"] = _g

REF_SRC = "/root/reference/src"
# the product package has the same import name; make sure the reference wins in this process
sys.path.insert(0, REF_SRC)
import track_estimators as _ref_pkg  # noqa: E402

assert _ref_pkg.__file__.startswith(REF_SRC), _ref_pkg.__file__
from track_estimators.kalman_filters.non_linear_process import geodetic_dynamics  # noqa: E402
from track_estimators.kalman_filters.unscented import UnscentedKalmanFilter  # noqa: E402
from track_estimators.ship_track import ShipTrack  # noqa: E402
from track_estimators.utils import generate_dts, haversine_formula, heading  # noqa: E402

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "ste_synthetic", os.path.join(ROOT, "ship-track-estimators_amd", "track_estimators", "synthetic.py")
)
synthetic = importlib.util.module_from_spec(_spec)
sys.modules["ste_synthetic"] = synthetic
_spec.loader.exec_module(synthetic)

warnings.simplefilter("ignore")


class NoisePatch:
    """Patch np.random.normal: mode 'zero' -> zeros; mode 'replay' -> seeded RandomState, every draw recorded."""

    def __init__(self, mode, seed=0):
        self.mode = mode
        self.rs = np.random.RandomState(seed)
        self.draws = []
        self._orig = None

    def _normal(self, loc=0.0, scale=1.0, size=None):
        if self.mode == "zero":
            out = np.zeros(size)
        else:
            out = self.rs.normal(loc=loc, scale=scale, size=size)
        self.draws.append(np.array(out, dtype=np.float64))
        return out

    def __enter__(self):
        self._orig = np.random.normal
        np.random.normal = self._normal
        return self

    def __exit__(self, *a):
        np.random.normal = self._orig


def ship_track_from_arrays(sb, i):
    st = ShipTrack()
    st.lon, st.lat, st.dts = sb.lon[i].copy(), sb.lat[i].copy(), sb.dts[i].copy()
    st.sog, st.cog = sb.sog[i].copy(), sb.cog[i].copy()
    st.sog_rate, st.cog_rate = sb.sog_rate[i].copy(), sb.cog_rate[i].copy()
    st.z = sb.z[i].copy()
    return st


def run_reference(st, H, Q, R, P, substeps, mode, seed=0):
    """Run ukf.run + run_rts_smoother of the reference on one ShipTrack; return a dict of arrays."""
    x0 = st.z[:, 0].reshape(-1, 1).copy()
    ukf = UnscentedKalmanFilter(H=H, Q=Q, R=R, P=P, x0=x0, non_linear_process=geodetic_dynamics)
    dt = generate_dts(st.dts, substeps)
    N = len(dt)
    with NoisePatch(mode, seed) as npf:
        means, covs = ukf.run(nsteps=N, dt=dt, ship_track=st)
        nfwd = len(npf.draws)
        st2 = copy.deepcopy(st)  # rts_step mutates its ShipTrack (unscented.py:287-292)
        sm, sc = ukf.run_rts_smoother(ship_track=st2)
        draws = npf.draws
    # re-index the draws per step
    cums = np.cumsum(st.dts)
    t = 0
    fires = np.zeros(N, dtype=bool)
    for k in range(N):
        t += dt[k]
        fires[k] = t in cums
    n = H.shape[1]
    noise_pred = np.zeros((N, n))
    noise_upd = np.zeros((N + 1, n))
    noise_rts = np.zeros((N, n))
    it = iter(draws)
    noise_upd[0] = next(it)
    for k in range(N):
        noise_pred[k] = next(it)
        if fires[k]:
            noise_upd[k + 1] = next(it)
    assert nfwd == 1 + N + int(fires.sum())
    for k in range(N - 1, -1, -1):
        noise_rts[k] = next(it)
    assert next(it, None) is None
    return dict(
        dt=dt, fires=fires, means=means, covs=covs, means_smoothed=sm, covs_smoothed=sc,
        noise_pred=noise_pred, noise_upd=noise_upd, noise_rts=noise_rts, x0=x0[:, 0],
    )


def pack_cases(cases):
    out = {"ncases": np.int64(len(cases))}
    for i, c in enumerate(cases):
        for k, v in c.items():
            out[f"c{i}_{k}"] = np.asarray(v)
    return out


def synthetic_cases():
    H, Q, R, P = synthetic.example_matrices()
    H = np.diag([1, 1, 0, 0])  # int64, as the batch example builds it (example_ukf_rts_smoother_batch.py:43)
    cases = []
    # (nobs, gap_h, substeps, seed0, mode) -> N = substeps * (nobs-1)
    plan = [
        (126, 1.0, 4, 0, "zero"), (126, 1.0, 4, 1, "zero"),
        (251, 1.0, 2, 2, "zero"), (501, 1.0, 1, 3, "zero"),
        (126, 1.0, 4, 4, "replay"), (251, 1.0, 2, 5, "replay"), (501, 1.0, 1, 6, "replay"),
        # non-dyadic sub-steps: 6.0 h / 10 -> most float-equality triggers are missed (SURVEY headline 5)
        (21, 6.0, 10, 7, "zero"), (21, 6.0, 10, 8, "replay"),
        # irregular gaps
        (40, 1.0, 3, 9, "zero"),
    ]
    for nobs, gap, s, seed0, mode in plan:
        sb = synthetic.make_batch(1, nobs=nobs, gap_h=gap, seed0=seed0)
        if seed0 == 9:
            rng = np.random.default_rng(99)
            sb.dts[0] = rng.choice([0.5, 1.0, 1.5, 3.0], size=nobs - 1)
        st = ship_track_from_arrays(sb, 0)
        res = run_reference(st, H, Q, R, P, s, mode, seed=1000 + seed0)
        res.update(
            mode=mode, substeps=s, H=H.astype(np.float64), Q=Q, R=R, P0=P,
            dts=sb.dts[0], z=sb.z[0], sog=sb.sog[0], cog=sb.cog[0], sog_rate=sb.sog_rate[0], cog_rate=sb.cog_rate[0],
        )
        cases.append(res)
        print(f"synthetic nobs={nobs} s={s} mode={mode}: N={len(res['dt'])} updates={int(res['fires'].sum())}")
    return cases


def edge_cases():
    """Dense H/R/Q/P, near-pole latitude, heading wrap through 0/360, indefinite prior."""
    cases = []
    rng = np.random.default_rng(4242)
    # (a) dense SPD matrices, H full rank (all four states observed)
    sb = synthetic.make_batch(1, nobs=31, gap_h=2.0, seed0=21)
    A = rng.normal(size=(4, 4))
    P = A @ A.T * 0.05 + np.diag([0.5, 0.5, 0.5, 0.5])
    Bq = rng.normal(size=(4, 4)) * 1e-2
    Q = Bq @ Bq.T + np.diag([1e-4, 1e-4, 1e-6, 1e-6])
    Br = rng.normal(size=(4, 4)) * 0.1
    R = Br @ Br.T + np.diag([0.05, 0.05, 0.5, 2.0])
    H = np.eye(4) + 0.05 * rng.normal(size=(4, 4))
    st = ship_track_from_arrays(sb, 0)
    res = run_reference(st, H, Q, R, P, 2, "replay", seed=77)
    res.update(mode="replay", substeps=2, H=H, Q=Q, R=R, P0=P, dts=sb.dts[0], z=sb.z[0], sog=sb.sog[0],
               cog=sb.cog[0], sog_rate=sb.sog_rate[0], cog_rate=sb.cog_rate[0])
    cases.append(res)
    # (b) high latitude + heading crossing 0/360 repeatedly
    H, Q, R, P = synthetic.example_matrices()
    sb = synthetic.make_batch(1, nobs=41, gap_h=1.0, seed0=22)
    # rebuild the truth at 78N with heading near 359 and a positive turn rate
    T = 41
    lon = np.empty(T); lat = np.empty(T); sog = np.full(T, 25.0); cog = np.empty(T)
    lon[0], lat[0], cog[0] = 10.0, 78.0, 357.0
    for k in range(T - 1):
        lo, la = synthetic._advance(lon[k], lat[k], sog[k], cog[k], 1.0)
        lon[k + 1], lat[k + 1] = lo, la
        cog[k + 1] = (cog[k] + 1.3) % 360.0
    on = np.random.default_rng(5).normal(0, 0.02, (2, T))
    sb.lon[0], sb.lat[0] = lon + on[0], lat + on[1]
    sb.sog[0], sb.cog[0] = sog, cog
    sb.sog_rate[0, 1:] = np.diff(sog); sb.sog_rate[0, 0] = 0
    sb.cog_rate[0, 1:] = np.diff(cog); sb.cog_rate[0, 0] = 0
    sb.z[0] = np.vstack([sb.lon[0], sb.lat[0], sog, cog])
    st = ship_track_from_arrays(sb, 0)
    res = run_reference(st, np.diag([1, 1, 0, 1]), Q, np.diag([0.25, 0.25, 0, 4.0]), P, 4, "zero")
    res.update(mode="zero", substeps=4, H=np.diag([1.0, 1, 0, 1]), Q=Q, R=np.diag([0.25, 0.25, 0, 4.0]), P0=P,
               dts=sb.dts[0], z=sb.z[0], sog=sb.sog[0], cog=sb.cog[0], sog_rate=sb.sog_rate[0], cog_rate=sb.cog_rate[0])
    cases.append(res)
    # (c) indefinite prior covariance: sqrtm goes complex, the reference keeps the real part (unscented.py:97-105)
    sb = synthetic.make_batch(1, nobs=11, gap_h=1.0, seed0=23)
    Pind = np.diag([1.0, 1.0, -0.5, 1.0])
    Pind[0, 1] = Pind[1, 0] = 0.3
    st = ship_track_from_arrays(sb, 0)
    res = run_reference(st, np.diag([1, 1, 0, 0]), Q, R, Pind, 2, "zero")
    res.update(mode="zero", substeps=2, H=H, Q=Q, R=R, P0=Pind, dts=sb.dts[0], z=sb.z[0], sog=sb.sog[0],
               cog=sb.cog[0], sog_rate=sb.sog_rate[0], cog_rate=sb.cog_rate[0])
    cases.append(res)
    for c in cases:
        print(f"edge: N={len(c['dt'])} finite={np.isfinite(c['means_smoothed']).all()}")
    return cases


def real_case():
    """Config 1: ship 01203823, input.json matrices, 2 sub-steps, haversine sog/cog (geographiclib absent)."""
    csv = "/root/reference/data/historical_ships/historical_ship_data.csv"
    st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
    st.read_csv(csv_file=csv, ship_id="01203823", id_col="primary.id", lat_col="lat", lon_col="lon")
    st.get_measurements(include_sog=True, include_cog=True)
    st.calculate_cog_rate()
    st.calculate_sog_rate()
    H = np.diag([1, 1, 0, 0]); R = np.diag([0.001, 0.001, 0, 0])
    Q = np.diag([1e-2, 1e-2, 1e-4, 1e-4]); P = np.diag([1.0, 1.0, 1.0, 1.0])
    cases = []
    for mode in ("zero", "replay"):
        res = run_reference(copy.deepcopy(st), H, Q, R, P, 2, mode, seed=2024)
        res.update(mode=mode, substeps=2, H=H.astype(np.float64), Q=Q, R=R.astype(np.float64), P0=P, dts=st.dts,
                   z=st.z, sog=st.sog, cog=st.cog, sog_rate=st.sog_rate, cog_rate=st.cog_rate, lon=st.lon, lat=st.lat)
        cases.append(res)
        print(f"ship 01203823 mode={mode}: T={len(st.lon)} N={len(res['dt'])}")
    return cases


def gp_cases():
    """GP path: the reference's GPRegression (a wrapper over scikit-learn) on ship 01203823 and on synthetic tracks.
    Per case: log-marginal likelihood + gradient at fixed thetas, predict mean/std at fixed theta (optimizer=None),
    and one seeded full fit (n_restarts_optimizer=3, random_state=0)."""
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel
    from track_estimators.gaussian_processes.gaussian_process import GPRegression

    out = {}
    st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
    st.read_csv(csv_file="/root/reference/data/historical_ships/historical_ship_data.csv", ship_id="01203823",
                id_col="primary.id", lat_col="lat", lon_col="lon")
    tracks = [("ship", st)]
    for name, nobs in (("syn130", 130), ("syn300", 300)):
        sb = synthetic.make_batch(1, nobs=nobs, gap_h=1.0, seed0=500 + nobs)
        t = ShipTrack()
        t.lon, t.lat, t.dts = sb.lon[0], sb.lat[0], sb.dts[0] * np.random.default_rng(nobs).choice([0.5, 1.0, 2.0], nobs - 1)
        tracks.append((name, t))
    thetas = np.log(np.array([[1.0, 1.0, 0.5], [50.0, 20.0, 0.01], [3.0, 150.0, 1e-3], [1e3, 5.0, 0.2]]))
    for name, t in tracks:
        kern = 1.0 * RBF() + WhiteKernel(noise_level=0.5)
        gp = GPRegression(kernel=kern)
        model = gp.fit(t, gpr_kwargs={"optimizer": None})
        lml = []; grad = []
        for th in thetas:
            l, g = model.log_marginal_likelihood(th, eval_gradient=True)
            lml.append(l); grad.append(g)
        times = np.insert(np.cumsum(t.dts), 0, 0)
        tq = np.concatenate([times, (times[:-1] + times[1:]) / 2, [times[-1] + 5.0]])
        preds = []; stds = []
        for th in thetas[:3]:
            kern2 = float(np.exp(th[0])) * RBF(float(np.exp(th[1]))) + WhiteKernel(float(np.exp(th[2])))
            gp2 = GPRegression(kernel=kern2)
            gp2.fit(t, gpr_kwargs={"optimizer": None})
            m, sd = gp2.predict(tq)
            preds.append(m); stds.append(sd)
        gp3 = GPRegression(kernel=1.0 * RBF() + WhiteKernel(noise_level=0.5))
        m3 = gp3.fit(t, gpr_kwargs={"n_restarts_optimizer": 3, "random_state": 0})
        out.update({f"{name}_dts": np.asarray(t.dts), f"{name}_lon": np.asarray(t.lon), f"{name}_lat": np.asarray(t.lat),
                    f"{name}_lml": np.array(lml), f"{name}_grad": np.array(grad), f"{name}_tq": tq,
                    f"{name}_pred": np.array(preds), f"{name}_std": np.array(stds),
                    f"{name}_fit_theta": m3.kernel_.theta, f"{name}_fit_lml": m3.log_marginal_likelihood_value_})
        print(f"gp {name}: n={len(t.lon)} fit theta={m3.kernel_.theta} lml={m3.log_marginal_likelihood_value_:.6f}")
    out["thetas"] = thetas
    return out


def modern_cases():
    """BASELINE configs[3]: data/modern_ships, all 7 ids, 2 sub-steps, haversine sog/cog, zero noise.  Two ships run
    clean (N = 16 794 and 17 084 steps); five contain duplicate timestamps (dt = 0 -> sog = dist/0) and the reference
    dies with ``LinAlgError: SVD did not converge`` inside pinv -- recorded as such.  Histories are sampled (every 50th
    row + the last 20) to keep the fixture small."""
    csv = "/root/reference/data/modern_ships/modern_ship_data.csv"
    H = np.diag([1, 1, 0, 0]); R = np.diag([0.25, 0.25, 0, 0]); Q = np.diag([1e-4, 1e-4, 1e-6, 1e-6]); P = np.eye(4)
    out = {}
    ids = ["AMOUK05", "WDG7520", "WCE5063", "WDA7827", "WGAE", "KAOU", "SJA4RSK"]
    for sid in ids:
        st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
        st.read_csv(csv, ship_id=sid, id_col="id", lat_col="lat", lon_col="lon")
        st.get_measurements(include_sog=True, include_cog=True)
        st.calculate_cog_rate()
        st.calculate_sog_rate()
        out[f"{sid}_T"] = np.int64(len(st.lon))
        out[f"{sid}_zero_dt"] = np.int64((st.dts == 0).sum())
        out[f"{sid}_zsum"] = np.array([np.nansum(np.where(np.isfinite(st.z), st.z, 0.0)), st.dts.sum()])
        try:
            res = run_reference(copy.deepcopy(st), H, Q, R, P, 2, "zero")
            N = len(res["dt"])
            rows = np.unique(np.concatenate([np.arange(0, N + 1, 50), np.arange(N - 19, N + 1)]))
            out[f"{sid}_ok"] = np.int64(1)
            out[f"{sid}_rows"] = rows
            for k in ("means", "covs", "means_smoothed", "covs_smoothed"):
                out[f"{sid}_{k}"] = res[k][rows]
            print(f"modern {sid}: N={N} ok")
        except np.linalg.LinAlgError as e:
            out[f"{sid}_ok"] = np.int64(0)
            print(f"modern {sid}: LinAlgError {e}")
    out["ids"] = np.array(ids)
    return out


def modern_robust_cases():
    """BASELINE configs[3] as written ("Mahalanobis outlier rejection on"): the two modern ships the reference can
    filter at all (modern_cases), run through ``_RobustUKF``; sampled rows as in modern_cases, plus how many updates were
    rescaled at all (criterion above chi_alpha = 50 on entry)."""
    csv = "/root/reference/data/modern_ships/modern_ship_data.csv"
    H = np.diag([1, 1, 0, 0]); R = np.diag([0.25, 0.25, 0, 0]); Q = np.diag([1e-4, 1e-4, 1e-6, 1e-6]); P = np.eye(4)
    out = {}
    ids = ["AMOUK05", "WCE5063"]
    for sid in ids:
        st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
        st.read_csv(csv, ship_id=sid, id_col="id", lat_col="lat", lon_col="lon")
        st.get_measurements(include_sog=True, include_cog=True)
        st.calculate_cog_rate()
        st.calculate_sog_rate()
        x0 = st.z[:, 0].reshape(-1, 1).copy()
        ukf = _RobustUKF(H=H, Q=Q, R=R, P=P, x0=x0, non_linear_process=geodetic_dynamics)
        rescaled = [0]
        orig = ukf.scale_measurement_uncertainty

        def counting(Rm, lam, _o=orig, _r=rescaled):
            _r[0] += 1
            return _o(Rm, lam)

        ukf.scale_measurement_uncertainty = counting
        dt = generate_dts(st.dts, 2)
        N = len(dt)
        with NoisePatch("zero"):
            means, covs = ukf.run(nsteps=N, dt=dt, ship_track=st)
            sm, sc = ukf.run_rts_smoother(ship_track=copy.deepcopy(st))
        rows = np.unique(np.concatenate([np.arange(0, N + 1, 50), np.arange(N - 19, N + 1)]))
        out[f"{sid}_rows"] = rows
        out[f"{sid}_rescalings"] = np.int64(rescaled[0])
        for k, v in (("means", means), ("covs", covs), ("means_smoothed", sm), ("covs_smoothed", sc)):
            out[f"{sid}_{k}"] = v[rows]
        print(f"modern robust {sid}: N={N} rescalings={rescaled[0]}")
    out["ids"] = np.array(ids)
    return out


_DEDUP_FULL = None  # set by modern_dedup_full_cases: the same runs, every row kept


def modern_dedup_full_cases():
    """modern_dedup_cases' reference runs with EVERY history row kept (the sampled fixture compares 1 545 of
    ~75 000 rows): tests/golden/modern_ships_dedup_full.npz.  Also rewrites the sampled fixture from the same runs."""
    global _DEDUP_FULL
    _DEDUP_FULL = {}
    try:
        sampled = modern_dedup_cases()
        full = dict(_DEDUP_FULL)
    finally:
        _DEDUP_FULL = None
    full["ids"] = sampled["ids"]
    np.savez_compressed(os.path.join(HERE, "modern_ships_dedup.npz"), **sampled)
    return full


def modern_dedup_cases():
    """BASELINE configs[3] beyond the two ships the reference can filter as they are: the five ids with duplicate
    timestamps, each read by the REFERENCE's ShipTrack from a file in which the rows that repeat the timestamp of the row
    before were deleted (what this package's opt-in ``drop_duplicate_times`` does on reading), then run with outlier
    rejection on (``_RobustUKF``) and zero noise.  Sampled rows as in modern_cases.  A ship the reference still cannot
    filter is recorded as such.

    Each ship is run a SECOND time with one change inside the reference's own arithmetic -- ``scipy.linalg.sqrtm`` (Schur)
    replaced by the symmetric eigen-decomposition square root, the same matrix function to 1e-15 -- and the fixture keeps, per
    sampled row, how far that moves each history (``*_sens_*``).  On most rows it is < 1e-8; in some episodes of two ships
    (a ship at rest: heading and speed unobservable, the rejection threshold crossed or not on a rounding error) it is
    1e-5 .. 1e-4: there no implementation that does not reproduce LAPACK's rounding can meet 1e-6, and the parity test
    holds those rows to a loose bound only."""
    import tempfile

    import pandas as pd

    csv = "/root/reference/data/modern_ships/modern_ship_data.csv"
    H = np.diag([1, 1, 0, 0]); R = np.diag([0.25, 0.25, 0, 0]); Q = np.diag([1e-4, 1e-4, 1e-6, 1e-6]); P = np.eye(4)
    out = {}
    ids = ["WDG7520", "WDA7827", "WGAE", "KAOU", "SJA4RSK"]
    df = pd.read_csv(csv)
    for sid in ids:
        d = df.loc[df["id"].astype(str) == sid]
        stamp = d["yr"].astype(str) + "-" + d["mo"].astype(str) + "-" + d["dy"].astype(str) + "T" + d["hr"].astype(str)
        d = d.loc[(stamp != stamp.shift(1)).values]
        with tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False) as f:
            d.to_csv(f, index=False)
            tmp = f.name
        try:
            st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
            st.read_csv(tmp, ship_id=sid, id_col="id", lat_col="lat", lon_col="lon")
        finally:
            os.unlink(tmp)
        st.get_measurements(include_sog=True, include_cog=True)
        st.calculate_cog_rate()
        st.calculate_sog_rate()
        out[f"{sid}_T"] = np.int64(len(st.lon))
        out[f"{sid}_dropped"] = np.int64(int((df["id"].astype(str) == sid).sum()) - len(st.lon))
        assert not (st.dts == 0).any()
        x0 = st.z[:, 0].reshape(-1, 1).copy()
        ukf = _RobustUKF(H=H, Q=Q, R=R, P=P, x0=x0, non_linear_process=geodetic_dynamics)
        dt = generate_dts(st.dts, 2)
        N = len(dt)
        try:
            with NoisePatch("zero"), contextlib.redirect_stdout(io.StringIO()):
                means, covs = ukf.run(nsteps=N, dt=dt, ship_track=st)
                sm, sc = ukf.run_rts_smoother(ship_track=copy.deepcopy(st))
        except (np.linalg.LinAlgError, IndexError, ValueError) as e:
            out[f"{sid}_ok"] = np.int64(0)
            print(f"modern dedup {sid}: {type(e).__name__} {e}")
            continue
        finite = bool(np.isfinite(means).all() and np.isfinite(sm).all())
        rows = np.unique(np.concatenate([np.arange(0, N + 1, 50), np.arange(N - 19, N + 1)]))
        # sensitivity run: the same reference code with the eigen square root in place of scipy's Schur sqrtm
        import track_estimators.kalman_filters.unscented as ref_unscented

        def eig_sqrtm(A):
            w, V = np.linalg.eigh(0.5 * (A + A.T))
            return (V * np.sqrt(np.maximum(w, 0.0))) @ V.T

        schur = ref_unscented.scipy.linalg.sqrtm  # the reference calls scipy.linalg.sqrtm by its full name (unscented.py:97)
        ref_unscented.scipy.linalg.sqrtm = eig_sqrtm
        try:
            ukf2 = _RobustUKF(H=H, Q=Q, R=R, P=P, x0=x0.copy(), non_linear_process=geodetic_dynamics)
            with NoisePatch("zero"), contextlib.redirect_stdout(io.StringIO()):
                means2, covs2 = ukf2.run(nsteps=N, dt=dt, ship_track=st)
                sm2, sc2 = ukf2.run_rts_smoother(ship_track=copy.deepcopy(st))
        finally:
            ref_unscented.scipy.linalg.sqrtm = schur
        for k, a, b in (("means", means, means2), ("means_smoothed", sm, sm2)):
            d = np.abs(a - b)
            d[:, 3] = np.abs((a[:, 3] - b[:, 3] + 180.0) % 360.0 - 180.0)
            out[f"{sid}_sens_{k}"] = np.max(d / np.maximum(np.abs(a), 1e-3), axis=1)[rows]
        for k, a, b in (("covs", covs, covs2), ("covs_smoothed", sc, sc2)):
            out[f"{sid}_sens_{k}"] = (np.max(np.abs(a - b), axis=(-1, -2)) / np.max(np.abs(a), axis=(-1, -2)))[rows]
        out[f"{sid}_ok"] = np.int64(1 if finite else 0)
        out[f"{sid}_rows"] = rows
        if _DEDUP_FULL is not None:
            # EVERY row (modern_ships_dedup_full.npz): means in fp64; covariances as fp32 upper triangles -- 6e-8 of an entry,
            # against a tolerance of 1e-5 of the matrix's largest entry -- and the sensitivities as fp32: 11 MB instead of 29
            iu = np.triu_indices(4)
            _DEDUP_FULL[f"{sid}_means"], _DEDUP_FULL[f"{sid}_means_smoothed"] = means, sm
            _DEDUP_FULL[f"{sid}_covs_tri_f32"] = covs[:, iu[0], iu[1]].astype(np.float32)
            _DEDUP_FULL[f"{sid}_covs_smoothed_tri_f32"] = sc[:, iu[0], iu[1]].astype(np.float32)
            for k, a, b in (("means", means, means2), ("means_smoothed", sm, sm2)):
                d = np.abs(a - b)
                d[:, 3] = np.abs((a[:, 3] - b[:, 3] + 180.0) % 360.0 - 180.0)
                _DEDUP_FULL[f"{sid}_sens_{k}"] = np.max(d / np.maximum(np.abs(a), 1e-3), axis=1).astype(np.float32)
            for k, a, b in (("covs", covs, covs2), ("covs_smoothed", sc, sc2)):
                _DEDUP_FULL[f"{sid}_sens_{k}"] = (np.max(np.abs(a - b), axis=(-1, -2)) / np.max(np.abs(a), axis=(-1, -2))).astype(np.float32)
        for k, v in (("means", means), ("covs", covs), ("means_smoothed", sm), ("covs_smoothed", sc)):
            out[f"{sid}_{k}"] = v[rows]
        print(f"modern dedup {sid}: T={len(st.lon)} N={N} finite={finite} sensitive sampled rows (> 1e-8): "
              + ", ".join(f"{k} {int((out[f'{sid}_sens_{k}'] > 1e-8).sum())}/{len(rows)}" for k in ("means", "means_smoothed")))
    out["ids"] = np.array(ids)
    return out


def prep_cases():
    """Observation preparation: the reference's ShipTrack.calculate_cog / calculate_sog /
    calculate_sog_rate / calculate_cog_rate / get_measurements(True, True) (ship_track.py:197-338) with its own pure-NumPy
    sphere pair (utils.haversine_formula / heading) on 40 ragged random tracks (2 .. 90 observations) and on a track
    with duplicate timestamps and coincident points (gap = 0 -> inf / NaN).  Arrays are padded to the longest track."""
    rng = np.random.default_rng(11)
    tracks = []
    for _ in range(40):
        T = int(rng.integers(2, 91))
        lon = rng.uniform(-170, 170) + np.cumsum(rng.normal(0, 0.3, T))
        lat = rng.uniform(-60, 60) + np.cumsum(rng.normal(0, 0.2, T))
        dts = rng.choice([0.5, 1.0, 6.0, 23.0, 24.0, 25.0], T - 1)
        tracks.append((lon, lat, dts))
    lon = np.array([10.0, 10.2, 10.2, 10.5, 10.6, 10.9, 11.0, 11.0])
    lat = np.array([50.0, 50.1, 50.1, 50.3, 50.3, 50.2, 50.2, 50.2])
    tracks.append((lon, lat, np.array([1.0, 0.0, 2.0, 1.0, 0.0, 1.0, 3.0])))  # zero gaps, moved and not moved
    tracks.append((np.array([359.9, 0.1, 0.3, 359.8]), np.array([-0.1, 0.0, 0.1, 0.0]), np.array([1.0, 1.0, 2.0])))  # 0/360 seam
    B, Tm = len(tracks), max(len(t[0]) for t in tracks)
    out = {"nobs": np.array([len(t[0]) for t in tracks], dtype=np.int64)}
    for k in ("lon", "lat", "dts", "sog", "cog", "sog_rate", "cog_rate"):
        out[k] = np.zeros((B, Tm))
    out["z"] = np.zeros((B, 4, Tm))
    for b, (lo, la, d) in enumerate(tracks):
        st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
        st.lon, st.lat, st.dts = lo.copy(), la.copy(), d.copy()
        with np.errstate(all="ignore"):
            st.calculate_cog()
            st.calculate_sog()
            st.calculate_sog_rate()
            st.calculate_cog_rate()
            z = st.get_measurements(include_sog=True, include_cog=True)
        T = len(lo)
        out["lon"][b, :T], out["lat"][b, :T], out["dts"][b, : T - 1] = lo, la, d
        out["sog"][b, :T], out["cog"][b, :T] = st.sog, st.cog
        out["sog_rate"][b, :T], out["cog_rate"][b, :T] = st.sog_rate, st.cog_rate
        out["z"][b, :, :T] = z
    print("prep: %d tracks, non-finite sog entries: %d" % (B, int((~np.isfinite(out["sog"])).sum())))
    return out


EXAMPLE_SEED_BASE = 20240000


def example_cases():
    """The compute part of the reference's two batch examples run as written (examples/example_ukf_rts_smoother_batch.py
    :15-90, examples/example_gaussian_process_batch.py:15-55) on data/historical_ships, with two harness changes only:
    the sphere pair (haversine_formula / heading) is injected because geographiclib is absent here, and the global
    generator is seeded per ship (EXAMPLE_SEED_BASE + position in ``ids``) so that the injected noise can be replayed.
    Recorded: the id list after ``ids.pop(1)``, what happened to each id (ok / skipped because dt > 48 / error), and for
    every ok ship rows 0, N//2 and N of the filtered and smoothed histories.  GP: the fitted theta, the log marginal
    likelihood and the predictions of the first 8 ships with n_restarts_optimizer = 2 and random_state = 0."""
    import pandas as pd
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel
    from track_estimators.gaussian_processes.gaussian_process import GPRegression

    csv = "/root/reference/data/historical_ships/historical_ship_data.csv"
    df = pd.read_csv(csv)
    ids = df["primary.id"].unique().tolist()
    ids.pop(1)
    out = {"ids": np.array([str(i) for i in ids]), "seed_base": np.int64(EXAMPLE_SEED_BASE)}
    cat = []
    H = np.diag([1, 1, 0, 0]); R = np.diag([0.25, 0.25, 0, 0]); Q = np.diag([1e-4, 1e-4, 1e-6, 1e-6])
    P = np.diag([1.0, 1.0, 1.0, 1.0])
    for i, sid in enumerate(ids):
        st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
        st.read_csv(csv_file=csv, ship_id=sid, id_col="primary.id", lat_col="lat", lon_col="lon2", reverse=False)
        z = st.get_measurements(include_sog=True, include_cog=True)
        st.calculate_cog_rate()
        st.calculate_sog_rate()
        x0 = z[:, 0].reshape(-1, 1).copy()
        ukf = UnscentedKalmanFilter(H=H, Q=Q, R=R, P=P, x0=x0, non_linear_process=geodetic_dynamics)
        dt_array = generate_dts(st.dts, 2)
        if dt_array.max() > 48:
            cat.append("skipped")
            continue
        np.random.seed(EXAMPLE_SEED_BASE + i)
        try:
            m, c = ukf.run(nsteps=len(dt_array), dt=dt_array, ship_track=st)
            sm, sc = ukf.run_rts_smoother(ship_track=st)
        except Exception as exc:
            cat.append("error")
            print("example: error in", sid, type(exc).__name__)
            continue
        cat.append("ok")
        N = len(dt_array)
        rows = np.array([0, N // 2, N])
        out[f"ukf_{i}_rows"] = rows
        out[f"ukf_{i}_means"], out[f"ukf_{i}_covs"] = m[rows], c[rows]
        out[f"ukf_{i}_means_smoothed"], out[f"ukf_{i}_covs_smoothed"] = sm[rows], sc[rows]
    out["category"] = np.array(cat)
    print("example ukf: %d ids, ok %d, skipped %d, error %d" % (len(ids), cat.count("ok"), cat.count("skipped"), cat.count("error")))
    for i, sid in enumerate(ids[:8]):
        st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
        st.read_csv(csv_file=csv, ship_id=sid, id_col="primary.id", lat_col="lat", lon_col="lon2", reverse=False)
        gpr = GPRegression(kernel=1.0 * RBF() + WhiteKernel(noise_level=0.5))
        model = gpr.fit(st, gpr_kwargs={"n_restarts_optimizer": 2, "random_state": 0})
        times = np.insert(np.cumsum(generate_dts(st.dts, substeps=1)), 0, 0)
        pred, std = gpr.predict(times=times)
        out[f"gp_{i}_theta"] = model.kernel_.theta.copy()
        out[f"gp_{i}_lml"] = np.float64(model.log_marginal_likelihood_value_)
        out[f"gp_{i}_pred"], out[f"gp_{i}_std"] = pred, std
    out["gp_count"] = np.int64(8)
    return out


def smooth_case():
    """SOG / COG pre-smoothing as the reference CLI does it (cli/main_cli.py:99-109 with utils.smooth, utils.py:150-172):
    calculate_cog / calculate_sog, moving average of width 5, then get_measurements(True, True) and the rates -- on ship
    01203823 of the reference's CLI example, sphere pair injected."""
    from track_estimators.utils import smooth

    st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
    st.read_csv("/root/reference/data/historical_ships/historical_ship_data.csv", ship_id="01203823", id_col="primary.id",
                lat_col="lat", lon_col="lon")
    st.calculate_cog()
    st.calculate_sog()
    raw_sog, raw_cog = st.sog.copy(), st.cog.copy()
    st.sog = smooth(st.sog, 5)
    st.cog = smooth(st.cog, 5)
    z = st.get_measurements(include_sog=True, include_cog=True)
    st.calculate_cog_rate()
    st.calculate_sog_rate()
    return dict(box=np.int64(5), raw_sog=raw_sog, raw_cog=raw_cog, sog=st.sog, cog=st.cog, z=z, sog_rate=st.sog_rate,
                cog_rate=st.cog_rate, smooth_even=smooth(np.arange(7.0) ** 2, 4))


def savgol_case():
    """The compute part of examples/example_ukf_rts_smoother_savgol.py:15-86 as written: ship 01205070 read in reverse,
    COG / SOG through scipy's Savitzky-Golay filter (windows 20 / 4, orders 4 / 2), the example's matrices, two
    sub-steps, UKF then RTS smoother.  Harness changes only: the sphere pair is injected (geographiclib is absent here)
    and the global noise draws are zeroed."""
    from scipy.signal import savgol_filter

    st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
    st.read_csv(csv_file="/root/reference/data/historical_ships/historical_ship_data.csv", ship_id="01205070", id_col="id",
                lat_col="lat", lon_col="lon", reverse=True)
    st.calculate_cog()
    st.calculate_sog()
    raw_sog, raw_cog = st.sog.copy(), st.cog.copy()
    st.sog = savgol_filter(st.sog, 20, 4)
    st.cog = savgol_filter(st.cog, 4, 2)
    z = st.get_measurements(include_sog=True, include_cog=True)
    st.calculate_cog_rate()
    st.calculate_sog_rate()
    H = np.diag([1, 1, 0, 0])
    R = np.diag([0.001, 0.001, 0, 0])
    Q = np.diag([1e-3, 1e-3, 1e-6, 1e-6])
    P = np.diag([1.0, 1.0, 1.0, 1.0])
    res = run_reference(st, H, Q, R, P, 2, "zero")
    res.update(raw_sog=raw_sog, raw_cog=raw_cog, sog=st.sog, cog=st.cog, sog_rate=st.sog_rate, cog_rate=st.cog_rate, z=z,
               dts=st.dts, lon=st.lon, lat=st.lat, H=H.astype(np.float64), Q=Q, R=R, P0=P)
    print(f"savgol example: T={len(st.lon)} N={len(res['dt'])}")
    return res


def illcond_cases():
    """Smoother gains on ill-conditioned P_b: tiny process noise, a prior that knows speed and heading
    far better than position, long steps.  cond(P_b) of the reference's own smoother pass is recorded per case (its P_b is
    recomputed here from the reference's filtered history with the reference's own calls) and spans ~1e4 ... 1e11, i.e.
    both sides of the factorisation's pivot threshold (kLdlPivotTol)."""
    H = np.diag([1, 1, 0, 0])
    plan = [
        # (P0 diagonal, Q diagonal, R position variance, gap hours, substeps, seed)
        ([1.0, 1.0, 1e-3, 1e-2], [1e-6, 1e-6, 1e-9, 1e-8], 1e-2, 3.0, 2, 21),
        ([1.0, 1.0, 1e-5, 1e-4], [1e-8, 1e-8, 1e-11, 1e-10], 1e-2, 6.0, 2, 22),
        ([1.0, 1.0, 1e-7, 1e-6], [1e-9, 1e-9, 1e-13, 1e-12], 1e-3, 6.0, 1, 23),
        ([10.0, 10.0, 1e-8, 1e-8], [1e-10, 1e-10, 1e-14, 1e-14], 1e-3, 12.0, 1, 24),
        ([100.0, 100.0, 1e-9, 1e-9], [0.0, 0.0, 1e-15, 1e-15], 1e-2, 12.0, 1, 25),
    ]
    cases = []
    for p0, q, r, gap, s, seed in plan:
        P, Q, R = np.diag(p0), np.diag(q), np.diag([r, r, 0, 0])
        sb = synthetic.make_batch(1, nobs=41, gap_h=gap, seed0=seed)
        st = ship_track_from_arrays(sb, 0)
        res = run_reference(st, H, Q, R, P, s, "zero")
        # cond(P_b) along the reference's smoother pass: its own fan / process model / weights on its own filtered history
        ukf = UnscentedKalmanFilter(H=H, Q=Q, R=R, P=P, x0=res["x0"].reshape(-1, 1), non_linear_process=geodetic_dynamics)
        ukf.compute_weights()
        rep = int((len(res["dt"]) + 1) / len(sb.dts[0]))
        sr, cr = np.repeat(sb.sog_rate[0], rep), np.repeat(sb.cog_rate[0], rep)
        conds = []
        for k in range(len(res["dt"])):
            xk, Pk = res["means"][k].reshape(-1, 1), res["covs"][k]
            sig = ukf.compute_sigma_points(xk, Pk).copy()
            for j in range(sig.shape[1]):
                sig[:, j] = geodetic_dynamics(sig[:, j], c=None, dt=res["dt"][k], sog_rate=sr[k], cog_rate=cr[k])
            S = sig - xk
            Pb = S @ ukf.weights @ S.T + Q
            conds.append(np.linalg.cond(Pb))
        res.update(mode="zero", substeps=s, H=H.astype(np.float64), Q=Q, R=R, P0=P, dts=sb.dts[0], z=sb.z[0], sog=sb.sog[0],
                   cog=sb.cog[0], sog_rate=sb.sog_rate[0], cog_rate=sb.cog_rate[0], cond_pb=np.asarray(conds))
        cases.append(res)
        print(f"illcond seed={seed}: N={len(res['dt'])} cond(P_b) {min(conds):.2e} .. {max(conds):.2e}")
    return cases


def two_runs_case():
    """A second ``run`` on the same filter object appends to the history and keeps the running time
    (kalman_filter.py:22-31,98): the update index restarts at 0 while ``self.time`` continues, so the float-equality
    trigger of the second run is evaluated against times offset by the first run's span."""
    H, Q, R, P = synthetic.example_matrices()
    sb = synthetic.make_batch(1, nobs=9, gap_h=1.0, seed0=77)
    st = ship_track_from_arrays(sb, 0)
    x0 = st.z[:, 0].reshape(-1, 1).copy()
    ukf = UnscentedKalmanFilter(H=np.diag([1, 1, 0, 0]), Q=Q, R=R, P=P, x0=x0, non_linear_process=geodetic_dynamics)
    dt1 = generate_dts(st.dts[:4], 2)
    dt2 = generate_dts(st.dts[4:], 2)
    with NoisePatch("zero"):
        m1, c1 = ukf.run(nsteps=len(dt1), dt=dt1, ship_track=st)
        m2, c2 = ukf.run(nsteps=len(dt2), dt=dt2, ship_track=st)
    return dict(dts=sb.dts[0], z=sb.z[0], sog=sb.sog[0], cog=sb.cog[0], sog_rate=sb.sog_rate[0], cog_rate=sb.cog_rate[0],
                dt1=dt1, dt2=dt2, means1=m1, covs1=c1, means2=m2, covs2=c2, time_end=np.float64(ukf.time), Q=Q, R=R, P0=P)


def csv_fixture():
    """Rows of ship 01203823 cut from the reference's data file (the input of its own CLI example), plus one of the
    header rows the source file repeats between ships, so the id column stays a string column as in the full file."""
    import pandas as pd

    df = pd.read_csv("/root/reference/data/historical_ships/historical_ship_data.csv", dtype=str, keep_default_na=False)
    sub = df[df["primary.id"] == "01203823"]
    hdr = pd.DataFrame([list(df.columns)], columns=df.columns)
    pd.concat([sub, hdr], ignore_index=True).to_csv(os.path.join(HERE, "ship_01203823.csv"), index=False)


class _RobustUKF(UnscentedKalmanFilter):
    """The reference filter with its commented-out robustification call site (unscented.py:228) enabled the way the
    method's signature says it is meant to be used: ``check_robustness`` (unscented.py:353-387) returns the rescaled
    measurement covariance and the update that follows runs with it.  Everything executed is the reference's code; this
    harness only routes the returned R into that one update."""

    def update(self, z):
        R0 = self.R
        with contextlib.redirect_stdout(io.StringIO()):  # check_robustness prints every iteration
            self.R = self.check_robustness(np.asarray(z, dtype=np.float64).reshape(-1, 1), self.P, self.R)
        try:
            super().update(z)
        finally:
            self.R = R0


def robust_cases():
    """Mahalanobis robustification, noise draws zeroed.

    ``cr_*``: direct calls of the reference's ``check_robustness`` on (x, P, z, R) triples whose outlier size spans no
    rescaling, one rescaling and several; the returned R and the number of loop iterations are recorded.
    ``run*_*``: whole tracks through ``_RobustUKF`` (run + run_rts_smoother) with gross outliers injected into some
    observations, 1 and 2 sub-steps."""
    rng = np.random.default_rng(2024)
    H, Q, R, P0 = synthetic.example_matrices()
    out = dict(H=H, Q=Q, R=R, P0=P0)
    xs, Ps, zs, Rs, Ro, its = [], [], [], [], [], []
    scales = [0.1, 0.5, 2.0, 4.0, 6.0, 9.0, 15.0, 30.0, 60.0, 120.0, 3.0, 8.0]
    for i, sc in enumerate(scales):
        A = rng.normal(size=(4, 4)) * 0.2
        P = A @ A.T + np.diag([0.02, 0.02, 0.05, 0.05])
        x = np.array([rng.uniform(-60, 60), rng.uniform(-60, 60), rng.uniform(5, 40), rng.uniform(0, 360)])
        z = x + rng.normal(0, 1.0, 4) * sc
        Rin = R if i < 10 else np.diag([0.25, 0.5, 0.1, 0.1]) + 0.01  # two cases with a dense R
        Hin = H if i < 10 else np.eye(4)
        u = UnscentedKalmanFilter(H=Hin, Q=Q, R=Rin, P=P, x0=x)
        buf = io.StringIO()
        with NoisePatch("zero"), contextlib.redirect_stdout(buf):
            Rout = u.check_robustness(z.reshape(-1, 1), P, Rin)
        xs.append(x); Ps.append(P); zs.append(z); Rs.append(Rin); Ro.append(np.asarray(Rout)); its.append(len(buf.getvalue().splitlines()) - 1)
    print("robust check_robustness iterations:", its)
    out.update(cr_x=np.array(xs), cr_P=np.array(Ps), cr_z=np.array(zs), cr_R=np.array(Rs), cr_Rout=np.array(Ro),
               cr_iters=np.array(its), cr_dense_from=np.int64(10))
    for ci, (nobs, sub, seed, bumps) in enumerate([(12, 1, 501, {4: (9.0, -6.0)}), (20, 2, 502, {3: (-14.0, 5.0), 11: (25.0, 0.0)}),
                                                   (16, 2, 503, {0: (7.0, 7.0), 15: (-8.0, 3.0)})]):
        sb = synthetic.make_batch(1, nobs=nobs, gap_h=1.0, seed0=seed)
        for col, (dlon, dlat) in bumps.items():
            sb.z[0, 0, col] += dlon
            sb.z[0, 1, col] += dlat
        st = ship_track_from_arrays(sb, 0)
        x0 = st.z[:, 0].reshape(-1, 1).copy()
        ukf = _RobustUKF(H=H, Q=Q, R=R, P=P0, x0=x0, non_linear_process=geodetic_dynamics)
        dt = generate_dts(st.dts, sub)
        with NoisePatch("zero"):
            means, covs = ukf.run(nsteps=len(dt), dt=dt, ship_track=st)
            sm, sc_ = ukf.run_rts_smoother(ship_track=copy.deepcopy(st))
        assert np.array_equal(ukf.R, R)
        plain = run_reference(ship_track_from_arrays(sb, 0), H, Q, R, P0, sub, "zero")
        print(f"robust run{ci}: N={len(dt)} max |robust - plain| lon = {np.abs(means[:, 0] - plain['means'][:, 0]).max():.3f}")
        out.update({f"run{ci}_z": sb.z[0], f"run{ci}_dts": sb.dts[0], f"run{ci}_sog_rate": sb.sog_rate[0],
                    f"run{ci}_cog_rate": sb.cog_rate[0], f"run{ci}_substeps": np.int64(sub), f"run{ci}_dt": dt,
                    f"run{ci}_means": means, f"run{ci}_covs": covs, f"run{ci}_means_smoothed": sm,
                    f"run{ci}_covs_smoothed": sc_, f"run{ci}_plain_means": plain["means"]})
    out["nruns"] = np.int64(3)
    return out


def kats():
    """Per-function known answers."""
    rng = np.random.default_rng(7)
    out = {}
    # geodetic_dynamics
    X = np.column_stack([rng.uniform(-180, 180, 64), rng.uniform(-85, 85, 64), rng.uniform(0, 60, 64),
                         rng.uniform(-30, 400, 64)])
    dts = rng.choice([0.25, 0.5, 1.0, 6.0, 12.5, -1.0], 64)
    sr = rng.normal(0, 0.05, 64); cr = rng.normal(0, 0.5, 64)
    Y = np.array([geodetic_dynamics(X[i], None, dts[i], sr[i], cr[i]) for i in range(64)])
    out.update(gd_x=X, gd_dt=dts, gd_sr=sr, gd_cr=cr, gd_y=Y)
    # sigma points (weights set, and weights never computed -> W[0,0]=0 branch), weights
    Ps, xs, sig_w, sig_now = [], [], [], []
    for i in range(16):
        A = rng.normal(size=(4, 4)); P = A @ A.T * rng.choice([1e-4, 1e-2, 1.0]) + np.diag([1e-5, 1e-5, 1e-3, 1e-3])
        x = rng.normal(size=4) * 10
        u = UnscentedKalmanFilter(H=np.eye(4), P=P, x0=x)
        sig_now.append(u.compute_sigma_points().copy())
        u.compute_weights()
        sig_w.append(u.compute_sigma_points().copy())
        Ps.append(P); xs.append(x)
    u = UnscentedKalmanFilter(H=np.eye(4))
    out.update(sp_P=np.array(Ps), sp_x=np.array(xs), sp_sig_weighted=np.array(sig_w), sp_sig_unweighted=np.array(sig_now),
               weights4=u.compute_weights().copy())
    u2 = UnscentedKalmanFilter(H=np.eye(2))
    out.update(weights2=u2.compute_weights().copy())
    # single predict / update calls
    H, Q, R, P0 = synthetic.example_matrices()
    px, pP, pdt, psr, pcr, pxo, pPo, uz, uxo, uPo = ([] for _ in range(10))
    with NoisePatch("zero"):
        for i in range(16):
            A = rng.normal(size=(4, 4)) * 0.1; P = A @ A.T + np.diag([1e-3, 1e-3, 1e-2, 1e-2])
            x = np.array([rng.uniform(-60, 60), rng.uniform(-60, 60), rng.uniform(5, 40), rng.uniform(0, 360)])
            u = UnscentedKalmanFilter(H=H, Q=Q, R=R, P=P, x0=x, non_linear_process=geodetic_dynamics)
            d, a, b = rng.choice([0.25, 1.0, 12.0]), rng.normal(0, 0.05), rng.normal(0, 0.5)
            u.predict(dt=d, c=None, sog_rate=a, cog_rate=b)
            px.append(x); pP.append(P); pdt.append(d); psr.append(a); pcr.append(b)
            pxo.append(u.x[:, 0].copy()); pPo.append(u.P.copy())
            zz = u.x[:, 0] + rng.normal(0, 0.3, 4); zz[3] += rng.choice([0, 360, -360, 180])
            u.update(zz.copy())
            uz.append(zz); uxo.append(u.x[:, 0].copy()); uPo.append(u.P.copy())
    out.update(pr_x=np.array(px), pr_P=np.array(pP), pr_dt=np.array(pdt), pr_sr=np.array(psr), pr_cr=np.array(pcr),
               pr_xo=np.array(pxo), pr_Po=np.array(pPo), up_z=np.array(uz), up_xo=np.array(uxo), up_Po=np.array(uPo),
               H=H, Q=Q, R=R)
    # robustification helpers, direct calls (unscented.py:389-511)
    ci, lf, zs, Ps2, xs2 = [], [], [], [], []
    for i in range(8):
        A = rng.normal(size=(4, 4)) * 0.3; P = A @ A.T + np.eye(4) * 0.05
        x = rng.normal(size=4)
        u = UnscentedKalmanFilter(H=H, Q=Q, R=R, P=P, x0=x)
        zz = x + rng.normal(0, 3.0, 4)
        c = u.criterion_index(zz.reshape(-1, 1), P, R)
        l = u.update_lambda_factor(1.0, c, 50.0, zz.reshape(-1, 1), P, R)
        ci.append(c); lf.append(l); zs.append(zz); Ps2.append(P); xs2.append(x)
    out.update(rb_x=np.array(xs2), rb_P=np.array(Ps2), rb_z=np.array(zs), rb_ci=np.array(ci), rb_lambda=np.array(lf))
    # generate_dts
    out.update(gdts_in=np.array([10, 5, 3, 2, 1, 10, 10.0]), gdts_2=generate_dts([10, 5, 3, 2, 1, 10, 10], 2),
               gdts_4=generate_dts([10, 5, 3, 2, 1, 10, 10], 4), gdts_3=generate_dts([23.0, 24.0, 25.0], 3))
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "ukf_synthetic.npz"), **pack_cases(synthetic_cases()))
    np.savez_compressed(os.path.join(HERE, "ukf_edge.npz"), **pack_cases(edge_cases()))
    np.savez_compressed(os.path.join(HERE, "ukf_ship_01203823.npz"), **pack_cases(real_case()))
    np.savez_compressed(os.path.join(HERE, "kats.npz"), **kats())
    csv_fixture()
    np.savez_compressed(os.path.join(HERE, "gp.npz"), **gp_cases())
    np.savez_compressed(os.path.join(HERE, "modern_ships.npz"), **modern_cases())
    np.savez_compressed(os.path.join(HERE, "two_runs.npz"), **two_runs_case())
    np.savez_compressed(os.path.join(HERE, "robust.npz"), **robust_cases())
    np.savez_compressed(os.path.join(HERE, "modern_ships_robust.npz"), **modern_robust_cases())
    np.savez_compressed(os.path.join(HERE, "modern_ships_dedup_full.npz"), **modern_dedup_full_cases())  # + the sampled file
    np.savez_compressed(os.path.join(HERE, "track_prep.npz"), **prep_cases())
    np.savez_compressed(os.path.join(HERE, "batch_examples.npz"), **example_cases())
    np.savez_compressed(os.path.join(HERE, "cli_smooth.npz"), **smooth_case())
    np.savez_compressed(os.path.join(HERE, "savgol_example.npz"), **savgol_case())
    np.savez_compressed(os.path.join(HERE, "ukf_illcond.npz"), **pack_cases(illcond_cases()))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


SELECTABLE = {"modern_dedup": (modern_dedup_cases, "modern_ships_dedup.npz"), "modern_dedup_full": (modern_dedup_full_cases, "modern_ships_dedup_full.npz"), "robust": (robust_cases, "robust.npz"), "modern_robust": (modern_robust_cases, "modern_ships_robust.npz"),
              "prep": (prep_cases, "track_prep.npz"), "examples": (example_cases, "batch_examples.npz"),
              "smooth": (smooth_case, "cli_smooth.npz"), "savgol": (savgol_case, "savgol_example.npz"),
              "illcond": (lambda: pack_cases(illcond_cases()), "ukf_illcond.npz")}

if __name__ == "__main__":
    if len(sys.argv) > 1:  # regenerate selected fixtures only: python make_golden.py robust prep ...
        for name in sys.argv[1:]:
            fn, fname = SELECTABLE[name]
            np.savez_compressed(os.path.join(HERE, fname), **fn())
    else:
        main()

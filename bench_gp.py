#!/usr/bin/env python3
"""
bench_gp.py — BASELINE.json configs[4]: GP-regression objective (RBF K-matrix build + Cholesky + gradient) for a batch
of tracks on one MI355X, beside scikit-learn's algorithm on the host.  Not the headline metric (that is bench.py); one
JSON line with the same roofline / cpu_baseline objects.

  python bench_gp.py --tracks 1000 --nobs 2000 --evals 3
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ship-track-estimators_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def host_core_share() -> int:
    """Cores this process may really use (what BLAS threads can run on): the affinity mask, cut by the cgroup CPU quota when
    there is one, and by 16 -- the CPU share of a one-GPU box -- when the quota is not visible.  os.cpu_count() reports the
    machine (256 on the GPU box), not the share."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (= vector peak on CDNA4)


def wiggle_tracks(B, n, seed=0):
    """Tracks whose GP optimum lies inside the theta bounds: a bounded oscillation (several periods over the window)
    plus a small drift and 0.05 of noise, hourly-ish gaps.  The great-circle tracks of synthetic.make_batch are almost
    straight lines over a window, which sends constant and length scale to their upper bound (round 1's fit figure)."""
    rng = np.random.default_rng(seed)
    xs, ys = [], []
    for b in range(B):
        x = np.insert(np.cumsum(rng.choice([0.5, 1.0, 2.0], n - 1)), 0, 0)
        p1, p2 = rng.uniform(60, 140), rng.uniform(90, 200)
        f = np.column_stack([np.sin(x / p1) + 0.0005 * x, np.cos(x / p2)])
        xs.append(x)
        ys.append(f + rng.normal(0, 0.05, f.shape))
    return xs, ys


def measure_fit(tracks=64, nobs=2000, restarts=15, cpu=True, lockstep_only=False):
    """A whole hyper-parameter fit the way the reference asks for it (L-BFGS-B from the example's kernel plus seeded
    restarts, best optimum kept) on data with an interior optimum: all (restarts + 1) x tracks optimisers in lock-step
    as one batch, beside the same fit with the restarts one after the other and scikit-learn's fit of one track."""
    import torch
    from track_estimators.gaussian_processes import gaussian_process as gpm
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    xs, ys = wiggle_tracks(tracks, nobs)
    batch = GpDeviceBatch(xs, ys)
    theta0 = np.log([1.0, 1.0, 0.5])  # 1.0 * RBF(1.0) + WhiteKernel(0.5), the reference example's kernel
    bounds = np.log(np.tile([1e-5, 1e5], (3, 1)))
    out = {"tracks": tracks, "nobs": nobs, "restarts": restarts}
    for label, cap in (("lockstep_all_restarts", gpm.MAX_LOCKSTEP_ENTRIES), ("restarts_one_after_the_other", 1))[:1 if lockstep_only else 2]:
        saved = gpm.MAX_LOCKSTEP_ENTRIES
        gpm.MAX_LOCKSTEP_ENTRIES = cap
        try:
            rngs = [np.random.RandomState(0) for _ in range(tracks)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th, best = gpm.fit_thetas(batch, theta0, bounds, restarts, rngs)
            torch.cuda.synchronize()
            out[label] = {"seconds": time.perf_counter() - t0, "tracks_per_s": tracks / (time.perf_counter() - t0)}
        finally:
            gpm.MAX_LOCKSTEP_ENTRIES = saved
        out[label]["lml_mean"] = float(np.mean(best))
    med = np.exp(np.median(th, axis=0))
    out["theta_median"] = med.tolist()
    out["interior_optimum"] = bool(np.all((med > 1e-4) & (med < 1e4)))
    if cpu:
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process.kernels import RBF, WhiteKernel

        t0 = time.perf_counter()
        ref = GaussianProcessRegressor(kernel=1.0 * RBF(1.0) + WhiteKernel(0.5), n_restarts_optimizer=restarts,
                                       random_state=0).fit(xs[0].reshape(-1, 1), ys[0])
        out["cpu_reference"] = {
            "seconds_per_track": time.perf_counter() - t0, "cores": host_core_share(), "kind": "reference",
            "sample": "scikit-learn GaussianProcessRegressor(n_restarts_optimizer, random_state=0).fit on track 0, the call "
                      "the reference's GPRegression.fit makes (gaussian_process.py:63-66)",
            "lml_rel_diff_track0": float(abs(best[0] - ref.log_marginal_likelihood_value_)
                                         / abs(ref.log_marginal_likelihood_value_))}
    return out


def measure(tracks=1000, nobs=2000, evals=3, cpu_evals=2, fit=False):
    """One batched objective evaluation timed with HIP events on the launch stream; returns the JSON object.
    bench.py calls this for its ``extra.gp_config4`` entry so that the driver's default run carries the GP line too."""
    args = argparse.Namespace(tracks=tracks, nobs=nobs, evals=evals, cpu_evals=cpu_evals, fit=fit)

    import torch
    from track_estimators import synthetic
    from track_estimators.gaussian_processes.device import GpDeviceBatch

    B, n = args.tracks, args.nobs
    sb = synthetic.make_batch(B, nobs=n, gap_h=1.0, seed0=0)
    xs = [np.insert(np.cumsum(sb.dts[b]), 0, 0) for b in range(B)]
    ys = [np.column_stack([sb.lon[b], sb.lat[b]]) for b in range(B)]
    batch = GpDeviceBatch(xs, ys)
    theta = np.tile(np.log([50.0, 20.0, 0.01]), (B, 1))
    batch.objective(theta)  # warm-up
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.evals):
        lml, grad, status = batch.objective(theta)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.evals
    flops = B * (float(n) ** 3)  # potrf n^3/3 + L^-T n^3/3 + K^-1 n^3/3
    out = {
        "metric": "GP objective evaluations/sec (LML + gradient, RBF+White kernel)", "value": B / (ms * 1e-3),
        "unit": "track-objectives/s", "n_gpus": 1, "ms_per_step": ms, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{B} tracks x {n} observations, K build + Cholesky + L^-T + K^-1-trace gradient "
                               "(BASELINE.json configs[4])"},
        "status_flagged": int((status != 0).sum()),
        "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                     "flops_per_objective": float(n) ** 3},
    }
    if (B, n) == (1000, 2000):
        # HBM bytes of one batched objective from the committed rocprofv3 --pmc summary of this command (all kernels of the
        # objective; FETCH_SIZE doubled per the gfx950 calibration, profiles/README.md)
        path = os.path.join(ROOT, "profiles", "r05_gp_counters_1000x2000.csv")
        try:
            import csv

            with open(path) as f:
                rows = list(csv.DictReader(ln for ln in f if not ln.startswith("#")))
            out["roofline"]["traffic"] = sum((float(r["hbm_read_GB"]) + float(r["hbm_write_GB"])) * 1e9 for r in rows)
            out["roofline"]["traffic_unit"] = "bytes per batched objective (all kernels)"
            out["roofline"]["traffic_source"] = os.path.relpath(path, ROOT)
            out["roofline"]["kernels_ms"] = {r["kernel"].replace("stegp::", ""): float(r["avg_ms"]) for r in rows}
        except (OSError, KeyError, ValueError):
            pass
    if args.cpu_evals > 0:
        from oracle import gp_oracle as gpo

        t0 = time.perf_counter()
        for k in range(args.cpu_evals):
            l, g, _, _ = gpo.lml_and_grad(theta[k], xs[k], ys[k])
        dt = (time.perf_counter() - t0) / args.cpu_evals
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "track-objectives/s", "cores": host_core_share(), "kind": "port",
                               "sample": f"{args.cpu_evals} objective evaluations at n={n} (oracle/gp_oracle.py: SciPy "
                                         "LAPACK cholesky/cho_solve, the same calls scikit-learn makes), BLAS threads = all it is given",
                               "gpu_vs_oracle_rel_err_lml": float(abs(lml[args.cpu_evals - 1] - l) / abs(l))}
        # scikit-learn itself -- the library the reference's GPRegression.fit hands its data to (gaussian_process.py:63-66) --
        # on the same objective: GaussianProcessRegressor.log_marginal_likelihood(theta, eval_gradient=True), the function
        # its L-BFGS-B calls ~50 times per restart
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel

        gpr = GaussianProcessRegressor(kernel=ConstantKernel(1.0) * RBF(1.0) + WhiteKernel(0.5), optimizer=None)
        gpr.fit(xs[0].reshape(-1, 1), ys[0])
        t0 = time.perf_counter()
        for k in range(args.cpu_evals):
            lk, gk = gpr.log_marginal_likelihood(theta[0], eval_gradient=True)
        dts = (time.perf_counter() - t0) / args.cpu_evals
        out["cpu_baseline"]["scikit_learn"] = {
            "value": 1.0 / dts, "unit": "track-objectives/s", "cores": host_core_share(), "kind": "reference",
            "sample": f"{args.cpu_evals} calls of scikit-learn {__import__('sklearn').__version__} GaussianProcessRegressor."
                      f"log_marginal_likelihood(theta, eval_gradient=True) at n={n} on track 0 ({dts:.2f} s each), BLAS threads = "
                      "all it is given",
            "gpu_vs_sklearn_rel_err_lml": float(abs(lml[0] - lk) / abs(lk)),
            "gpu_vs_sklearn_rel_err_grad": float(np.max(np.abs(grad[0] - gk) / np.maximum(np.abs(gk), 1e-12)))}
    if args.fit:
        from track_estimators.gaussian_processes import gaussian_process as gpm

        theta0 = np.log([1.0, 1.0, 0.5])  # 1.0 * RBF(1.0) + WhiteKernel(0.5), the reference example's kernel
        bounds = np.log(np.tile([1e-5, 1e5], (3, 1)))
        calls = [0]
        plain = batch.objective

        def counted(*a, **k):
            calls[0] += 1
            return plain(*a, **k)

        batch.objective = counted
        t0 = time.perf_counter()
        th, best = gpm.fit_thetas(batch, theta0, bounds, 0, None)
        torch.cuda.synchronize()
        t_fit = time.perf_counter() - t0
        batch.objective = plain
        out["fit"] = {"seconds": t_fit, "tracks": B, "tracks_per_s": B / t_fit, "batched_objective_launches": calls[0],
                      "theta_median": np.exp(np.median(th, axis=0)).tolist()}
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process.kernels import RBF, WhiteKernel

        t0 = time.perf_counter()
        ref = GaussianProcessRegressor(kernel=1.0 * RBF(1.0) + WhiteKernel(0.5)).fit(xs[0].reshape(-1, 1), ys[0])
        t_ref = time.perf_counter() - t0
        out["fit"]["cpu_reference"] = {
            "seconds_per_track": t_ref, "cores": host_core_share(), "kind": "reference",
            "sample": "scikit-learn GaussianProcessRegressor.fit on track 0, the call the reference's GPRegression.fit "
                      "makes (gaussian_process.py:63-66)",
            "lml_rel_diff_track0": float(abs(best[0] - ref.log_marginal_likelihood_value_)
                                         / abs(ref.log_marginal_likelihood_value_))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=1000)
    ap.add_argument("--nobs", type=int, default=2000)
    ap.add_argument("--evals", type=int, default=3)
    ap.add_argument("--cpu-evals", type=int, default=2)
    ap.add_argument("--fit", action="store_true",
                    help="also time a full hyper-parameter fit of every track (lock-step batched L-BFGS-B, no restarts) "
                         "beside scikit-learn's fit of one track on the host (SURVEY.md §8d config 5 (iii))")
    ap.add_argument("--fit-restarts", type=int, default=None,
                    help="instead: time a fit WITH this many restarts (run as extra batch entries) of --fit-tracks tracks x "
                         "--nobs observations on data with an interior optimum")
    ap.add_argument("--fit-tracks", type=int, default=64)
    ap.add_argument("--fit-lockstep-only", action="store_true", help="--fit-restarts: skip the restart-by-restart comparison run")
    ap.add_argument("--fit-no-cpu", action="store_true", help="--fit-restarts: skip scikit-learn's fit of one track on the host")
    a = ap.parse_args()
    if a.fit_restarts is not None:
        print(json.dumps({"metric": "GP fit with restarts (tracks/s)",
                          "fit": measure_fit(a.fit_tracks, a.nobs, a.fit_restarts, cpu=not a.fit_no_cpu,
                                             lockstep_only=a.fit_lockstep_only)}))
        return
    print(json.dumps(measure(a.tracks, a.nobs, a.evals, a.cpu_evals, a.fit)))


if __name__ == "__main__":
    main()

"""
Gaussian-process regression of (lon, lat) against time.  Mirrors reference
``track_estimators.gaussian_processes.gaussian_process.GPRegression``
(/root/reference/src/track_estimators/gaussian_processes/gaussian_process.py:10-89): same constructor, ``fit`` and
``predict`` signatures and return values.

The reference delegates every number to scikit-learn's ``GaussianProcessRegressor``.  Here the default regressor is
``DeviceGaussianProcessRegressor``, which keeps scikit-learn's *procedure* (L-BFGS-B from the kernel's initial theta
plus ``n_restarts_optimizer`` log-uniform restarts drawn from the same random stream, best optimum kept) but evaluates
the objective -- K build, Cholesky, alpha, log-marginal likelihood, K^-1, gradient -- and the predictions with the HIP
kernels of csrc/ste_gp.hip.  scikit-learn kernel objects are accepted as the description of the kernel (they carry the
initial hyper-parameters and bounds); only ``ConstantKernel * RBF + WhiteKernel`` (what the reference's examples use,
examples/example_gaussian_process_batch.py:41) has a device implementation.  No CPU fallback.

Additive extra: ``fit_batch`` / ``predict_batch`` fit many tracks at once, advancing all their optimisers in lock-step so
that each objective evaluation is one batched launch (the reference loops over ships in Python,
examples/example_gaussian_process_batch.py:19).
"""
from __future__ import annotations

import threading
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.optimize

from ..ship_track import ShipTrack
from .device import GpDeviceBatch


def _kernel_spec(kernel):
    """theta0 (3,), bounds (3,2) in log space for ConstantKernel * RBF + WhiteKernel given as scikit-learn objects."""
    from sklearn.gaussian_process import kernels as sk

    ok = (isinstance(kernel, sk.Sum) and isinstance(kernel.k1, sk.Product) and isinstance(kernel.k1.k1, sk.ConstantKernel)
          and isinstance(kernel.k1.k2, sk.RBF) and isinstance(kernel.k2, sk.WhiteKernel)
          and np.ndim(kernel.k1.k2.length_scale) == 0)
    if not ok:
        raise NotImplementedError("the HIP GP path implements ConstantKernel * RBF(isotropic) + WhiteKernel only "
                                  f"(got {kernel!r}); no CPU fallback")
    theta = np.asarray(kernel.theta, dtype=np.float64)
    bounds = np.asarray(kernel.bounds, dtype=np.float64)
    if theta.shape != (3,):
        raise NotImplementedError("fixed hyper-parameters are not supported on the HIP GP path")
    return theta, bounds


def _clone_with_theta(kernel, theta):
    return kernel.clone_with_theta(np.asarray(theta, dtype=np.float64))


class _LockstepObjective:
    """Lets B independent scipy optimisers (one thread each) share batched device evaluations."""

    def __init__(self, batch: GpDeviceBatch, thetas0: np.ndarray):
        self.batch = batch
        self.theta = np.array(thetas0, dtype=np.float64)
        self.cv = threading.Condition()
        self.pending = set()
        self.active = set(range(batch.B))
        self.generation = 0
        self.results = None
        self.nevals = 0

    def _launch_if_ready(self):
        if self.active and self.pending >= self.active:
            # only the tracks whose optimiser is still running are evaluated; the converged ones drop out of the launch
            subset = None if len(self.active) == self.batch.B else sorted(self.active)
            lml, grad, _ = self.batch.objective(self.theta, eval_gradient=True, active=subset)
            self.results = (lml, grad)
            self.pending.clear()
            self.generation += 1
            self.nevals += 1
            self.cv.notify_all()

    def evaluate(self, b: int, theta: np.ndarray):
        with self.cv:
            self.theta[b] = theta
            self.pending.add(b)
            gen = self.generation
            self._launch_if_ready()
            while self.generation == gen:
                self.cv.wait()
            lml, grad = self.results
            return -lml[b], -grad[b].copy()

    def finish(self, b: int):
        with self.cv:
            self.active.discard(b)
            self.pending.discard(b)
            self._launch_if_ready()


def _minimize_lockstep(batch: GpDeviceBatch, starts: np.ndarray, bounds: np.ndarray):
    """One L-BFGS-B run per track from starts[b]; returns (theta_opt[B,3], fun[B])."""
    B = batch.B
    shared = _LockstepObjective(batch, starts)
    out_theta = np.array(starts, dtype=np.float64)
    out_fun = np.full(B, np.inf)
    errors: List[Optional[BaseException]] = [None] * B

    def work(b):
        try:
            res = scipy.optimize.minimize(lambda th: shared.evaluate(b, th), starts[b], method="L-BFGS-B", jac=True,
                                          bounds=bounds)
            out_theta[b], out_fun[b] = res.x, res.fun
        except BaseException as exc:  # surfaced after the join
            errors[b] = exc
        finally:
            shared.finish(b)

    if B == 1:
        work(0)
    else:
        threads = [threading.Thread(target=work, args=(b,), daemon=True) for b in range(B)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    for e in errors:
        if e is not None:
            raise e
    return out_theta, out_fun


MAX_LOCKSTEP_ENTRIES = 1024  # optimisers advanced together (one Python thread each)
RESTART_HBM_BUDGET = 96 << 30  # bytes of HBM a group of restarts may take beside the batch itself


def fit_thetas(batch: GpDeviceBatch, theta0: np.ndarray, bounds: np.ndarray, n_restarts_optimizer: int, rng,
               optimizer="fmin_l_bfgs_b"):
    """scikit-learn's GaussianProcessRegressor.fit procedure for every track of the batch: L-BFGS-B from the kernel's
    theta, then ``n_restarts_optimizer`` more runs from log-uniform starts, best optimum kept (first one wins a tie, as
    in scikit-learn's loop).  ``rng``: one RandomState shared by the batch (starts drawn restart by restart, in track
    order) or a list of one RandomState per track (each track then sees exactly the stream a fit of its own would).

    The restarts do not depend on one another, so they run as extra batch entries that share their track's data: all
    (restarts + 1) x B optimisers advance in lock-step, one batched device launch per round, in groups bounded by HBM
    (``RESTART_HBM_BUDGET``) and by ``MAX_LOCKSTEP_ENTRIES`` -- instead of 51 sequential lock-step fits for the
    reference's default of 50 restarts (gaussian_process.py:50)."""
    B = batch.B
    if optimizer is None:
        thetas = np.tile(theta0, (B, 1))
        lml, _, _ = batch.objective(thetas, eval_gradient=False)
        return thetas, lml
    if optimizer != "fmin_l_bfgs_b":
        raise NotImplementedError("only optimizer='fmin_l_bfgs_b' or None are supported on the HIP GP path")
    starts = [np.tile(theta0, (B, 1))]
    if n_restarts_optimizer > 0:
        if not np.isfinite(bounds).all():
            raise ValueError("Multiple optimizer restarts (n_restarts_optimizer>0) requires that all bounds are finite.")
        rngs = rng if isinstance(rng, (list, tuple)) else [rng] * B
        for _ in range(n_restarts_optimizer):
            # scikit-learn draws one start per restart and fit; a batch draws B of them in track order
            starts.append(np.stack([rngs[b].uniform(bounds[:, 0], bounds[:, 1]) for b in range(B)]))
    per_group = max(1, min(len(starts), MAX_LOCKSTEP_ENTRIES // B, RESTART_HBM_BUDGET // max(1, batch.bytes_per_track * B)))
    best_theta, best_fun = None, None
    big = None
    for g0 in range(0, len(starts), per_group):
        group = starts[g0:g0 + per_group]
        if len(group) == 1:
            th, fun = _minimize_lockstep(batch, group[0], bounds)
            th, fun = th[None], fun[None]
        else:
            if big is None or big.B != len(group) * B:
                big = None  # release the previous group's buffers before the next allocation
                big = batch.replicated(len(group))
            th, fun = _minimize_lockstep(big, np.concatenate(group), bounds)
            th, fun = th.reshape(len(group), B, 3), fun.reshape(len(group), B)
        for r in range(len(group)):
            if best_theta is None:
                best_theta, best_fun = th[r].copy(), fun[r].copy()
            else:
                better = fun[r] < best_fun
                best_theta[better], best_fun[better] = th[r][better], fun[r][better]
    return best_theta, -best_fun


def _check_random_state(seed):
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError(f"{seed!r} cannot be used to seed a numpy.random.RandomState instance")


class DeviceGaussianProcessRegressor:
    """The subset of ``sklearn.gaussian_process.GaussianProcessRegressor`` the reference wrapper uses, on the GPU."""

    def __init__(self, kernel=None, *, alpha=1e-10, optimizer="fmin_l_bfgs_b", n_restarts_optimizer=0,
                 normalize_y=False, copy_X_train=True, n_targets=None, random_state=None):
        if normalize_y:
            raise NotImplementedError("normalize_y=True is not supported on the HIP GP path")
        self.kernel = kernel
        self.alpha = alpha
        self.optimizer = optimizer
        self.n_restarts_optimizer = n_restarts_optimizer
        self.normalize_y = normalize_y
        self.copy_X_train = copy_X_train
        self.n_targets = n_targets
        self.random_state = random_state

    def fit(self, X, y):
        X = np.asarray(X, dtype=np.float64)
        if X.ndim != 2 or X.shape[1] != 1:
            raise NotImplementedError("the HIP GP path takes one input column (time); got X of shape %s" % (X.shape,))
        y = np.asarray(y, dtype=np.float64).reshape(len(X), -1)
        theta0, bounds = _kernel_spec(self.kernel)
        self._rng = _check_random_state(self.random_state)
        self.X_train_, self.y_train_ = X.copy(), y.copy()
        self._batch = GpDeviceBatch([X[:, 0]], [y], jitter=float(self.alpha))
        thetas, lml = fit_thetas(self._batch, theta0, bounds, self.n_restarts_optimizer, self._rng, self.optimizer)
        self.kernel_ = _clone_with_theta(self.kernel, thetas[0])
        self.log_marginal_likelihood_value_ = float(self._batch.objective(thetas, eval_gradient=False)[0][0])
        self.alpha_ = self._batch.alpha()[0]
        return self

    def log_marginal_likelihood(self, theta=None, eval_gradient=False, clone_kernel=True):
        if theta is None:
            if eval_gradient:
                raise ValueError("Gradient can only be evaluated for theta!=None")
            return self.log_marginal_likelihood_value_
        lml, grad, _ = self._batch.objective(np.asarray(theta, dtype=np.float64)[None], eval_gradient=eval_gradient)
        return (float(lml[0]), grad[0]) if eval_gradient else float(lml[0])

    def predict(self, X, return_std=False, return_cov=False):
        if return_cov:
            raise NotImplementedError("return_cov is not supported on the HIP GP path")
        X = np.asarray(X, dtype=np.float64).reshape(-1)
        mean, std = self._batch.predict(self.kernel_.theta[None], [X])[0]
        if self.y_train_.shape[1] == 1:
            mean, std = mean[:, 0], std[:, 0]
        return (mean, std) if return_std else mean


class GPRegression:
    """
    Joint GP model of longitude and latitude against time (gaussian_process.py:10-26).

    Parameters
    ----------
    kernel
        scikit-learn kernel object, e.g. ``1.0 * RBF() + WhiteKernel(noise_level=0.5)``.
    gpr
        Regressor class; defaults to the GPU implementation (the reference's default is scikit-learn's class).
    """

    def __init__(self, kernel, gpr=DeviceGaussianProcessRegressor, *args, **kwargs):
        self._kernel = kernel
        self._gpr = gpr
        self._model = None

    @staticmethod
    def _training_data(ship_track: ShipTrack):
        times = np.insert(np.cumsum(ship_track.dts), 0, 0)  # gaussian_process.py:53-54
        return times.reshape(-1, 1), np.column_stack((ship_track.lon, ship_track.lat))  # :58,66

    def fit(self, ship_track: ShipTrack, gpr_kwargs: Optional[Dict[str, Any]] = None, *args, **kwargs):
        """Fit to one track; ``gpr_kwargs`` defaults to ``{"n_restarts_optimizer": 50}`` (gaussian_process.py:50)."""
        gpr_kwargs = gpr_kwargs or {"n_restarts_optimizer": 50}
        X, y = self._training_data(ship_track)
        assert self._kernel is not None, "Kernel must be specified."
        self._model = self._gpr(kernel=self._kernel, **gpr_kwargs)
        self._model.fit(X, y)
        return self._model

    def predict(self, times: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Posterior mean and standard deviation, each (m, 2), at ``times`` (gaussian_process.py:70-89)."""
        assert self._model is not None, "Model has not been fit yet."
        predicted, std = self._model.predict(np.asarray(times).reshape(-1, 1), return_std=True)
        return predicted, std

    # -- batched extras (not in the reference) ------------------------------------------------------------------
    def fit_batch(self, ship_tracks: Sequence[ShipTrack], gpr_kwargs: Optional[Dict[str, Any]] = None):
        """Fit every track with the same kernel description; returns the fitted thetas (B, 3) and lml (B,)."""
        gpr_kwargs = dict(gpr_kwargs or {"n_restarts_optimizer": 50})
        data = [self._training_data(st) for st in ship_tracks]
        theta0, bounds = _kernel_spec(self._kernel)
        seed = gpr_kwargs.get("random_state")
        # an integer seed means what it means for a loop of single fits (the reference's batch example builds one
        # regressor per ship): every track gets its own stream from that seed
        rng = [np.random.RandomState(seed) for _ in data] if isinstance(seed, (int, np.integer)) else _check_random_state(seed)
        self._batch = GpDeviceBatch([X[:, 0] for X, _ in data], [y for _, y in data],
                                    jitter=float(gpr_kwargs.get("alpha", 1e-10)))
        self._thetas, self._lml = fit_thetas(self._batch, theta0, bounds, int(gpr_kwargs.get("n_restarts_optimizer", 0)),
                                             rng, gpr_kwargs.get("optimizer", "fmin_l_bfgs_b"))
        return self._thetas, self._lml

    def predict_batch(self, times: Sequence[np.ndarray]):
        """[(mean (m_b, 2), std (m_b, 2)) for every track] after ``fit_batch``."""
        return self._batch.predict(self._thetas, [np.asarray(t).reshape(-1) for t in times])

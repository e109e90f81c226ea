"""
Unscented Kalman filter + unscented RTS smoother.  Mirrors reference
``track_estimators.kalman_filters.unscented.UnscentedKalmanFilter``
(/root/reference/src/track_estimators/kalman_filters/unscented.py:19-351): same constructor, methods, attributes and
error behaviour.  All filter arithmetic runs in the HIP kernels (csrc/) through the C ABI of include/ste.h; this
class only prepares inputs, draws the noise the reference draws, and reshapes outputs.  There is no CPU fallback.

Noise: the reference adds ``np.random.normal(scale=sqrt(diag(Q or R)), size=n)`` to the predicted mean, to every
consumed observation and to the smoother's back-prediction (unscented.py:198,232,320).  This class draws from the same
global NumPy generator with the same arguments in the same call order, so ``np.random.seed(s)`` yields the same stream
as it does for the reference.  Opt-in extra (not in the reference): set ``inject_noise = False`` (class or instance
attribute) to feed zeros instead.
"""
from __future__ import annotations

import ctypes as C
import logging
import types
from typing import Callable, Optional

import numpy as np

from .. import batch as _batch
from ..ship_track import ShipTrack
from .kalman_filter import KalmanFilterBase
from .non_linear_process import geodetic_dynamics as _geodetic_dynamics


def _dev():
    import torch

    return torch, torch.device("cuda", torch.cuda.current_device())


def _up(torch, dev, a, shape):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))).to(dev)


class UnscentedKalmanFilter(KalmanFilterBase):
    inject_noise = True

    def __init__(self, H=None, Q=None, R=None, P=None, x0=None, non_linear_process: Optional[Callable] = None,
                 measurement_model: Optional[Callable] = None):
        super().__init__()
        if H is None:
            raise ValueError("Set proper system dynamics.")
        self.H = H
        self.n = H.shape[1]
        self.Q = np.eye(self.n) if Q is None else np.asarray(Q)
        self.R = np.eye(self.n) if R is None else np.asarray(R)
        self.P_orig = np.eye(self.n) if P is None else np.asarray(P)
        self.P = np.eye(self.n) if P is None else np.asarray(P)
        self.x = np.zeros((self.n, 1)) if x0 is None else np.asarray(x0).reshape(-1, 1)
        self.n_sigma_points = 2 * self.n + 1
        self.sigma_points = np.zeros((self.n, self.n_sigma_points))
        self.sigma_points_orig = None
        self.weights = np.zeros((self.n_sigma_points, self.n_sigma_points))
        self.non_linear_process = non_linear_process
        self.measurement_model = measurement_model

    # -- helpers -----------------------------------------------------------------------------------------------
    def _require_dim4(self, what):
        if self.n != 4:
            raise NotImplementedError(
                f"{what}: the HIP path implements the 4-state model [lon, lat, speed, heading] only (n={self.n}); "
                "the reference itself hard-codes the heading at index 3 (unscented.py:250). No CPU fallback.")

    def _require_geodetic(self, fn):
        assert fn is not None, "Non-linear process is not set."
        assert callable(fn), "Non-linear process model must be callable."
        if fn is not _geodetic_dynamics:
            raise NotImplementedError("the HIP path implements track_estimators.kalman_filters.non_linear_process."
                                      "geodetic_dynamics only; arbitrary Python process models have no device form")

    def _draw(self, cov):
        """One reference-style draw (same generator, arguments and shape as unscented.py:198-200)."""
        if self.inject_noise:
            return np.random.normal(scale=np.sqrt(np.diag(cov)), size=(self.n))
        return np.zeros(self.n)

    def _measure(self, z):
        z = np.asarray(z, dtype=np.float64).reshape(-1, 1)
        if self.measurement_model is not None:
            assert callable(self.measurement_model), "Measurement model must be callable."
            z = self.measurement_model(z)
        return np.asarray(z, dtype=np.float64).reshape(-1)

    def _fan_constants(self):
        n = self.n
        w0 = self.weights[0, 0]
        return n / (1 - w0), w0, self.weights[1, 1]

    # -- sigma points / weights (unscented.py:76-142) -------------------------------------------------------------
    def compute_sigma_points(self, x: Optional[np.ndarray] = None, P: Optional[np.ndarray] = None) -> np.ndarray:
        """Sigma fan ``x, x +/- columns of sqrtm(n/(1-W0) P)`` as an (n, 2n+1) array, computed on the GPU."""
        from .._hip import binding

        if x is None:
            assert self.x is not None, "Set proper initial state estimate."
            x = self.x
        if P is None:
            assert self.P is not None, "Set proper initial state covariance matrix."
            P = self.P
        n = self.n
        lib = binding.require_gpu()
        torch, dev = _dev()
        xs, Ps = _up(torch, dev, x, (n, 1)), _up(torch, dev, P, (n * n, 1))
        out = torch.empty((2 * n + 1, n, 1), dtype=torch.float64, device=dev)
        scale = n / (1 - self.weights[0, 0])
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if n == 4:
            binding.check(lib.ste_sigma_points_f64(1, xs.data_ptr(), Ps.data_ptr(), C.c_double(scale), out.data_ptr(),
                                                   stream), "ste_sigma_points_f64")
        else:
            binding.check(lib.ste_sigma_points_generic_f64(n, 1, xs.data_ptr(), Ps.data_ptr(), C.c_double(scale),
                                                           out.data_ptr(), stream), "ste_sigma_points_generic_f64")
        self.sigma_points[:, :] = out.cpu().numpy()[:, :, 0].T
        return self.sigma_points

    def compute_weights(self, weight0: Optional[float] = None) -> np.ndarray:
        """Diagonal (2n+1)x(2n+1) weight matrix: W0 = 1 - n/3 (or ``weight0``), Wi = (1 - W0)/(2n)."""
        if weight0 is None:
            weight0 = 1 - self.n / 3.0
        assert weight0 < 1.0 and weight0 > -1.0, "Weight0 value ({}) is outside [-1, 1] range.".format(weight0)
        weightn = (1 - weight0) / (2 * self.n)
        np.fill_diagonal(self.weights, weightn)
        self.weights[0, 0] = weight0
        logging.debug("Weights\n\n%s", self.weights)
        return self.weights

    # -- single-step API (unscented.py:144-265) -------------------------------------------------------------------
    def predict(self, non_linear_process: Optional[Callable] = None, **non_linear_process_kwargs) -> None:
        """One predict step (unscented.py:144-207).  kwargs: ``dt``, ``c`` (must be None), ``sog_rate``, ``cog_rate``."""
        from .._hip import binding

        if non_linear_process is None:
            assert self.non_linear_process is not None, "Non-linear process is not set."
            non_linear_process = self.non_linear_process
        self._require_geodetic(non_linear_process)
        self._require_dim4("predict")
        kw = dict(non_linear_process_kwargs)
        c = kw.pop("c", None)
        if c is not None and np.size(c):
            raise NotImplementedError("control vector c must be None on the HIP path (the reference passes None, "
                                      "kalman_filter.py:92)")
        dt = kw.pop("dt")
        sr, cr = kw.pop("sog_rate", 0.0), kw.pop("cog_rate", 0.0)
        if kw:
            raise TypeError(f"unexpected process-model arguments: {sorted(kw)}")
        self.x = np.asarray(self.x).reshape(-1, 1)
        self.compute_weights()
        fan_scale, w0, wi = self._fan_constants()
        noise = self._draw(self.Q)
        lib = binding.require_gpu()
        torch, dev = _dev()
        x, P = _up(torch, dev, self.x, (4, 1)), _up(torch, dev, self.P, (16, 1))
        d, a, b = (_up(torch, dev, v, (1,)) for v in (dt, sr, cr))
        nz = _up(torch, dev, noise, (4, 1))
        Q = _batch._as44(self.Q, "Q", True)
        _batch.require_symmetric(np.asarray(self.P, dtype=np.float64).reshape(4, 4), "P")
        xo, Po = torch.empty_like(x), torch.empty_like(P)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        binding.check(lib.ste_ukf_predict_f64(1, x.data_ptr(), P.data_ptr(), d.data_ptr(), a.data_ptr(), b.data_ptr(),
                                              nz.data_ptr(), Q.ctypes.data, fan_scale, w0, wi, xo.data_ptr(),
                                              Po.data_ptr(), None, stream), "ste_ukf_predict_f64")
        self.x = xo.cpu().numpy().reshape(4, 1)
        self.P = Po.cpu().numpy().reshape(4, 4)

    def update(self, z: np.ndarray) -> None:
        """One measurement update (unscented.py:209-265): pinv gain, heading wrap, Joseph-form covariance."""
        from .._hip import binding

        self._require_dim4("update")
        zz = self._measure(z)
        noise = self._draw(self.R)
        lib = binding.require_gpu()
        torch, dev = _dev()
        x, P = _up(torch, dev, self.x, (4, 1)), _up(torch, dev, self.P, (16, 1))
        zt, nz = _up(torch, dev, zz, (4, 1)), _up(torch, dev, noise, (4, 1))
        H, R = _batch._as44(self.H, "H"), _batch._as44(self.R, "R", True)
        _batch.require_symmetric(np.asarray(self.P, dtype=np.float64).reshape(4, 4), "P")
        xo, Po = torch.empty_like(x), torch.empty_like(P)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        binding.check(lib.ste_ukf_update_f64(1, x.data_ptr(), P.data_ptr(), zt.data_ptr(), nz.data_ptr(), H.ctypes.data,
                                             R.ctypes.data, xo.data_ptr(), Po.data_ptr(), None, stream),
                      "ste_ukf_update_f64")
        self.x = xo.cpu().numpy().reshape(4, 1)
        self.P = Po.cpu().numpy().reshape(4, 4)

    # -- whole-track API ------------------------------------------------------------------------------------------
    def _launch_forward(self, dt, ship_track):
        """Forward pass of ``run`` for this one track on the GPU (kalman_filter.py:61-117)."""
        self._require_geodetic(self.non_linear_process)
        self._require_dim4("run")
        self.compute_weights()
        N = len(dt)
        z = np.asarray(ship_track.z, dtype=np.float64)
        if self.measurement_model is not None:
            z = np.stack([self._measure(z[:, j]) for j in range(z.shape[1])], axis=1)
        upd_idx, _, t_end = _batch.update_schedule(dt, ship_track.dts, self.time)
        # noise in the reference's call order: initial update, then per step predict (+ update if it fires)
        npred = np.zeros((N, 4))
        nupd = np.zeros((N + 1, 4))
        nupd[0] = self._draw(self.R)
        for k in range(N):
            npred[k] = self._draw(self.Q)
            if upd_idx[k] >= 0:
                nupd[k + 1] = self._draw(self.R)
        trk = types.SimpleNamespace(z=z, dts=ship_track.dts, sog_rate=ship_track.sog_rate, cog_rate=ship_track.cog_rate)
        hb = _batch.pack_tracks([trk], [dt], [np.asarray(self.x, dtype=np.float64).reshape(-1)], self.H, self.Q, self.R,
                                np.asarray(self.P, dtype=np.float64), t0s=[self.time],
                                noise=[dict(noise_pred=npred, noise_upd=nupd, noise_rts=np.zeros((N, 4)))],
                                on_error="raise")
        out = _batch.run_batch(hb, smooth=False)
        self._status = int(out["status"][0])
        if self._status & 0x1:
            # the reference dies inside np.linalg.pinv once a NaN/inf reaches the covariance (e.g. dt = 0 -> sog = inf)
            raise np.linalg.LinAlgError("SVD did not converge")
        return out["means"][0], out["covs"][0], upd_idx, t_end

    def rts_step(self, fwd_means, fwd_vars, ship_track: ShipTrack, *args, **kwargs):
        """
        Unscented RTS smoother over a stored forward history (unscented.py:267-351).

        ``fwd_means`` (N+1, n, 1) or (N+1, n); ``fwd_vars`` (N+1, n, n).  Like the reference this expands
        ``ship_track.sog_rate`` / ``cog_rate`` in place with ``np.repeat`` (:287-292), so a second call on the same
        ShipTrack sees the already-expanded arrays.
        """
        self._require_geodetic(self.non_linear_process)
        self._require_dim4("rts_step")
        fwd_means = np.asarray(fwd_means, dtype=np.float64)
        nrows = fwd_means.shape[0]
        rep = int(nrows / len(ship_track.dts))
        ship_track.sog_rate = np.repeat(ship_track.sog_rate, rep)
        ship_track.cog_rate = np.repeat(ship_track.cog_rate, rep)
        N = nrows - 1
        dt = np.asarray(self.dt, dtype=np.float64)
        if N > len(dt) or N > len(ship_track.sog_rate):
            raise IndexError("index out of bounds: smoother needs dt and expanded rates for every stored step")
        self.compute_weights()
        nrts = np.zeros((N, 4))
        for k in range(N - 1, -1, -1):  # the reference draws while walking backwards (:297,320)
            nrts[k] = self._draw(self.Q)
        from .._hip import binding

        lib = binding.require_gpu()
        torch, dev = _dev()
        m = fwd_means.reshape(nrows, 4)
        _batch.require_symmetric(np.asarray(fwd_vars, dtype=np.float64).reshape(nrows, 4, 4), "fwd_vars")
        hb = _batch.HostBatch(
            B=1, Nmax=N, Tmax=1, H=_batch._as44(self.H, "H"), Q=_batch._as44(self.Q, "Q", True), R=_batch._as44(self.R, "R", True),
            nsteps=np.array([N], dtype=np.int32), x0=np.ascontiguousarray(m[0].reshape(4, 1)),
            P0=np.ascontiguousarray(np.asarray(fwd_vars[0], dtype=np.float64).reshape(16)),
            dt=np.ascontiguousarray(dt[:N].reshape(N, 1)),
            sog_rate=np.ascontiguousarray(np.asarray(ship_track.sog_rate[:N], dtype=np.float64).reshape(N, 1)),
            cog_rate=np.ascontiguousarray(np.asarray(ship_track.cog_rate[:N], dtype=np.float64).reshape(N, 1)),
            sog_rate_rts=None, cog_rate_rts=None, upd_idx=np.full((N, 1), -1, dtype=np.int32), z=np.zeros((1, 4, 1)),
            noise_pred=None, noise_upd=None, noise_rts=nrts.reshape(N, 4, 1))
        # the history comes from the caller, not from a forward launch on this batch: no precomputed gains
        db = _batch.DeviceBatch(hb, fuse_gains=False, packed_cov=False)  # the caller's full matrices go in as they are
        db.fwd_mean.copy_(_up(torch, dev, m, (nrows, 4, 1)))
        db.fwd_cov.copy_(_up(torch, dev, fwd_vars, (nrows, 16, 1)))
        db.backward()
        torch.cuda.synchronize(dev)
        sm, sP = db.smoothed()
        self._status_smoother = int(db.status_host()[0])
        if self._status_smoother & 0x1:
            raise np.linalg.LinAlgError("SVD did not converge")
        return sm[0].reshape(nrows, 4, 1), sP[0]

    # -- robustification helpers (unscented.py:353-511) -------------------------------------------------------------
    # The reference's call site is commented out (unscented.py:228), so ``run`` never invokes these; they are kept as
    # callable methods with the reference's signatures.  The batched path offers the same loop as an opt-in flag
    # (``HostBatch.robust``), evaluated inside the update kernel.
    def _robust_terms(self, z, P, R):
        """(gamma, denom) of unscented.py:420-426 and :468-478 for this filter's x and H, computed on the GPU."""
        from .._hip import binding

        self._require_dim4("robustification")
        lib = binding.require_gpu()
        torch, dev = _dev()
        x, Pm = _up(torch, dev, self.x, (4, 1)), _up(torch, dev, P, (16, 1))
        zt = _up(torch, dev, np.asarray(z, dtype=np.float64).reshape(-1), (4, 1))
        H, Rm = _batch._as44(self.H, "H"), _batch._as44(R, "R", True)
        _batch.require_symmetric(np.asarray(P, dtype=np.float64).reshape(4, 4), "P")
        g, d = torch.empty((1,), dtype=torch.float64, device=dev), torch.empty((1,), dtype=torch.float64, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        binding.check(lib.ste_ukf_robust_terms_f64(1, x.data_ptr(), Pm.data_ptr(), zt.data_ptr(), H.ctypes.data,
                                                   Rm.ctypes.data, g.data_ptr(), d.data_ptr(), stream),
                      "ste_ukf_robust_terms_f64")
        return float(g.item()), float(d.item())

    def criterion_index(self, z: np.ndarray, P: np.ndarray, R: np.ndarray) -> float:
        """Mahalanobis judging index ``|y^T (H P H^T + R)^+ y|`` with ``y = z - x`` (unscented.py:389-428)."""
        return self._robust_terms(z, P, R)[0]

    def update_lambda_factor(self, lambda_factor: float, criterion_index: float, chi_alpha: float, z: np.ndarray,
                             P: np.ndarray, R: np.ndarray) -> float:
        """``lambda + (criterion - chi_alpha) / (y^T S^+ R S^+ y)`` (unscented.py:430-483)."""
        _, denom = self._robust_terms(z, P, R)
        return float(lambda_factor + (criterion_index - chi_alpha) / denom)

    def scale_measurement_uncertainty(self, R: np.ndarray, lambda_factor: float) -> np.ndarray:
        """``R * lambda`` (unscented.py:485-511)."""
        return R * lambda_factor

    def check_robustness(self, z: np.ndarray, P: np.ndarray, R: np.ndarray) -> np.ndarray:
        """The reference's rescaling loop (unscented.py:353-387), including its fresh noise draw per iteration and its
        prints; returns the scaled R."""
        lambda_factor = 1
        chi_alpha = 50
        z = np.asarray(z, dtype=np.float64).reshape(-1, 1)
        zn = z + self._draw(R).reshape(-1, 1)
        judging_index = self.criterion_index(zn, P, R)
        print(judging_index, lambda_factor)
        while judging_index > chi_alpha:
            zn = z + self._draw(R).reshape(-1, 1)
            lambda_factor = self.update_lambda_factor(lambda_factor, judging_index, chi_alpha, zn, P, R)
            R = self.scale_measurement_uncertainty(R, lambda_factor)
            judging_index = self.criterion_index(zn, P, R)
            print(judging_index, lambda_factor)
        return R

"""
Device side of the GP-regression path: owns the HBM buffers of a batch of tracks and calls the second kernel set
(csrc/ste_gp.hip) through the C ABI (include/ste.h: ste_gp_*).  PyTorch is used for device memory only.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from .._hip import binding

JITTER = 1e-10  # GaussianProcessRegressor(alpha=1e-10), the value the reference's wrapper leaves at its default


class GpDeviceBatch:
    """B tracks (1-D inputs x_b, outputs y_b (n_b, nout)) resident on the GPU; evaluates the GP objective for all."""

    def __init__(self, xs: Sequence[np.ndarray], ys: Sequence[np.ndarray], device="cuda:0", jitter: float = JITTER,
                 inverse_order: int = binding.STE_GP_INVERSE_AUTO):
        import torch

        self.torch = torch
        self.lib = binding.require_gpu()
        self.device = torch.device(device)
        self.B = len(xs)
        if self.B == 0:
            raise ValueError("empty batch")
        self._xs, self._ys, self._jitter = list(xs), list(ys), float(jitter)
        self.n = np.array([len(x) for x in xs], dtype=np.int32)
        self.nout = int(np.asarray(ys[0]).reshape(len(xs[0]), -1).shape[1])
        self.nmax = int(self.n.max())
        self.nb = (self.nmax + 63) // 64
        self.ld = 64 * self.nb + 16  # STE_GP_LD_PAD (include/ste.h)
        B, nmax, ld, nout = self.B, self.nmax, self.ld, self.nout
        xh = np.zeros((B, nmax))
        yh = np.zeros((B, nout, nmax))
        for b in range(B):
            xb = np.asarray(xs[b], dtype=np.float64).reshape(-1)
            yb = np.asarray(ys[b], dtype=np.float64).reshape(len(xb), nout)
            xh[b, : len(xb)] = xb
            yh[b, :, : len(xb)] = yb.T
        f64 = dict(dtype=torch.float64, device=self.device)
        self.t_n = torch.from_numpy(self.n).to(self.device)
        self.t_x = torch.from_numpy(xh).to(self.device)
        self.t_y = torch.from_numpy(yh).to(self.device)
        self.t_theta = torch.zeros((B, 3), **f64)
        self.t_K = torch.empty((B, ld, ld), **f64)
        self.t_U = torch.empty((B, ld, ld), **f64)
        self.t_Dinv = torch.empty((B, self.nb, 64, 64), **f64)
        self.t_Kinv = None
        self.t_alpha = torch.zeros((B, nout, nmax), **f64)
        self.t_lml = torch.zeros((B,), **f64)
        self.t_grad = torch.zeros((B, 3), **f64)
        nbm = self.nb
        self.t_tr = torch.zeros((B, 3, nbm * (nbm + 1) // 2), **f64)
        self.t_status = torch.zeros((B,), dtype=torch.int32, device=self.device)
        s = binding.SteGpBatchF64()
        s.B, s.nmax, s.nout, s.jitter = B, nmax, nout, float(jitter)
        s.n, s.x, s.y, s.theta = self.t_n.data_ptr(), self.t_x.data_ptr(), self.t_y.data_ptr(), self.t_theta.data_ptr()
        s.K, s.U, s.Dinv = self.t_K.data_ptr(), self.t_U.data_ptr(), self.t_Dinv.data_ptr()
        s.Kinv = None
        s.alpha, s.lml, s.grad, s.tr = (self.t_alpha.data_ptr(), self.t_lml.data_ptr(), self.t_grad.data_ptr(),
                                        self.t_tr.data_ptr())
        s.status = self.t_status.data_ptr()
        # The kernel that forms L^-T is named explicitly once resolved, so that a replicated copy of this batch (whose B is
        # a multiple of this one's and may cross the library's size threshold) sums in the same order: a track's first
        # optimiser start and its restarts then see bit-identical objectives for identical theta.
        if inverse_order == binding.STE_GP_INVERSE_AUTO:
            inverse_order = binding.STE_GP_INVERSE_COLS if B >= 128 else binding.STE_GP_INVERSE_ROWS
        self.inverse_order = s.inverse_order = int(inverse_order)
        self.struct = s

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def bytes_per_track(self) -> int:
        """HBM held per track by this batch (K, U and the small arrays)."""
        return 8 * (2 * self.ld * self.ld + self.nb * 64 * 64 + (self.nout + 2) * self.nmax + 3 * self.nb * (self.nb + 1) // 2)

    def replicated(self, copies: int) -> "GpDeviceBatch":
        """A batch holding ``copies`` consecutive copies of this one's tracks (entry c * B + b is track b): the optimiser
        restarts of a fit run as extra batch entries that share their track's data."""
        return GpDeviceBatch(self._xs * copies, self._ys * copies, device=self.device, jitter=self._jitter,
                             inverse_order=self.inverse_order)

    def _set_theta(self, thetas):
        th = np.ascontiguousarray(np.asarray(thetas, dtype=np.float64).reshape(self.B, 3))
        self.t_theta.copy_(self.torch.from_numpy(th))

    def objective(self, thetas, eval_gradient: bool = True, keep_kinv: bool = False, active=None):
        """(lml[B], grad[B,3] or None, status[B]) at thetas[B,3] = log(constant, length_scale, noise).

        ``active``: optional sequence of track indices; only those are evaluated (``ste_gp_lml_subset_f64``) and the
        entries of the other tracks in the returned arrays are whatever their last evaluation left."""
        self._set_theta(thetas)
        s = self.struct
        s.grad = self.t_grad.data_ptr() if eval_gradient else None
        if keep_kinv:
            if self.t_Kinv is None:
                self.t_Kinv = self.torch.empty((self.B, self.ld, self.ld), dtype=self.torch.float64, device=self.device)
            s.Kinv = self.t_Kinv.data_ptr()
        else:
            s.Kinv = None
        if active is None:
            binding.check(self.lib.ste_gp_lml_f64(C.byref(s), self._stream()), "ste_gp_lml_f64")
        else:
            idx = np.unique(np.asarray(list(active), dtype=np.int32))
            if len(idx) and (idx[0] < 0 or idx[-1] >= self.B):
                raise IndexError("active track index out of range")
            t_idx = self.torch.from_numpy(idx).to(self.device)
            binding.check(self.lib.ste_gp_lml_subset_f64(C.byref(s), len(idx), t_idx.data_ptr(), self._stream()),
                          "ste_gp_lml_subset_f64")
        lml = self.t_lml.cpu().numpy()
        grad = self.t_grad.cpu().numpy() if eval_gradient else None
        status = self.t_status.cpu().numpy()
        lml = np.where(status != 0, -np.inf, lml)
        if grad is not None:
            grad = np.where((status != 0)[:, None], 0.0, grad)
        return lml, grad, status

    def kmatrix(self, thetas):
        """K(X,X) + (noise + jitter) I of every track as (B, ld, ld) NumPy (lower tiles valid)."""
        self._set_theta(thetas)
        binding.check(self.lib.ste_gp_rbf_kmatrix_f64(C.byref(self.struct), self._stream()), "ste_gp_rbf_kmatrix_f64")
        return self.t_K.cpu().numpy()

    def cholesky(self):
        """Factor the K currently in the buffer; returns (L (B, ld, ld) lower, status)."""
        binding.check(self.lib.ste_gp_potrf_f64(C.byref(self.struct), self._stream()), "ste_gp_potrf_f64")
        return np.tril(self.t_K.cpu().numpy()), self.t_status.cpu().numpy()

    def alpha(self):
        return [self.t_alpha[b, :, : self.n[b]].cpu().numpy().T for b in range(self.B)]

    def predict(self, thetas, xq: Sequence[np.ndarray]):
        """Posterior (mean (m_b, nout), std (m_b, nout)) per track at query inputs xq[b]."""
        torch = self.torch
        self.objective(thetas, eval_gradient=False, keep_kinv=True)
        m = np.array([len(q) for q in xq], dtype=np.int32)
        mmax = int(m.max())
        mb = (mmax + 63) // 64
        xs = np.zeros((self.B, mmax))
        for b in range(self.B):
            xs[b, : m[b]] = np.asarray(xq[b], dtype=np.float64).reshape(-1)
        t_m = torch.from_numpy(m).to(self.device)
        t_xs = torch.from_numpy(xs).to(self.device)
        t_ks = torch.empty((self.B, mb * 64, self.ld), dtype=torch.float64, device=self.device)
        t_mean = torch.zeros((self.B, self.nout, mmax), dtype=torch.float64, device=self.device)
        t_var = torch.zeros((self.B, mmax), dtype=torch.float64, device=self.device)
        binding.check(self.lib.ste_gp_predict_f64(C.byref(self.struct), mmax, t_m.data_ptr(), t_xs.data_ptr(),
                                                  t_ks.data_ptr(), t_mean.data_ptr(), t_var.data_ptr(), self._stream()),
                      "ste_gp_predict_f64")
        mean = t_mean.cpu().numpy()
        var = t_var.cpu().numpy()
        out = []
        for b in range(self.B):
            v = var[b, : m[b]]
            v = np.where(v < 0, 0.0, v)  # GaussianProcessRegressor.predict clips negative variances to 0
            std = np.sqrt(v)
            out.append((mean[b, :, : m[b]].T.copy(), np.repeat(std[:, None], self.nout, axis=1)))
        return out

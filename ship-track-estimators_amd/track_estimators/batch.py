"""
Batched UKF + URTSS over many independent ship tracks on one MI355X.

This is the host side of the hot path: it turns B ``ShipTrack``-like objects into the structure-of-arrays batch
that include/ste.h describes (track index fastest), precomputes the per-step update schedule with the same float
accumulation the reference driver performs (kalman_filter.py:73,98-102), launches the HIP kernels through the C ABI
and hands back per-track histories shaped like the reference's return values.

PyTorch is used only for device memory and streams.  There is no CPU fallback.

Reference call sites replaced (relative to /root/reference):
  examples/example_ukf_rts_smoother_batch.py:19-90   the per-ship Python loop -> one batched launch
  src/track_estimators/kalman_filters/kalman_filter.py:36-137   run / run_rts_smoother
"""
from __future__ import annotations

import atexit
import ctypes as C
import dataclasses
import weakref
from typing import Optional, Sequence

import numpy as np

from ._hip import binding


# ----------------------------------------------------------------------------------------------------------------
# host-side logic (pure NumPy, no arithmetic of the filter itself)
# ----------------------------------------------------------------------------------------------------------------
def sigma_constants(n: int, weights_computed: bool = True):
    """(fan_scale, w0, wi) as unscented.py:95,125,132 computes them in Python floats."""
    w0 = 1 - n / 3.0
    wi = (1 - w0) / (2 * n)
    fan_scale = n / (1 - (w0 if weights_computed else 0.0))
    return fan_scale, w0, wi


def _isin_exact(times: np.ndarray, cums: np.ndarray) -> np.ndarray:
    """``np.isin(times, cums)`` for 1-D float arrays (exact equality, NaN never a member) by sort + binary search; an
    order of magnitude cheaper per call than np.isin's concatenate/unique path, which matters once per track."""
    if len(cums) == 0 or len(times) == 0:
        return np.zeros(len(times), dtype=bool)
    cs = np.sort(cums)
    pos = np.searchsorted(cs, times)
    np.minimum(pos, len(cs) - 1, out=pos)
    return cs[pos] == times


def update_schedule(dt, dts, t0=0):
    """
    Float-equality update trigger of the reference driver (kalman_filter.py:73,98-102), precomputed.

    Returns ``upd_idx`` (int32[N]: observation column consumed after step k, -1 = none), ``rate_idx`` (int64[N]:
    index into sog_rate/cog_rate that predict uses at step k, kalman_filter.py:93-94) and the final time.
    The running time is accumulated by sequential float addition exactly like ``self.time += dt``.
    """
    dt = np.asarray(dt, dtype=np.float64)
    cums = np.cumsum(np.asarray(dts, dtype=np.float64))
    times = np.cumsum(np.concatenate([[np.float64(t0)], dt]))[1:]
    fires = _isin_exact(times, cums)
    after = np.cumsum(fires)
    upd_idx = np.where(fires, after, -1).astype(np.int32)
    rate_idx = after - fires
    return upd_idx, rate_idx, (times[-1] if len(times) else np.float64(t0))


def rts_rate_index(nrows: int, dts_len: int, T: int) -> np.ndarray:
    """Index into the length-T rate arrays read by the smoother at step k (unscented.py:287-292, 310-311)."""
    rep = int(nrows / dts_len)
    idx = np.repeat(np.arange(T), rep)
    if len(idx) < nrows - 1:
        raise IndexError(
            f"smoother rate expansion too short: np.repeat(rate[{T}], {rep}) has {len(idx)} entries for {nrows - 1} steps"
        )
    return idx[: nrows - 1]


def draw_reference_noise(Q, R, dt, dts, t0=0):
    """
    Noise arrays for one track drawn from NumPy's global generator with the reference's calls in the reference's order
    (unscented.py:198,232,320): initial update, then per step the predict draw (+ an update draw when the trigger
    fires), then the smoother's draws for k = N-1 ... 0.  Returns the dict ``pack_tracks(noise=[...])`` expects.
    """
    N = len(dt)
    upd_idx, _, _ = update_schedule(dt, dts, t0)
    sq, sr = np.sqrt(np.diag(Q)), np.sqrt(np.diag(R))
    npred, nupd, nrts = np.zeros((N, 4)), np.zeros((N + 1, 4)), np.zeros((N, 4))
    nupd[0] = np.random.normal(scale=sr, size=(4))
    for k in range(N):
        npred[k] = np.random.normal(scale=sq, size=(4))
        if upd_idx[k] >= 0:
            nupd[k + 1] = np.random.normal(scale=sr, size=(4))
    for k in range(N - 1, -1, -1):
        nrts[k] = np.random.normal(scale=sq, size=(4))
    return dict(noise_pred=npred, noise_upd=nupd, noise_rts=nrts)


@dataclasses.dataclass
class HostBatch:
    """NumPy image of ``struct ste_ukf_batch_f64``; arrays are C-contiguous with the track index last."""

    B: int
    Nmax: int
    Tmax: int
    H: np.ndarray
    Q: np.ndarray
    R: np.ndarray
    nsteps: np.ndarray  # (B,) int32
    x0: np.ndarray  # (4, B)
    P0: np.ndarray  # (16,) shared or (16, B)
    dt: np.ndarray  # (Nmax, B)
    sog_rate: np.ndarray
    cog_rate: np.ndarray
    sog_rate_rts: Optional[np.ndarray]
    cog_rate_rts: Optional[np.ndarray]
    upd_idx: np.ndarray  # (Nmax, B) int32
    z: np.ndarray  # (Tmax, 4, B)
    noise_pred: Optional[np.ndarray] = None  # (Nmax, 4, B)
    noise_upd: Optional[np.ndarray] = None  # (Nmax+1, 4, B)
    noise_rts: Optional[np.ndarray] = None  # (Nmax, 4, B)
    weights_computed: bool = True
    initial_update: bool = True
    robust: bool = False  # opt-in Mahalanobis robust update (not on the reference's shipped path)
    chi_alpha: float = 50.0
    host_status: Optional[np.ndarray] = None  # (B,) int32 bits set while packing (STATUS_HOST_INDEX)
    order: Optional[np.ndarray] = None  # (B,) batch slot -> index of the caller's track (length-bucketed packing)
    lanes: Optional[int] = None  # forward-kernel lane mapping for this batch: 1, 4, 0 = by batch size; None = default_lanes

    @property
    def shared_p0(self) -> bool:
        return self.P0.ndim == 1

    @property
    def track_steps(self) -> int:
        return int(self.nsteps.sum())


SYMMETRY_RTOL = 1e-12  # |M - M^T| above this fraction of max|M| is "not symmetric" (rounding leaves ~1e-16)


def require_symmetric(M, name):
    """Raise ValueError unless every (.., 4, 4) matrix in ``M`` is symmetric to rounding.

    The kernels treat Q, R and every covariance as symmetric matrices (symmetric square root, eigenvalue pseudo-inverse,
    packed triangles); the reference calls scipy.linalg.sqrtm / np.linalg.pinv on whatever it is given, and for a
    non-symmetric argument those are a different computation (Schur form, SVD).  Such input is not a covariance, so it
    is refused rather than silently symmetrised (hard part 1)."""
    M = np.asarray(M, dtype=np.float64)
    asym = np.abs(M - np.swapaxes(M, -1, -2))
    scale = np.max(np.abs(M), axis=(-1, -2), keepdims=True)
    bad = ~(asym <= SYMMETRY_RTOL * scale) & np.isfinite(M) & np.isfinite(np.swapaxes(M, -1, -2))
    if np.any(bad):
        raise ValueError(f"{name} must be symmetric: max |{name} - {name}^T| = {float(np.max(np.where(bad, asym, 0.0))):.3e}; "
                         "the HIP path works on symmetric covariances (scipy.linalg.sqrtm / np.linalg.pinv of a "
                         "non-symmetric matrix, unscented.py:97,243, is a different computation)")
    return M


def _as44(M, name, symmetric: bool = False):
    M = np.ascontiguousarray(np.asarray(M, dtype=np.float64))
    if M.shape != (4, 4):
        raise ValueError(f"{name} must be 4x4 for the HIP path (got {M.shape}); the reference hard-codes the heading at "
                         "index 3 (unscented.py:250)")
    if symmetric:
        require_symmetric(M, name)
    return M


STATUS_HOST_INDEX = binding.STE_STATUS_HOST_INDEX  # the reference would raise IndexError for this track (update index past the last observation)

# Lane mapping used by batches that do not name one (HostBatch.lanes is None): 0 = the library picks by batch size.
# A per-call flag of the C ABI underneath (STE_FLAG_LANES_1 / _4); the tests set this to run every case in both mappings.
default_lanes = 0
# ste_ukf_batch_f64.tuning of batches that do not name one (DeviceBatch(tuning=0)).  Bits (include/ste.h): 0x100 every smoother
# gain by the eigenvalue route, 0x200 / 0x400 the two-kernel / one-kernel smoother whatever the batch size, 0x800 the
# two-kernel form's recurrence with a lane instead of a quad per track.  Tests set it.
default_tuning = 0


def pack_tracks(tracks: Sequence, dts_per_track: Sequence, x0s: Sequence, H, Q, R, P0, t0s=None,
                noise: Optional[Sequence[dict]] = None, on_error: str = "flag",
                bucket_by_length: bool = True) -> HostBatch:
    """
    Pack B tracks (objects carrying ``z`` (4,T), ``dts`` (T-1,), ``sog_rate`` (T,), ``cog_rate`` (T,) like a
    reference ``ShipTrack``, ship_track.py:70-83) with their per-track ``dt`` arrays and priors into a HostBatch.
    Ragged batches are padded to the longest track.  ``P0`` is one 4x4 shared matrix or a sequence of B matrices.
    ``noise`` (test-only) is a per-track list of dicts with ``noise_pred`` (N,4), ``noise_upd`` (N+1,4), ``noise_rts`` (N,4).
    ``on_error``: a track whose update index runs past its last observation (duplicate timestamps make the
    float-equality trigger fire twice per gap) raises IndexError in the reference (kalman_filter.py:105).  "raise" does
    the same; "flag" (default, the batch example's try/except/continue) truncates that track at the offending step and
    sets STATUS_HOST_INDEX in ``host_status``.
    ``bucket_by_length``: tracks are laid out longest first, so that the 16 or 64 tracks sharing a wave have similar
    step counts and the wave does not idle through the tail of one long track (real data: 30 .. 9 619 observations per
    ship).  ``HostBatch.order`` records the permutation; ``run_batch`` returns results in the caller's order.
    """
    B = len(tracks)
    if B == 0:
        raise ValueError("empty batch")
    order = None
    if bucket_by_length and B > 1:
        lens = np.array([len(d) for d in dts_per_track])
        if len(set(lens.tolist())) > 1:
            order = np.argsort(-lens, kind="stable")
            tracks = [tracks[i] for i in order]
            dts_per_track = [dts_per_track[i] for i in order]
            x0s = [x0s[i] for i in order]
            t0s = None if t0s is None else [t0s[i] for i in order]
            noise = None if noise is None else [noise[i] for i in order]
            P0a = np.asarray(P0, dtype=np.float64)
            if P0a.ndim == 3:
                P0 = P0a[order]
    H, Q, R = _as44(H, "H"), _as44(Q, "Q", True), _as44(R, "R", True)
    Ns = [len(d) for d in dts_per_track]
    Ts = [np.asarray(tr.z).shape[1] for tr in tracks]
    Nmax, Tmax = max(Ns), max(Ts)
    nsteps = np.asarray(Ns, dtype=np.int32)
    host_status = np.zeros(B, dtype=np.int32)
    x0 = np.zeros((4, B))
    dt = np.zeros((Nmax, B))
    sr = np.zeros((Nmax, B))
    cr = np.zeros((Nmax, B))
    srr = np.zeros((Nmax, B))
    crr = np.zeros((Nmax, B))
    ui = np.full((Nmax, B), -1, dtype=np.int32)
    z = np.zeros((Tmax, 4, B))
    P0 = require_symmetric(P0, "P0")
    if P0.shape == (4, 4):
        P0p = np.ascontiguousarray(P0.reshape(16))
    elif P0.shape == (B, 4, 4):
        P0p = np.ascontiguousarray(P0.reshape(B, 16).T)
    else:
        raise ValueError(f"P0 must be (4,4) or (B,4,4), got {P0.shape}")
    have_noise = noise is not None
    npred = np.zeros((Nmax, 4, B)) if have_noise else None
    nupd = np.zeros((Nmax + 1, 4, B)) if have_noise else None
    nrts = np.zeros((Nmax, 4, B)) if have_noise else None
    for b, tr in enumerate(tracks):
        zb = np.asarray(tr.z, dtype=np.float64)
        if zb.shape[0] != 4:
            raise ValueError("measurement matrix z must have 4 rows (lon, lat, sog, cog): call "
                             "get_measurements(include_sog=True, include_cog=True)")
        T, N = Ts[b], Ns[b]
        d = np.asarray(dts_per_track[b], dtype=np.float64)
        dts = np.asarray(tr.dts, dtype=np.float64)
        sog_rate = np.asarray(tr.sog_rate, dtype=np.float64)
        cog_rate = np.asarray(tr.cog_rate, dtype=np.float64)
        u, ridx, _ = update_schedule(d, dts, 0 if t0s is None else t0s[b])
        if N and (u.max() >= T or ridx.max() >= len(sog_rate)):
            if on_error == "raise":
                raise IndexError("update index runs past the last observation (kalman_filter.py:105)")
            bad = np.flatnonzero((u >= T) | (ridx >= len(sog_rate)))[0]
            N = int(bad)
            Ns[b] = N
            nsteps[b] = N
            host_status[b] |= STATUS_HOST_INDEX
            d, u, ridx = d[:N], u[:N], ridx[:N]
        x0[:, b] = np.asarray(x0s[b], dtype=np.float64).reshape(-1)
        z[:T, :, b] = zb.T
        dt[:N, b] = d
        ui[:N, b] = u
        sr[:N, b] = sog_rate[ridx]
        cr[:N, b] = cog_rate[ridx]
        if N and len(dts) and not host_status[b]:
            try:
                rr = rts_rate_index(N + 1, len(dts), len(sog_rate))
                srr[:N, b] = sog_rate[rr]
                crr[:N, b] = cog_rate[rr]
            except IndexError:
                # only the smoother needs these; surface the reference's IndexError when it is called
                srr[:N, b] = np.nan
                crr[:N, b] = np.nan
        if have_noise:
            nb = noise[b]
            npred[:N, :, b] = np.asarray(nb["noise_pred"])[:N]
            nupd[: N + 1, :, b] = np.asarray(nb["noise_upd"])[: N + 1]
            nrts[:N, :, b] = np.asarray(nb["noise_rts"])[:N]
    same_rts = np.array_equal(sr, srr) and np.array_equal(cr, crr)
    return HostBatch(B=B, Nmax=Nmax, Tmax=Tmax, H=H, Q=Q, R=R, nsteps=nsteps, x0=x0, P0=P0p, dt=dt, sog_rate=sr,
                     cog_rate=cr, sog_rate_rts=None if same_rts else srr, cog_rate_rts=None if same_rts else crr,
                     upd_idx=ui, z=z, noise_pred=npred, noise_upd=nupd, noise_rts=nrts, host_status=host_status,
                     order=order)


def pack_uniform(sb, substeps: int, H, Q, R, P0) -> HostBatch:
    """
    Fast path for a batch where every track has the same number of observations (``synthetic.SyntheticBatch``):
    vectorised over tracks.  x0 = z[:, 0] (example_ukf_rts_smoother_batch.py:60); dt = generate_dts(dts, substeps).
    """
    H, Q, R = _as44(H, "H"), _as44(Q, "Q", True), _as44(R, "R", True)
    B, T = sb.lon.shape
    s = int(substeps)
    N = s * (T - 1)
    dt = np.repeat(sb.dts / s, s, axis=1)  # (B, N): utils.py:194-198
    cums = np.cumsum(sb.dts, axis=1)
    times = np.cumsum(dt, axis=1)  # sequential per row, starting from 0 + dt[0] == dt[0]
    fires = np.empty((B, N), dtype=bool)
    for b in range(B):
        fires[b] = _isin_exact(times[b], cums[b])
    after = np.cumsum(fires, axis=1, dtype=np.int32)
    upd_idx = np.where(fires, after, np.int32(-1))
    ridx = after - fires
    if upd_idx.max() >= T:
        raise IndexError("update index runs past the last observation (kalman_filter.py:105)")
    # everything below is produced directly in the device layout [step][track] (gathers with transposed index views)
    c = np.ascontiguousarray
    cols = np.arange(B)[None, :]
    sog_t, cog_t = c(sb.sog_rate.T), c(sb.cog_rate.T)  # (T, B)
    ridx_t = ridx.T
    sr = sog_t[ridx_t, cols]
    cr = cog_t[ridx_t, cols]
    rr = rts_rate_index(N + 1, T - 1, T)
    same_rts = bool(np.array_equal(ridx, np.broadcast_to(rr, (B, N))))
    if not same_rts:
        srr, crr = sog_t[rr], cog_t[rr]
        same_rts = np.array_equal(sr, srr) and np.array_equal(cr, crr)
    P0 = require_symmetric(_as44(P0, "P0"), "P0")
    return HostBatch(
        B=B, Nmax=N, Tmax=T, H=H, Q=Q, R=R, nsteps=np.full(B, N, dtype=np.int32), x0=c(sb.z[:, :, 0].T),
        P0=c(P0.reshape(16)), dt=np.repeat(c(sb.dts.T) / s, s, axis=0), sog_rate=sr, cog_rate=cr,
        sog_rate_rts=None if same_rts else srr, cog_rate_rts=None if same_rts else crr,
        upd_idx=c(upd_idx.T), z=c(sb.z.transpose(2, 1, 0)),
    )


# ----------------------------------------------------------------------------------------------------------------
# device side
# ----------------------------------------------------------------------------------------------------------------
class DeviceBatch:
    """A HostBatch resident in HBM plus its output buffers; ``forward`` / ``backward`` / ``run`` launch the kernels."""

    _IN = ("nsteps", "x0", "P0", "dt", "sog_rate", "cog_rate", "sog_rate_rts", "cog_rate_rts", "upd_idx", "z",
           "noise_pred", "noise_upd", "noise_rts")

    # position of (r, c) in a packed upper triangle, for all 16 entries of the full matrix (include/ste.h: STE_FLAG_PACKED_COV)
    _PACKED_INDEX = [min(r, c) * 4 - (min(r, c) * (min(r, c) - 1)) // 2 + abs(r - c) for r in range(4) for c in range(4)]

    def __init__(self, hb: HostBatch, device="cuda:0", alloc_smoothed: bool = True, fuse_gains: bool = True,
                 tuning: int = 0, packed_cov: bool = True, sm_pos: bool = False, upload: bool = True):
        """``sm_pos``: also allocate the smoother's optional [N+1][2][B] output of smoothed lon / lat (what a multi-GPU
        run exchanges).  ``upload=False``: allocate the input tensors without filling them -- ``upload_tracks(lo, hi)``
        then brings the host batch up window by window (``run_fleet``)."""
        import torch

        self.lib = binding.require_gpu()
        self.torch = torch
        self.hb = hb
        self.device = torch.device(device)
        self.t = {}
        for name in self._IN:
            a = getattr(hb, name)
            if a is None:
                self.t[name] = None
            elif upload:
                self.t[name] = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            else:
                self.t[name] = torch.empty(a.shape, dtype=torch.from_numpy(a[..., :0]).dtype, device=self.device)
        B, N = hb.B, hb.Nmax
        self.ntracks, self.lo, self.parent = B, 0, None
        f64 = dict(dtype=torch.float64, device=self.device)
        self.fwd_mean = torch.empty((N + 1, 4, B), **f64)
        # covariance histories as upper triangles on the device (they are symmetric by construction); download()
        # expands them to the reference's (.., 4, 4).  packed_cov=False keeps full matrices in HBM.
        self.packed_cov = bool(packed_cov)
        cov_rows = 10 if self.packed_cov else 16
        self.fwd_cov = torch.empty((N + 1, cov_rows, B), **f64)
        self.sm_mean = torch.empty((N + 1, 4, B), **f64) if alloc_smoothed else None
        self.sm_cov = torch.empty((N + 1, cov_rows, B), **f64) if alloc_smoothed else None
        # rows past a short track's end are never written: zeros, so that a gathered tensor is defined everywhere
        self.sm_pos = torch.zeros((N + 1, 2, B), **f64) if (alloc_smoothed and sm_pos) else None
        self.status = torch.zeros((B,), dtype=torch.int32, device=self.device)
        # workspace for the smoother gains the forward pass can produce on the way (include/ste.h: rts_work)
        self.rts_work = None
        if alloc_smoothed and fuse_gains and N > 0:
            self.rts_work = torch.empty((N * binding.STE_RTS_WORK_ROWS + 1, B), **f64)  # include/ste.h: rts_work
        fan_scale, w0, wi = sigma_constants(4, hb.weights_computed)
        self._keep = (hb.H, hb.Q, hb.R)
        s = binding.SteUkfBatchF64()
        s.B, s.Nmax, s.Tmax, s.n = B, N, hb.Tmax, 4
        lanes = default_lanes if hb.lanes is None else hb.lanes
        if lanes not in (0, 1, 4):
            raise ValueError(f"lanes must be 0 (automatic), 1 or 4, got {lanes!r}")
        s.flags = (binding.STE_FLAG_SHARED_P0 if hb.shared_p0 else 0) | (
            0 if hb.initial_update else binding.STE_FLAG_NO_INITIAL_UPDATE) | (
            binding.STE_FLAG_ROBUST if hb.robust else 0) | {0: 0, 1: binding.STE_FLAG_LANES_1, 4: binding.STE_FLAG_LANES_4}[lanes] | (
            binding.STE_FLAG_PACKED_COV if self.packed_cov else 0)
        s.tuning = int(tuning) if tuning else int(default_tuning)
        s.chi_alpha, s.robust_max_iter = float(hb.chi_alpha), 50
        s.fan_scale, s.w0, s.wi = fan_scale, w0, wi
        s.H, s.Q, s.R = hb.H.ctypes.data, hb.Q.ctypes.data, hb.R.ctypes.data
        for name in self._IN:
            ten = self.t[name]
            setattr(s, name, None if ten is None else ten.data_ptr())
        s.fwd_mean, s.fwd_cov = self.fwd_mean.data_ptr(), self.fwd_cov.data_ptr()
        s.sm_mean = None if self.sm_mean is None else self.sm_mean.data_ptr()
        s.sm_cov = None if self.sm_cov is None else self.sm_cov.data_ptr()
        s.status = self.status.data_ptr()
        s.rts_work = None if self.rts_work is None else self.rts_work.data_ptr()
        s.track_stride = 0
        s.sm_pos = None if self.sm_pos is None else self.sm_pos.data_ptr()
        s.step_begin = s.step_end = 0
        self.struct = s
        # The uploads above were queued on the current stream.  One event, recorded now, is what other streams wait on
        # before the first launch (SmootherPipeline.submit): waiting on the current stream *at submit time* would put a
        # marker on the legacy default stream, which every blocking stream synchronises with -- a device-wide barrier in
        # the middle of a pipelined sequence (two of them cost a --steps 20 --warmup 5 run 10 of its 32 ms in round 2).
        self._uploaded = torch.cuda.Event()
        self._uploaded.record(torch.cuda.current_stream(self.device))
        self._pipeline_done = None
        self._last_use = None  # event after the last forward / backward / run issued outside a pipeline

    # -- windows of a resident batch ----------------------------------------------------------------------------
    _PER_TRACK_F64 = ("x0", "dt", "sog_rate", "cog_rate", "sog_rate_rts", "cog_rate_rts", "z", "noise_pred",
                      "noise_upd", "noise_rts", "fwd_mean", "fwd_cov", "sm_mean", "sm_cov", "rts_work", "sm_pos")
    _PER_TRACK_I32 = ("nsteps", "upd_idx", "status")

    def window(self, lo: int, hi: int) -> "DeviceBatch":
        """Tracks [lo, hi) of this resident batch as a batch of their own: same tensors, a batch struct whose pointers
        name track ``lo`` and whose ``track_stride`` is this batch's width (include/ste.h) -- nothing is copied.  The
        windows of a fleet go through ``SmootherPipeline`` like separate batches and write straight into the fleet's
        histories (``run_fleet``)."""
        if not (0 <= lo < hi <= self.ntracks):
            raise ValueError(f"window [{lo}, {hi}) outside the batch's {self.ntracks} tracks")
        w = object.__new__(DeviceBatch)
        w.__dict__.update(self.__dict__)
        w.parent = self if self.parent is None else self.parent
        w.lo, w.ntracks = self.lo + lo, hi - lo
        s = binding.SteUkfBatchF64.from_buffer_copy(self.struct)
        s.B = hi - lo
        s.track_stride = self.struct.track_stride or self.struct.B
        for name in self._PER_TRACK_F64 + self._PER_TRACK_I32:
            ptr = getattr(s, name)
            if ptr:
                setattr(s, name, ptr + lo * (4 if name in self._PER_TRACK_I32 else 8))
        if not self.hb.shared_p0:
            s.P0 = s.P0 + lo * 8
        w.struct = s
        for name in ("fwd_mean", "fwd_cov", "sm_mean", "sm_cov", "sm_pos"):
            t = getattr(self, name)
            setattr(w, name, None if t is None else t[..., lo:hi])
        w.status = self.status[lo:hi]
        w._pipeline_done = None
        w._last_use = None
        return w

    def upload_tracks(self, lo: int, hi: int, stream=None):
        """Bring tracks [lo, hi) of the host batch into the (``upload=False``) input tensors: page-locked staging, filled by
        a few threads (the strided slice copies release the GIL; one thread moves ~12 GB/s, and direct strided DMA from
        page-locked NumPy memory was measured far slower: 0.3 GB/s), then asynchronous copies on ``stream``; returns the event
        that marks them resident."""
        torch = self.torch
        stream = stream or torch.cuda.current_stream(self.device)
        jobs = []
        for name in self._IN:
            ten = self.t[name]
            if ten is None or (name == "P0" and self.hb.shared_p0):
                continue
            src = getattr(self.hb, name)[..., lo:hi]
            stage = torch.empty(src.shape, dtype=ten.dtype, pin_memory=True)
            jobs.append((ten, stage, src))

        def fill(job):
            job[1].numpy()[...] = job[2]

        if sum(j[2].nbytes for j in jobs) > (64 << 20):
            list(_staging_pool().map(fill, jobs))
        else:
            for j in jobs:
                fill(j)
        with torch.cuda.stream(stream):
            for ten, stage, _ in jobs:
                ten[..., lo:hi].copy_(stage, non_blocking=True)
            if self.hb.shared_p0 and lo == 0:
                self.t["P0"].copy_(torch.from_numpy(np.ascontiguousarray(self.hb.P0)))
            ev = torch.cuda.Event()
            ev.record(stream)
        return ev

    def _stream(self, stream):
        if stream is None:
            stream = self.torch.cuda.current_stream(self.device)
        return C.c_void_p(stream.cuda_stream)

    def _mark_use(self, stream):
        # kernels issued outside a pipeline: a later SmootherPipeline.submit of this batch orders itself behind them
        if stream is None:
            stream = self.torch.cuda.current_stream(self.device)
        if not stream.cuda_stream:
            return  # the legacy default stream orders itself against every blocking stream (the pipeline's included)
        self._last_use = self.torch.cuda.Event()
        self._last_use.record(stream)

    @staticmethod
    def slice_bounds(nsteps: int, slices: int):
        """Step ranges of a forward pass cut into ``slices`` time slices: boundaries on multiples of STE_SLICE_ALIGN (64
        steps), where the eigen-solve's warm start restarts anyway, so the slices reproduce the whole pass bit for bit."""
        a = binding.STE_SLICE_ALIGN
        if slices <= 1 or nsteps <= a:
            return [(0, nsteps)]
        step = -(-nsteps // (slices * a)) * a
        cuts = list(range(0, nsteps, step)) + [nsteps]
        return list(zip(cuts[:-1], cuts[1:]))

    def forward(self, stream=None, slices: int = 1, mark: bool = True):
        """The forward pass; ``slices`` > 1 issues it as that many launches over consecutive step ranges (include/ste.h:
        step_begin / step_end), bit-identical to the single launch."""
        s = self.struct
        try:
            for k0, k1 in self.slice_bounds(int(s.Nmax), int(slices)):
                s.step_begin, s.step_end = k0, k1
                binding.check(self.lib.ste_ukf_forward_f64(C.byref(s), self._stream(stream)), "ste_ukf_forward_f64")
        finally:
            s.step_begin = s.step_end = 0
        if mark:
            self._mark_use(stream)

    def backward(self, stream=None, mark: bool = True):
        binding.check(self.lib.ste_urtss_backward_f64(C.byref(self.struct), self._stream(stream)),
                      "ste_urtss_backward_f64")
        if mark:
            self._mark_use(stream)

    def run(self, stream=None):
        binding.check(self.lib.ste_ukf_urtss_f64(C.byref(self.struct), self._stream(stream)), "ste_ukf_urtss_f64")
        self._mark_use(stream)

    # -- results ------------------------------------------------------------------------------------------------
    _OUT = {"means": ("fwd_mean", 4), "covs": ("fwd_cov", 16), "means_smoothed": ("sm_mean", 4),
            "covs_smoothed": ("sm_cov", 16)}

    def download(self, names=("means", "covs", "means_smoothed", "covs_smoothed"), track_index=None):
        """Histories as NumPy arrays shaped like the reference's return values, track first: ``means*`` (B, Nmax+1, 4),
        ``covs*`` (B, Nmax+1, 4, 4); rows past nsteps[b] are padding.  ``track_index`` (device int64 tensor) reorders /
        selects tracks on the device.  Each tensor is transposed on the device (one pass at HBM speed) and copied into
        page-locked host memory with an asynchronous copy; the arrays returned are views of that memory (PyTorch's
        host allocator recycles it once they are dropped, so only a process's first large download pays for pinning).
        10 000 x 500 with all four histories: 1.6 GB, ~35 ms over PCIe gen 5 instead of ~170 ms through pageable memory."""
        torch = self.torch
        if self._pipeline_done is not None:  # results of a SmootherPipeline.submit still in flight on another stream
            torch.cuda.current_stream(self.device).wait_event(self._pipeline_done)
        out, pending = {}, []
        for name in names:
            attr, width = self._OUT[name]
            t = getattr(self, attr)
            if t is None:
                raise ValueError(f"{name} was not computed for this batch (alloc_smoothed=False)")
            if track_index is not None:
                t = t.index_select(2, track_index)
            if width == 16 and self.packed_cov:  # upper triangles -> full symmetric matrices, still on the device
                if getattr(self, "_packed_index_t", None) is None:
                    self._packed_index_t = torch.tensor(self._PACKED_INDEX, dtype=torch.int64, device=self.device)
                t = t.index_select(1, self._packed_index_t)
            dev_t = t.permute(2, 0, 1).contiguous()  # (B, N+1, width) on the device
            host = torch.empty(dev_t.shape, dtype=dev_t.dtype, pin_memory=True)
            host.copy_(dev_t, non_blocking=True)
            pending.append(dev_t)  # keep the source alive until the copy has run
            a = host.numpy()
            out[name] = a.reshape(a.shape[0], a.shape[1], 4, 4) if width == 16 else a
        torch.cuda.current_stream(self.device).synchronize()
        return out

    def _download_into(self, names, host, lo, hi):
        """Asynchronous part of ``download`` for one window of a fleet: each history, track-major, into rows [lo, hi) of the
        page-locked tensors ``host[name]`` on the current stream.  Returns the device temporaries (to be kept alive until
        the stream has been synchronised)."""
        torch = self.torch
        pending = []
        for name in names:
            attr, width = self._OUT[name]
            t = getattr(self, attr)
            if width == 16 and self.packed_cov:
                if getattr(self, "_packed_index_t", None) is None:
                    self._packed_index_t = torch.tensor(self._PACKED_INDEX, dtype=torch.int64, device=self.device)
                t = t.index_select(1, self._packed_index_t)
            dev_t = t.permute(2, 0, 1).contiguous()
            host[name][lo:hi].copy_(dev_t, non_blocking=True)
            pending.append(dev_t)
        return pending

    def filtered(self):
        """(means (B, Nmax+1, 4), covs (B, Nmax+1, 4, 4)) as NumPy arrays (rows past nsteps[b] are padding)."""
        d = self.download(("means", "covs"))
        return d["means"], d["covs"]

    def smoothed(self):
        d = self.download(("means_smoothed", "covs_smoothed"))
        return d["means_smoothed"], d["covs_smoothed"]

    def status_host(self):
        if self._pipeline_done is not None:
            self._pipeline_done.synchronize()
        return self.status.cpu().numpy()


def forward_schedule(ntiles: Sequence[int], nslices: Sequence[int], nwaves: int, stagger: float = 0.0) -> np.ndarray:
    """
    The item table of a scheduled forward launch (include/ste.h: ``ste_fwd_sched_f64.items``): which (window, tile) every
    one of ``nwaves`` resident waves runs in every round, a round being one time slice of one 64-track tile.

    Window w has ``ntiles[w]`` tiles of ``nslices[w]`` slices each; a tile's slices run in order, one per round at most (a
    slice starts from the history row the one before it left).  With whole forward waves per tile -- one launch per window --
    W windows of T tiles cost ceil(W T / nwaves) pass times; here the makespan is R = ceil(total slices / nwaves) rounds
    (at least the longest tile): McNaughton's bound for preemptive scheduling of chains of unit jobs.  It is reached by list
    scheduling with two rules: a tile whose remaining slices equal the rounds left until its deadline runs now ("critical"),
    and the other waves go to the unfinished tiles in window order.

    ``stagger``: 0 gives every window the deadline R -- windows then finish in generations of as many as fill the chip,
    like whole launches do, only the last generation is spread under the ones before it.  1 spreads the deadlines of the
    windows evenly from the first possible finish to R, so that smoothers find forward passes to hide behind from the
    first finished window to the last.  Values between interpolate.

    Returns int32 [R][nwaves][2]; entries (-1, 0) are idle waves.  A tile stays on its wave from round to round when it
    keeps running.
    """
    ntiles = np.asarray(ntiles, dtype=np.int64)
    nslices = np.asarray(nslices, dtype=np.int64)
    if ntiles.ndim != 1 or ntiles.shape != nslices.shape or len(ntiles) == 0 or (ntiles < 1).any() or (nslices < 1).any():
        raise ValueError("ntiles and nslices: one positive entry per window")
    nwaves = int(nwaves)
    if nwaves < 1:
        raise ValueError("nwaves must be >= 1")
    nwin = len(ntiles)
    win = np.repeat(np.arange(nwin), ntiles)            # window of every job (tile), window order
    tile = np.concatenate([np.arange(n) for n in ntiles])
    remaining = nslices[win].copy()
    njobs = len(win)
    total = int(remaining.sum())
    R = max(-(-total // nwaves), int(nslices.max()))
    # deadline of window w: by then all work of windows <= w must fit on the chip, and no tile ends before its own length
    cum = np.cumsum(ntiles * nslices)
    earliest = np.maximum(np.ceil(cum / nwaves), np.maximum.accumulate(nslices)).astype(np.int64)
    first, span = int(earliest[0]), R - int(earliest[0])
    even = first + np.ceil(span * (np.arange(nwin) + 1 - 1) / max(nwin - 1, 1)).astype(np.int64) if nwin > 1 else np.array([R])
    dl_w = np.minimum(R, np.maximum(earliest, np.round(stagger * even + (1.0 - stagger) * R).astype(np.int64)))
    dl_w = np.maximum.accumulate(dl_w)
    dl = dl_w[win]
    items = np.full((R, nwaves, 2), (-1, 0), dtype=np.int32)
    wave_of = np.full(njobs, -1, dtype=np.int64)  # wave a job ran on in the previous round, -1: it did not run
    order = np.arange(njobs)
    r = 0
    while remaining.any():
        if r >= len(items):  # deadlines were too tight for the list scheduler somewhere: one more round
            items = np.concatenate([items, np.full((1, nwaves, 2), (-1, 0), dtype=np.int32)])
        live = order[remaining > 0]
        crit = (dl[live] - r) <= remaining[live]
        # critical tiles first, then earliest deadline, then window / tile order (np.lexsort: last key is the primary one)
        pick = live[np.lexsort((live, dl[live], ~crit))][:nwaves]
        keep = pick[wave_of[pick] >= 0]
        used = np.zeros(nwaves, dtype=bool)
        used[wave_of[keep]] = True
        new = pick[wave_of[pick] < 0]
        free = np.flatnonzero(~used)[: len(new)]
        ran = np.full(njobs, -1, dtype=np.int64)
        ran[keep] = wave_of[keep]
        ran[new] = free
        items[r, ran[pick], 0] = win[pick]
        items[r, ran[pick], 1] = tile[pick]
        remaining[pick] -= 1
        ran[remaining == 0] = -1
        wave_of = ran
        r += 1
    return items[:r]


# the most recent scheduled forward launch per device, of any pipeline of this process: (its counters, index of its
# started-waves word, its wave count, the event behind its kernel) -- SmootherPipeline.submit_sequence
_last_scheduled = {}


class SmootherPipeline:
    """
    Forward passes and smoothers of consecutive batches side by side on the GPU, several of each in flight.

    A batch of BASELINE size (10 000 tracks) is 157 long-running forward waves on 1 024 SIMDs followed by a smoother that
    is a latency chain (one wave per 64 tracks, ~1.5 us per step); run back to back they leave most of the chip idle.
    Two ways to overlap consecutive batches, both on streams that own a hardware queue each
    (``ste_stream_create_cu_range``; ordinary HIP streams share a handful of queues and their launches serialise):

    ``shared=True`` (default)  every stream may use every compute unit.  A lane-per-track forward wave is built to hold
        264 registers -- one per SIMD, never two -- and a smoother wave 240, so the smoothers of earlier batches slot in
        beside the forward waves of later ones and take the issue slots those leave (a forward wave issues ~80 % of its
        cycles; a smoother wave mostly waits for memory).  ``forward_streams`` forward passes fill the chip's SIMDs
        (seven at 10 000 tracks), ``smoother_streams`` smoothers hide each other's latency (six).
    ``shared=False``  round 1-2's split: forward passes on the first ``forward_cus`` compute units, smoothers on the rest.

    ``sequence_only=True``: a pipeline that will only see ``submit_sequence`` (one scheduled forward launch per sequence)
    keeps two forward streams.  Every stream here is a hardware queue, the device has about two dozen for everything that
    runs on it (measured: this pipeline's 14 plus nine idle ones elsewhere in the process and launches start to take turns),
    and scheduled launches -- long-lived kernels beside one-wave gates that never leave their queues idle -- are the first
    to suffer when they run out.

    Each ``DeviceBatch`` owns its histories and work rows; a batch is not resubmitted before its previous smoother has
    finished (events), so the caller rotates through ``buffers_needed`` or more of them.

        with SmootherPipeline(device, ntracks=hb.B) as pipe:
            dbs = [DeviceBatch(hb, device) for _ in range(pipe.buffers_needed)]
            for k in range(nbatches):
                pipe.submit(dbs[k % len(dbs)], final=(k == nbatches - 1))
            pipe.synchronize()

    The CU-masked streams are destroyed by ``close()`` (the context manager, ``__del__``, and at the latest an atexit
    hook): left to the HIP runtime's static destructors they outlive any profiler tool.
    """

    def __init__(self, device="cuda:0", forward_cus: Optional[int] = None, ntracks: Optional[int] = None,
                 forward_streams: Optional[int] = None, smoother_streams: Optional[int] = None, forward_lanes: int = 1,
                 shared: Optional[bool] = None, reserve_cus: int = 0, slices: Optional[int] = None, sequence_only: bool = False):
        import torch

        self.torch = torch
        self.lib = binding.require_gpu()
        self.device = torch.device(device)
        if forward_lanes not in (0, 1, 4):
            raise ValueError(f"forward_lanes must be 0 (the library's choice by batch size), 1 or 4, got {forward_lanes!r}")
        self.forward_lanes = int(forward_lanes)
        ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
        quad = forward_lanes == 4 or (forward_lanes == 0 and (ntracks or 10_000) <= 32_768)
        if shared is None:  # naming a forward partition asks for the split; otherwise everything shares the chip
            shared = forward_cus is None
        self.shared = bool(shared)
        if shared:
            # no partition: forward passes and smoothers on streams that each own a hardware queue but may use every CU.
            # A lane-per-track forward wave holds 264 registers (one per SIMD by construction), a smoother wave 240, so
            # the smoother of one batch slots in beside the forward waves of the next ones and takes the issue slots they
            # leave (its waves mostly wait for memory).
            forward_cus = ncu
        if forward_cus is None:
            # shared=False, the round 1-2 split (kept for measurements): the fraction of the chip the forward passes get was
            # sized then, against a workgroup smoother that no longer exists -- five eighths for lane-per-track passes
            # (160 + 96 CUs), three quarters for quad-per-track ones (192 + 64).  With today's kernels the same split
            # measured 0.86 ms per step at 10 000 tracks against 0.68 shared; it is not re-tuned.
            forward_cus = (ncu * (3 if quad else 5) // (4 if quad else 8)) // 8 * 8
            # small devices / partitioned compute modes: keep at least one CU on either side of the split
            forward_cus = max(1, min(forward_cus if forward_cus > 0 else ncu // 2, ncu - 1))
        if forward_streams is None:
            # as many forward passes in flight as fill the partition's wave slots: a lane-per-track wave holds a SIMD's
            # whole register file, quad-per-track waves (256 VGPRs) fit two to a SIMD.  Quad passes: rounded up (the waves
            # of the last pass start as slots come free: 3 on 192 CUs beat 2).  Lane-per-track passes: rounded to nearest
            # (10 000 tracks on 160 CUs: 4.08 -> 4, 1.02 ms per step with 4 or 5; 12 500 tracks: 3.27 -> 3, 1.53 ms
            # against 1.68 with 4).
            nt = ntracks or 10_000
            waves = -(-nt * 4 // 64) if quad else -(-nt // 64)
            slots = forward_cus * (8 if quad else 4)
            if shared:  # fill the SIMDs (rounded up: the waves of the last pass start as slots come free), at most eight
                forward_streams = max(1, min(8, -(-slots // waves)))
            else:
                forward_streams = max(1, min(3, -(-slots // waves)) if quad else min(8, (2 * slots + waves) // (2 * waves)))
        if smoother_streams is None:
            # shared: measured at 10 000 and 12 500 tracks (profiles/r03_pipeline_sweeps.txt): six smoothers in flight keep
            # up with the forward passes (five were enough until the forward kernel lost its last lane reads: 6.96 against
            # 7.43e9 track-steps/s at 10 000 tracks); seven or eight change nothing.  A larger batch fills the chip with
            # fewer forward passes and needs as few smoothers beside them (and each buffer set is the larger for it:
            # 224 + 240 B per track-step, 23 GB at 100 000 x 500): never more smoother streams than forward streams, at least two.
            smoother_streams = max(2, min(6, forward_streams)) if shared else 2
        if sequence_only:
            # a pipeline for submit_sequence alone: a scheduled launch is one kernel on one stream however many windows it
            # covers (two streams: the next sequence's launch starts as the waves of this one leave), and every stream not
            # created is a hardware queue left to whatever else runs on the device
            forward_streams = min(forward_streams, 2)
        if not (0 < forward_cus < ncu) and not shared:
            raise ValueError(f"forward_cus must be in 1..{ncu - 1} (got {forward_cus}): the smoother needs CUs of its own")
        if forward_streams < 1 or smoother_streams < 1:
            raise ValueError("forward_streams and smoother_streams must be >= 1")
        self.forward_cus, self.smoother_cus = int(forward_cus), int(ncu if shared else ncu - forward_cus)
        self._raw = []
        self.fwd_streams, self.bwd_streams = [], []
        # unrestricted stream for the smoother of the last batch of a sequence; created here, not at first use, so
        # that a timed sequence never pays for it
        self._tail_stream = torch.cuda.Stream(self.device)
        self._count = 0
        self._batches = []  # weak references to the DeviceBatches that carry one of this pipeline's events
        self._schedules = {}  # item tables of scheduled forward launches, by shape (submit_sequence)
        self._sched_live = []  # workspaces / counters of scheduled launches not yet synchronised
        self._sched_free = []  # ... and of retired ones, kept for the next launch: no allocator call on the launch path
        self._sched_pinned = True  # page-locked host workspaces: the table is uploaded by a kernel (False: staged copy; tests)
        self._sched_free_bwd = []  # the same for the tables of the sequences' smoother launches
        self._sched_tile_smoothers = True  # one smoother launch per sequence, a wave per tile (False: a launch per window; tests)
        self.buffers_needed = forward_streams + smoother_streams + 1
        # time slices per forward pass (DeviceBatch.forward): the waves of the passes in flight re-balance over the SIMDs at
        # every slice boundary instead of once per pass (3.6-4.5 ms at 500 steps) -- what a short sequence of batches, or
        # a fleet of a few windows, loses at its ends
        self.slices = DEFAULT_SLICES if slices is None else int(slices)
        if self.slices < 1:
            raise ValueError("slices must be >= 1")
        # shared mode only: the last ``reserve_cus`` compute units (spread over the XCDs: mask bit n is CU n / 8 of XCD n % 8)
        # stay free of this pipeline's kernels -- room that a collective's own kernels can always find (multi-GPU runs)
        self.reserve_cus = int(reserve_cus) if shared else 0
        if not 0 <= self.reserve_cus < ncu:
            raise ValueError(f"reserve_cus must be in 0..{ncu - 1}, got {reserve_cus!r}")
        try:
            with torch.cuda.device(self.device):
                for first, count, n, out in ((0, self.forward_cus - self.reserve_cus, forward_streams, self.fwd_streams),
                                             (0 if shared else self.forward_cus, self.smoother_cus - self.reserve_cus,
                                              smoother_streams, self.bwd_streams)):
                    for _ in range(n):
                        h = C.c_void_p()
                        binding.check(self.lib.ste_stream_create_cu_range(first, count, C.byref(h)),
                                      "ste_stream_create_cu_range")
                        self._raw.append(h)
                        out.append(torch.cuda.ExternalStream(h.value, device=self.device))
        except BaseException:
            self.close()  # do not leak the streams created before the failing one
            raise
        # CU-masked streams own HSA queues of their own.  Left alive until the process exits they are torn down by the
        # HIP runtime's static destructors, after Python, torch and any profiler tool have finalised -- under rocprofv3
        # that order ended every pipelined run of round 1 in a SIGSEGV inside __cxa_finalize.  So the streams are
        # destroyed while everything is still up: close() / the context manager / __del__, and at the latest atexit.
        _live_pipelines.add(self)
        _register_atexit()
        # Every stream of a pipeline owns a hardware queue.  Past ~16 of them in a process the firmware multiplexes queues and
        # launches crawl (a 100 000-track fleet: 8.9 ms on 10 + 6 streams, 12.0 on 13 + 8, 28 on 16 + 8; profiles/r04_fleet_sweep.txt).
        nq = sum(len(q.fwd_streams) + len(q.bwd_streams) for q in _live_pipelines)
        if nq > MAX_PIPELINE_QUEUES:
            import warnings

            warnings.warn(f"{nq} pipeline streams are alive in this process (more than {MAX_PIPELINE_QUEUES}): each owns a hardware "
                          "queue, and beyond that many the GPU multiplexes them and every launch slows down; close() the "
                          "pipelines that are no longer used", RuntimeWarning, stacklevel=2)

    # single-stream names kept for callers that look at them
    @property
    def fwd_stream(self):
        return self.fwd_streams[0]

    @property
    def bwd_stream(self):
        return self.bwd_streams[0]

    def submit(self, db: "DeviceBatch", after_smoother=None, timing=None, final: bool = False, smooth: bool = True,
               slices: Optional[int] = None):
        """Queue forward + smoother of ``db``; returns the event that marks its smoother (and ``after_smoother``) done.

        ``after_smoother(stream)``: optional callable run with the smoother stream current, right after the smoother
        kernels are queued (the multi-GPU driver starts its all-gather of the smoothed positions there).  It may return
        an event; the next use of ``db``'s buffers then waits for that event too (a collective still reading them).
        ``smooth=False``: forward pass only.  ``slices``: time slices of this batch's forward pass (default: the pipeline's).
        ``timing``: optional list of four timing-enabled events, recorded before / after the forward kernel on its
        forward stream and before / after the smoother kernels on its smoother stream.
        ``final``: nothing follows this batch, so its smoother gets an unrestricted stream (the whole chip) instead of the
        smoother partition."""
        torch = self.torch
        if self.closed:
            raise RuntimeError("SmootherPipeline is closed")
        k = self._count
        self._count += 1
        fwd_stream = self.fwd_streams[k % len(self.fwd_streams)]
        bwd_stream = self.bwd_streams[k % len(self.bwd_streams)]
        if final:
            bwd_stream = self._tail_stream
        done = getattr(db, "_pipeline_done", None)
        if done is not None:
            fwd_stream.wait_event(done)  # the previous use of these buffers has drained
        else:
            # uploads queued by the constructor: an event it recorded then.  (Not wait_stream(current stream): that
            # records on the legacy default stream, which drains every blocking stream -- this pipeline's included.)
            fwd_stream.wait_event(db._uploaded)
        for name in ("_last_use", "_reader_done"):
            # kernels issued on this batch outside the pipeline (forward / backward / run on a stream of the caller's), or a
            # collective started by an earlier after_smoother that may still be reading its outputs
            e = getattr(db, name, None)
            if e is not None:
                fwd_stream.wait_event(e)
                setattr(db, name, None)
        if timing is not None:
            timing[0].record(fwd_stream)
        # Lane mapping of the forward pass: with several passes sharing the partition a lane per track is the better
        # shape even for small batches -- half the instructions per track of the quad mapping, and the other passes'
        # waves fill the SIMDs a 157-wave pass leaves empty.  A mapping named by the batch itself (HostBatch.lanes) wins.
        flags = db.struct.flags
        if self.forward_lanes and not (flags & (binding.STE_FLAG_LANES_1 | binding.STE_FLAG_LANES_4)):
            db.struct.flags = flags | (binding.STE_FLAG_LANES_1 if self.forward_lanes == 1 else binding.STE_FLAG_LANES_4)
        try:
            db.forward(fwd_stream, slices=self.slices if slices is None else int(slices), mark=False)
        finally:
            db.struct.flags = flags
        ready = timing[1] if timing is not None else torch.cuda.Event()
        ready.record(fwd_stream)
        if not smooth:
            if getattr(db, "_pipeline_done", None) is None:
                self._batches.append(weakref.ref(db))
            db._pipeline_done = ready
            return ready
        bwd_stream.wait_event(ready)
        if timing is not None:
            timing[2].record(bwd_stream)
        db.backward(bwd_stream, mark=False)
        if timing is not None:
            timing[3].record(bwd_stream)
        if after_smoother is not None:
            with torch.cuda.stream(bwd_stream):
                db._reader_done = after_smoother(bwd_stream)
        done = torch.cuda.Event()
        done.record(bwd_stream)
        if getattr(db, "_pipeline_done", None) is None:
            self._batches.append(weakref.ref(db))
        db._pipeline_done = done
        return done

    # -- many batches (or windows of a fleet) as ONE scheduled forward launch ---------------------------------------------
    def submit_sequence(self, dbs: Sequence["DeviceBatch"], smooth: bool = True, after_smoother=None, timing=None,
                        final: bool = True, stagger: float = 0.0, slice_steps: int = 0, timeout_s: float = 4.0,
                        tile_smoothers: Optional[bool] = None):
        """Queue forward + smoother of every batch of ``dbs`` -- distinct buffer sets, or the windows of one resident fleet --
        with ALL their forward passes as one launch of resident waves that work through a schedule of (64-track tile, time
        slice) items (``forward_schedule``; include/ste.h: ``ste_ukf_forward_sched_f64``).  One launch per batch makes a
        forward wave indivisible for a whole pass, so W batches of T tiles on S SIMDs cost ceil(W T / S) pass times; scheduled,
        they cost W T / S (rounded up to a slice).  Results are those of ``submit`` on every batch, bit for bit.

        Each batch's smoother goes on a smoother stream behind a one-wave gate that leaves when the batch's last tile has
        finished its last slice.  ``after_smoother(k, stream)``: optional, called for batch k with its smoother stream current,
        may return an event the batch's next use must wait for.  ``timing``: optional dict, gets ``forward`` = (start, end)
        timing events around the scheduled launch and ``smoothers`` = (start, end) around the smoother of every
        ``timing["every"]``-th batch (default 4: every record is a packet on the stream's queue between two launches).
        ``final``: nothing follows this sequence: its last smoother gets the unrestricted stream.
        ``tile_smoothers``: the smoothers of the whole sequence as ONE launch, a wave per 64-track tile that waits for its own
        tile's forward pass (``ste_urtss_backward_sched_f64``) instead of a gate and a launch per batch; needs batches of more
        than 4 096 tracks (the one-kernel smoother) and no ``after_smoother``.  Default: when ``final`` and no scheduled launch
        is in flight in front of this one (a job that is one sequence: a fleet, a short run).  Same bits either way.
        Returns the list of the batches' completion events."""
        torch = self.torch
        if self.closed:
            raise RuntimeError("SmootherPipeline is closed")
        dbs = list(dbs)
        if not dbs:
            return []
        if self.forward_lanes == 4:
            raise ValueError("scheduled forward launches are lane-per-track (this pipeline was built with forward_lanes=4)")
        seen = set()
        for db in dbs:
            key = (db.fwd_mean.data_ptr(), db.ntracks)
            if key in seen:
                raise ValueError("submit_sequence: every batch of a sequence needs histories of its own (a buffer set appears twice)")
            seen.add(key)
        if len(self._sched_live) > 8:
            # a caller that never synchronises the pipeline: retire the launches that have long finished (their tables and
            # counters are only kept for the error word, which is looked at now)
            finished = lambda e: all(ev.query() for ev in e[5] + e[6])  # noqa: E731
            done_ones = [e for e in self._sched_live if finished(e)]
            if done_ones:
                keep = [e for e in self._sched_live if not any(e is d for d in done_ones)]
                self._sched_live = done_ones
                try:
                    self._check_scheduled()
                finally:
                    self._sched_live = keep
        k = self._count
        self._count += 1
        fwd_stream = self.fwd_streams[k % len(self.fwd_streams)]
        for db in dbs:
            done = getattr(db, "_pipeline_done", None)
            fwd_stream.wait_event(done if done is not None else db._uploaded)
            for name in ("_last_use", "_reader_done"):
                e = getattr(db, name, None)
                if e is not None:
                    fwd_stream.wait_event(e)
                    setattr(db, name, None)
        n = len(dbs)
        structs = (binding.SteUkfBatchF64 * n)()
        for i, db in enumerate(dbs):
            st = binding.SteUkfBatchF64.from_buffer_copy(db.struct)
            st.flags = (st.flags & ~binding.STE_FLAG_LANES_4) | binding.STE_FLAG_LANES_1
            st.step_begin = st.step_end = 0
            structs[i] = st
        step = int(slice_steps) or binding.STE_SLICE_ALIGN
        ntiles = [-(-int(st.B) // 64) for st in structs]
        nslices = [max(1, -(-int(st.Nmax) // step)) for st in structs]
        nwaves = 4 * (self.forward_cus - self.reserve_cus)  # one per SIMD this pipeline's forward streams may use
        skey = (tuple(ntiles), tuple(nslices), nwaves, float(stagger))
        items = self._schedules.get(skey)
        if items is None:
            items = np.ascontiguousarray(forward_schedule(ntiles, nslices, nwaves, stagger=stagger))
            if len(self._schedules) > 16:
                self._schedules.clear()
            self._schedules[skey] = items
        nbytes = int(self.lib.ste_ukf_forward_sched_workspace(n, max(nslices), sum(ntiles), items.shape[0], nwaves))
        # workspace and counters come from the launches retired before (synchronize / the pruning above): an allocation here
        # -- pinned host memory above all -- is a driver call of unbounded length in the middle of a sequence of launches
        host_ws = dev_ws = counters_all = None
        for j, (cap, h, d, c) in enumerate(self._sched_free):
            if cap >= nbytes and c.numel() >= n + 2:
                host_ws, dev_ws, counters_all = h, d, c
                del self._sched_free[j]
                break
        with torch.cuda.stream(fwd_stream):
            if host_ws is None:
                cap = max(1 << 20, 1 << (nbytes - 1).bit_length())
                host_ws = torch.empty(cap, dtype=torch.uint8, pin_memory=self._sched_pinned)
                dev_ws = torch.empty(cap, dtype=torch.uint8, device=self.device)
                counters_all = torch.empty(max(64, n + 2), dtype=torch.int32, device=self.device)
            counters = counters_all[:n + 1]  # [0 .. n) window_done, [n] error; [n + 1]: waves of the launch that have begun
            counters_all[:n + 2].zero_()
            zeroed = torch.cuda.Event()
            zeroed.record(fwd_stream)
        # One scheduled launch becomes resident at a time: every wave of a launch may wait for any other, so two launches
        # dispatched together -- this one while the one before it (of this or another pipeline) is still finding its SIMDs
        # behind a backlog of smoother waves -- could each hold part of the chip and wait for the rest.  The launch before
        # this one counts its waves as they begin; a one-wave gate on this launch's stream waits for all of them.
        dkey = self.device.index or 0
        prev = _last_scheduled.get(dkey)
        prev_keep = None
        if prev is not None and not prev[3].query():
            prev_keep = prev[0]
            binding.check(self.lib.ste_stream_wait_counter(prev[0].data_ptr() + 4 * prev[1], prev[2], counters.data_ptr() + 4 * n,
                                                           float(timeout_s) * 4, C.c_void_p(fwd_stream.cuda_stream)),
                          "ste_stream_wait_counter")
        sc = binding.SteFwdSchedF64()
        sc.nwindows, sc.windows, sc.slice_steps = n, C.addressof(structs), step
        sc.nwaves, sc.nrounds, sc.items = nwaves, int(items.shape[0]), items.ctypes.data
        sc.host_ws, sc.dev_ws, sc.ws_bytes = host_ws.data_ptr(), dev_ws.data_ptr(), nbytes
        sc.window_done, sc.error, sc.timeout_s = counters.data_ptr(), counters.data_ptr() + 4 * n, float(timeout_s)
        sc.started = counters_all.data_ptr() + 4 * (n + 1)
        if timing is not None:
            timing["forward"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            timing["smoothers"] = []
            timing["forward"][0].record(fwd_stream)
        binding.check(self.lib.ste_ukf_forward_sched_f64(C.byref(sc), C.c_void_p(fwd_stream.cuda_stream)),
                      "ste_ukf_forward_sched_f64")
        if timing is not None:
            timing["forward"][1].record(fwd_stream)
        ready = torch.cuda.Event()
        ready.record(fwd_stream)
        # the launch's tables and counters stay alive until the pipeline is synchronised, or every gate and smoother that
        # looks at them has finished (their error word is read then)
        events = []
        self._sched_live.append((host_ws, dev_ws, counters, structs, items, [ready], events, counters_all, prev_keep))
        _last_scheduled[dkey] = (counters_all, n + 1, nwaves, ready, id(self))
        # One smoother launch for the whole sequence, a wave per tile that waits for ITS tile's forward pass (include/ste.h:
        # ste_urtss_backward_sched_f64) -- when every window takes the one-kernel smoother and nothing has to happen behind
        # individual windows; otherwise a gate and a smoother launch per window (below).
        lean = 4096  # launch_backward's bound (csrc/ste_kernels.hip: kLeanSmootherMaxTracks)
        if tile_smoothers is None:
            # measured (profiles/r05_scheduled_forward.txt): ONE sequence of 20 batches 14.5-14.7 ms with tile smoothers against
            # 15.0-15.5 with a smoother per window, a fleet the same either way; sequences that overlap (7 + 13) 15.1-15.3
            # against 14.5-14.7 -- so: when nothing follows this sequence and none is in flight before it
            tile_smoothers = self._sched_tile_smoothers and final and prev_keep is None
        merged = (smooth and after_smoother is None and tile_smoothers and len({bool(st.sog_rate_rts or st.cog_rate_rts) for st in structs}) == 1
                  and all(st.rts_work and not (st.tuning & 0x200) and ((st.tuning & 0x400) or st.B > lean) and st.Nmax > 0 for st in structs))
        if merged:
            okey = ("order",) + skey
            order = self._schedules.get(okey)
            if order is None:
                # tiles in the order the schedule finishes them: last round in which (window, tile) appears
                tile0 = np.concatenate(([0], np.cumsum(ntiles)))
                last = np.full(int(tile0[-1]), -1, dtype=np.int64)
                live = items[..., 0] >= 0
                rr = np.broadcast_to(np.arange(items.shape[0])[:, None], items.shape[:2])[live]
                flat = tile0[items[..., 0][live]] + items[..., 1][live]
                np.maximum.at(last, flat, rr)
                idx = np.argsort(last, kind="stable")
                win = np.searchsorted(tile0, idx, side="right") - 1
                order = np.ascontiguousarray(np.stack([win, idx - tile0[win]], axis=1).astype(np.int32))
                self._schedules[okey] = order
            bwd_stream = self._tail_stream if final else self.bwd_streams[k % len(self.bwd_streams)]
            bwd_stream.wait_event(zeroed)
            # behind the forward launch's residency: a waiting smoother wave must not sit where a forward wave still has to go
            binding.check(self.lib.ste_stream_wait_counter(counters_all.data_ptr() + 4 * (n + 1), nwaves, counters.data_ptr() + 4 * n,
                                                           float(timeout_s) * 4, C.c_void_p(bwd_stream.cuda_stream)),
                          "ste_stream_wait_counter")
            sbytes = int(self.lib.ste_urtss_backward_sched_workspace(n, int(order.shape[0])))
            s_host = s_dev = None
            for j, (cap, h, d) in enumerate(self._sched_free_bwd):
                if cap >= sbytes:
                    s_host, s_dev = h, d
                    del self._sched_free_bwd[j]
                    break
            if s_host is None:
                cap = max(1 << 18, 1 << (sbytes - 1).bit_length())
                with torch.cuda.stream(bwd_stream):
                    s_host = torch.empty(cap, dtype=torch.uint8, pin_memory=self._sched_pinned)
                    s_dev = torch.empty(cap, dtype=torch.uint8, device=self.device)
            bs = binding.SteBwdSchedF64()
            bs.nwindows, bs.windows, bs.slice_steps = n, C.addressof(structs), step
            bs.nitems, bs.items = int(order.shape[0]), order.ctypes.data
            bs.host_ws, bs.dev_ws, bs.ws_bytes = s_host.data_ptr(), s_dev.data_ptr(), sbytes
            bs.progress = dev_ws.data_ptr() + int(self.lib.ste_ukf_forward_sched_progress_offset(n, max(nslices), sum(ntiles), items.shape[0], nwaves))
            bs.error, bs.timeout_s = counters.data_ptr() + 4 * n, float(timeout_s) * 4
            if timing is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                timing["smoothers"].append(ev)
                timing["tile_smoothers"] = True
                ev[0].record(bwd_stream)
            binding.check(self.lib.ste_urtss_backward_sched_f64(C.byref(bs), C.c_void_p(bwd_stream.cuda_stream)),
                          "ste_urtss_backward_sched_f64")
            if timing is not None:
                ev[1].record(bwd_stream)
            done = torch.cuda.Event()
            done.record(bwd_stream)
            self._sched_live[-1] = self._sched_live[-1] + ((s_host, s_dev, order),)
            for db in dbs:
                if getattr(db, "_pipeline_done", None) is None:
                    self._batches.append(weakref.ref(db))
                db._pipeline_done = done
                events.append(done)
            return events
        waited = set()
        for i, db in enumerate(dbs):
            if getattr(db, "_pipeline_done", None) is None:
                self._batches.append(weakref.ref(db))
            if not smooth:
                db._pipeline_done = ready
                events.append(ready)
                continue
            bwd_stream = self.bwd_streams[(k + i) % len(self.bwd_streams)]
            if final and i == n - 1:
                bwd_stream = self._tail_stream
            if id(bwd_stream) not in waited:  # the counters were zeroed on the forward stream: no gate may read them earlier
                bwd_stream.wait_event(zeroed)
                waited.add(id(bwd_stream))
            binding.check(self.lib.ste_stream_wait_counter(counters.data_ptr() + 4 * i, ntiles[i], counters.data_ptr() + 4 * n,
                                                           float(timeout_s) * 4, C.c_void_p(bwd_stream.cuda_stream)),
                          "ste_stream_wait_counter")
            timed = timing is not None and i % int(timing.get("every", 4)) == 0
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                timing["smoothers"].append(ev)
                ev[0].record(bwd_stream)
            db.backward(bwd_stream, mark=False)
            if timed:
                ev[1].record(bwd_stream)
            if after_smoother is not None:
                with torch.cuda.stream(bwd_stream):
                    db._reader_done = after_smoother(i, bwd_stream)
            done = torch.cuda.Event()
            done.record(bwd_stream)
            db._pipeline_done = done
            events.append(done)
        return events

    def _check_scheduled(self):
        """After a synchronisation: the error words of the scheduled launches issued since the last one."""
        live, self._sched_live = self._sched_live, []
        bad = [int(e[2][-1].item()) for e in live]
        self._sched_free = ([(e[0].numel(), e[0], e[1], e[7]) for e in live] + self._sched_free)[:16]
        self._sched_free_bwd = ([(e[9][0].numel(), e[9][0], e[9][1]) for e in live if len(e) > 9] + self._sched_free_bwd)[:16]
        if any(bad):
            raise binding.SteError("a scheduled forward launch could not progress (error word %s: 1 = a forward wave, 2 = a smoother "
                                   "gate waited longer than its bound); results of that sequence are incomplete" % bad)

    def synchronize(self):
        for s in self.fwd_streams + self.bwd_streams:
            s.synchronize()
        if self._tail_stream is not None:
            self._tail_stream.synchronize()
        if self._sched_live:
            self._check_scheduled()

    def shrink(self, forward_streams: int, smoother_streams: int):
        """Drain the pipeline and destroy all but its first ``forward_streams`` forward and ``smoother_streams`` smoother
        streams: the hardware queues of a pipeline that goes on with scheduled launches only (they need two forward streams,
        one when sequences do not overlap; tile smoothers need one smoother stream) go back to the device without a new
        pipeline -- new queues -- being built."""
        if forward_streams < 1 or smoother_streams < 1:
            raise ValueError("forward_streams and smoother_streams must be >= 1")
        self.synchronize()
        for ref in self._batches:  # (events recorded on streams that are about to go)
            db = ref()
            if db is not None and getattr(db, "_pipeline_done", None) is not None:
                db._pipeline_done = None
        self._batches = []
        for dkey in [k for k, v in _last_scheduled.items() if v[4] == id(self)]:
            del _last_scheduled[dkey]
        drop = self.fwd_streams[forward_streams:] + self.bwd_streams[smoother_streams:]
        self.fwd_streams, self.bwd_streams = self.fwd_streams[:forward_streams], self.bwd_streams[:smoother_streams]
        gone = {int(st.cuda_stream) for st in drop}
        del drop
        keep = []
        for h in self._raw:
            if h is not None and h.value and int(h.value) in gone:
                self.lib.ste_stream_destroy(h)
            else:
                keep.append(h)
        self._raw = keep
        self.buffers_needed = len(self.fwd_streams) + len(self.bwd_streams) + 1

    @property
    def closed(self) -> bool:
        return not self._raw and not self.fwd_streams and not self.bwd_streams

    def close(self):
        """Drain and destroy the CU-masked streams.  Idempotent; the pipeline cannot be used afterwards."""
        try:
            if self.fwd_streams or self.bwd_streams or self._tail_stream is not None:
                self.synchronize()
        finally:
            # the events recorded on these streams and the ExternalStream wrappers go first, then the streams themselves
            for ref in self._batches:
                db = ref()
                if db is not None and getattr(db, "_pipeline_done", None) is not None:
                    db._pipeline_done = None
            self._batches = []
            self._sched_live = []
            self._sched_free = []
            self._sched_free_bwd = []
            for dkey in [k for k, v in _last_scheduled.items() if v[4] == id(self)]:  # (drained above: nothing to wait for)
                del _last_scheduled[dkey]
            self.fwd_streams, self.bwd_streams, self._tail_stream = [], [], None
            raw, self._raw = self._raw, []
            for h in raw:
                if h is not None and h.value:
                    self.lib.ste_stream_destroy(h)
            _live_pipelines.discard(self)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: nothing left to report to
            pass


_live_pipelines = weakref.WeakSet()
_atexit_registered = False
_pool = None


def _staging_pool():
    """A few threads for the host-side staging copies of fleet uploads (NumPy slice copies release the GIL)."""
    global _pool
    if _pool is None:
        import os
        from concurrent.futures import ThreadPoolExecutor

        _pool = ThreadPoolExecutor(max_workers=max(1, min(8, len(os.sched_getaffinity(0)) // 2)), thread_name_prefix="ste-stage")
    return _pool
# Time slices per pipelined forward pass when SmootherPipeline is not told.  One: slices are bit-identical and let the waves
# of the passes in flight re-balance at every boundary, but a stream's next slice waits for ALL waves of the one before, and
# with every window advancing in step no smoother has anything to do until the end -- measured (10 000-track batches, 7 + 6
# streams): 0.672 / 0.715 / 0.768 ms per step at 1 / 2 / 4 slices over 100 steps, 0.804 / 0.781 / 0.819 at the driver's 20;
# a 100 000-track fleet: 8.6 / 8.6 / 8.5 ms (profiles/r04_pipeline_sweeps.txt, r04_fleet_sweep.txt).
DEFAULT_SLICES = 1
MAX_PIPELINE_QUEUES = 16


def _close_live_pipelines():
    for pipe in list(_live_pipelines):
        try:
            pipe.close()
        except Exception:
            pass


def _register_atexit():
    # Registered when the first pipeline is built, i.e. after torch has been imported: atexit runs handlers in reverse
    # order of registration, so this one runs BEFORE torch's own teardown and the streams are destroyed while the HIP
    # runtime, torch and any profiler tool are still up.
    global _atexit_registered
    if not _atexit_registered:
        atexit.register(_close_live_pipelines)
        _atexit_registered = True


def prepare_observations(lons: Sequence, lats: Sequence, gaps: Sequence, model: str = "wgs84", device="cuda:0"):
    """Speed / course over ground and their rates for many tracks in one launch (``ste_track_prep_f64``).

    ``lons[b]``, ``lats[b]`` (length T_b, degrees) and ``gaps[b]`` (length T_b - 1, hours) are what ``ShipTrack.read_csv``
    returns.  ``model`` is ``"wgs84"`` (the ShipTrack defaults, geographiclib_distance / _heading) or ``"sphere"``
    (haversine_formula / heading).  Returns one dict per track with ``sog, cog, sog_rate, cog_rate`` (length T_b),
    ``z`` (4, T_b) = the result of ``get_measurements(include_sog=True, include_cog=True)`` (ship_track.py:197-338) and
    ``status`` (always 0 since 0.3.1: the WGS84 model is Karney's solver, which converges on every leg).
    """
    import torch

    lib = binding.require_gpu()
    models = {"sphere": binding.STE_PREP_SPHERE, "wgs84": binding.STE_PREP_WGS84}
    if model not in models:
        raise ValueError(f"model must be one of {sorted(models)}, got {model!r}")
    B = len(lons)
    if B == 0:
        return []
    nobs = np.asarray([len(v) for v in lons], dtype=np.int32)
    for b in range(B):
        if len(lats[b]) != nobs[b] or len(gaps[b]) != nobs[b] - 1:
            raise ValueError(f"track {b}: need len(lat) == len(lon) and len(gaps) == len(lon) - 1")
        if nobs[b] < 2:
            raise IndexError(f"track {b} has fewer than 2 observations (ship_track.py:220 indexes sog[-1])")
    T = int(nobs.max())
    lon = np.zeros((T, B))
    lat = np.zeros((T, B))
    gap = np.ones((max(T - 1, 1), B))
    for b in range(B):
        lon[: nobs[b], b] = lons[b]
        lat[: nobs[b], b] = lats[b]
        gap[: nobs[b] - 1, b] = gaps[b]
    dev = torch.device(device)
    up = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    t_n, t_lon, t_lat, t_gap = up(nobs), up(lon), up(lat), up(gap)
    outs = torch.empty((4, T, B), dtype=torch.float64, device=dev)
    z = torch.empty((T, 4, B), dtype=torch.float64, device=dev)
    s = binding.StePrepBatchF64()
    s.B, s.Tmax, s.model = B, T, models[model]
    s.nobs, s.lon, s.lat, s.gap = t_n.data_ptr(), t_lon.data_ptr(), t_lat.data_ptr(), t_gap.data_ptr()
    s.sog, s.cog, s.sog_rate, s.cog_rate = (outs[i].data_ptr() for i in range(4))
    s.z = z.data_ptr()
    st = torch.zeros((B,), dtype=torch.int32, device=dev)
    s.status = st.data_ptr()
    stream = torch.cuda.current_stream(dev)
    binding.check(lib.ste_track_prep_f64(C.byref(s), C.c_void_p(stream.cuda_stream)), "ste_track_prep_f64")
    h = outs.cpu().numpy()
    zh = z.cpu().numpy()
    sth = st.cpu().numpy()
    if sth.any():
        import warnings

        warnings.warn(f"observation preparation flagged track(s) {np.flatnonzero(sth).tolist()} (status {sth[sth != 0].tolist()})",
                      RuntimeWarning, stacklevel=2)
    res = []
    for b in range(B):
        n = nobs[b]
        res.append({"sog": h[0, :n, b].copy(), "cog": h[1, :n, b].copy(), "sog_rate": h[2, :n, b].copy(),
                    "cog_rate": h[3, :n, b].copy(), "z": np.ascontiguousarray(zh[:n, :, b].T), "status": int(sth[b])})
    return res


def prepare_ship_tracks(ship_tracks: Sequence, device="cuda:0"):
    """Fill ``sog, cog, sog_rate, cog_rate, z`` of every ShipTrack (already ``read_csv``'d) in one launch -- what the
    reference does per ship with calculate_cog / calculate_sog / calculate_*_rate / get_measurements(True, True)
    (examples/example_ukf_rts_smoother_batch.py:33-40).  The distance / heading pair configured on each ShipTrack
    selects the model; anything but the two pairs the reference ships raises."""
    from . import utils

    pairs = {(utils.geographiclib_distance, utils.geographiclib_heading): "wgs84",
             (utils.haversine_formula, utils.heading): "sphere"}
    groups = {}
    for i, st in enumerate(ship_tracks):
        key = pairs.get((st.calc_distance_func, st.calc_heading_func))
        if key is None:
            raise ValueError("prepare_ship_tracks supports the (geographiclib_distance, geographiclib_heading) and "
                             "(haversine_formula, heading) pairs only; call ShipTrack.calculate_* for custom functions")
        groups.setdefault(key, []).append(i)
    for model, idx in groups.items():
        res = prepare_observations([ship_tracks[i].lon for i in idx], [ship_tracks[i].lat for i in idx],
                                   [ship_tracks[i].dts for i in idx], model=model, device=device)
        for i, r in zip(idx, res):
            st = ship_tracks[i]
            st.sog, st.cog, st.sog_rate, st.cog_rate, st.z = r["sog"], r["cog"], r["sog_rate"], r["cog_rate"], r["z"]
    return ship_tracks


def run_batch(hb: HostBatch, device="cuda:0", smooth: bool = True, fuse_gains: bool = True, outputs=None,
              sm_pos: bool = False):
    """Convenience: upload, run forward (+ smoother) as ONE launch each, download.  Returns a dict of NumPy arrays (tracks
    in the caller's order).  ``outputs``: which histories to bring back -- any of "means", "covs", "means_smoothed",
    "covs_smoothed" (default: all that were computed); "status" and "nsteps" always come along.  A fleet larger than a
    chip-full of waves belongs in ``run_fleet``."""
    return _run_batch(hb, device, smooth, fuse_gains, outputs, sm_pos)[0]


def _run_batch(hb, device, smooth, fuse_gains, outputs, sm_pos):
    db = DeviceBatch(hb, device=device, alloc_smoothed=smooth, fuse_gains=fuse_gains, sm_pos=sm_pos)
    if smooth:
        if hb.sog_rate_rts is not None and np.isnan(hb.sog_rate_rts[:, (hb.host_status == 0) if hb.host_status is not None else slice(None)]).any():
            raise IndexError("smoother rate expansion too short for at least one track (unscented.py:287-292,310)")
        db.run()
    else:
        db.forward()
    if outputs is None:
        outputs = ("means", "covs") + (("means_smoothed", "covs_smoothed") if smooth else ())
    inv = None
    if hb.order is not None:  # back to the caller's track order, on the device
        inv = np.empty_like(hb.order)
        inv[hb.order] = np.arange(len(hb.order))
    out = db.download(outputs, None if inv is None else db.torch.from_numpy(inv).to(db.device))
    status = db.status_host()
    if hb.host_status is not None:
        status = status | hb.host_status
    nsteps = hb.nsteps.copy()
    out["status"], out["nsteps"] = (status, nsteps) if inv is None else (status[inv], nsteps[inv])
    return out, db


def fleet_windows(ntracks: int, chunk: int):
    """[lo, hi) ranges that cut ``ntracks`` into ceil(ntracks / chunk) windows of nearly equal size, each a whole number of
    64-track waves (but the last)."""
    n = max(1, -(-ntracks // max(int(chunk), 1)))
    w = -(-ntracks // (n * 64)) * 64
    return [(lo, min(lo + w, ntracks)) for lo in range(0, ntracks, w)]


# Window size of run_fleet when the caller names none: 256 waves of 64 tracks, a quarter of the chip's 1 024 SIMDs, so that
# whole windows make up a chip-full of forward waves.  Measured on a resident 100 000 x 500 fleet
# (profiles/r04_fleet_sweep.txt): 7 windows of 14 336 tracks 8.5 ms, 10 of 10 048 9.1 ms, 13 of 7 744 9.2 ms, one launch 9.5 ms.
FLEET_CHUNK = 16_384
# run_fleet's default: a resident fleet of at most this many windows runs its forward passes as one scheduled launch
SCHEDULED_FLEET_MAX_WINDOWS = 24


def run_fleet(fleet, chunk: int = FLEET_CHUNK, device="cuda:0", smooth: bool = True, outputs=None, pipeline=None,
              slices: Optional[int] = None, sm_pos: bool = False, scheduled: Optional[bool] = None):
    """
    UKF + URTSS over a fleet of any size -- the batch dimension of the reference's example loop
    (examples/example_ukf_rts_smoother_batch.py:19-90, one ship at a time) at the rate the pipelined kernels sustain.

    The fleet is cut into windows of at most ``chunk`` tracks, all nearly the same size (length-bucketed: ``pack_tracks`` lays tracks out longest
    first).  Windows are not copies: a window is the fleet's own tensors seen through a batch struct with
    ``track_stride`` = the fleet's width (include/ste.h), so every window's kernels read the fleet's inputs and write the
    fleet's histories in place.  The windows go through a ``SmootherPipeline``: the forward passes of several windows share
    the chip with the smoothers of the windows before them.

    ``fleet``: a ``HostBatch`` -- window k + 1 is uploaded (page-locked staging, a copy stream) while window k filters,
        and each window's histories come down as soon as its smoother has finished, overlapped with the windows behind it
        -- or a ``DeviceBatch`` that is already resident: nothing moves, the results stay in its tensors.
    ``outputs``: histories to bring back as NumPy arrays (any of "means", "covs", "means_smoothed", "covs_smoothed");
        default for a HostBatch: all that were computed, for a DeviceBatch: none.  "status" and "nsteps" always come along,
        and ``"device_batch"`` is the resident fleet.
    ``pipeline``: a ``SmootherPipeline`` to reuse (otherwise one is built for this call and closed at its end).
    ``scheduled``: a resident fleet's windows go through ``SmootherPipeline.submit_sequence`` -- all their forward passes as
        one launch of resident waves over (tile, time slice) items -- instead of one forward launch per window (same bits;
        measured on a 100 000-track fleet: 7.9-8.1 against 8.4-8.6 ms).  Default (None): yes for a
        resident fleet of up to ``SCHEDULED_FLEET_MAX_WINDOWS`` windows on a lane-per-track pipeline -- a job that is mostly
        fill and drain --, no for longer ones (a long stream of windows re-balances by itself) and for a ``HostBatch``
        (its windows arrive one upload at a time).
    Results are those of ``run_batch`` on the same tracks with the lane-per-track mapping, bit for bit.
    """
    import torch

    if isinstance(fleet, DeviceBatch):
        db, hb, resident = fleet, fleet.hb, True
        if smooth and db.sm_mean is None:
            raise ValueError("this DeviceBatch was built without smoothed outputs (alloc_smoothed=False)")
        outputs = () if outputs is None else tuple(outputs)
        if len(fleet_windows(db.ntracks, chunk)) == 1:  # one window: one launch, the mapping the library picks for its size
            db.run() if smooth else db.forward()
            out = db.download(outputs) if outputs else {}
            out["status"], out["nsteps"], out["device_batch"] = db.status_host(), hb.nsteps.copy(), db
            return out
    else:
        hb, resident = fleet, False
        if len(fleet_windows(hb.B, chunk)) == 1:
            out, db = _run_batch(hb, device, smooth, True, outputs, sm_pos)
            out["device_batch"] = db
            return out
        if smooth and hb.sog_rate_rts is not None and np.isnan(
                hb.sog_rate_rts[:, (hb.host_status == 0) if hb.host_status is not None else slice(None)]).any():
            raise IndexError("smoother rate expansion too short for at least one track (unscented.py:287-292,310)")
        if outputs is None:
            outputs = ("means", "covs") + (("means_smoothed", "covs_smoothed") if smooth else ())
        outputs = tuple(outputs)
        db = DeviceBatch(hb, device=device, alloc_smoothed=smooth, sm_pos=sm_pos, upload=False)
    dev = db.device
    B = db.ntracks
    wins = fleet_windows(B, chunk)
    own_pipe = pipeline is None
    if scheduled is None:
        scheduled = (resident and 1 < len(wins) <= SCHEDULED_FLEET_MAX_WINDOWS and (pipeline is None or pipeline.forward_lanes != 4))
    # (a pipeline built for scheduled launches needs two forward streams, not seven: every stream is a hardware queue, the device has
    #  about two dozen for everything that runs on it, and scheduled launches are the first to suffer when they run out --
    #  DESIGN.md section 5)
    pipe = pipeline or SmootherPipeline(dev, ntracks=wins[0][1] - wins[0][0], slices=slices,
                                        sequence_only=bool(scheduled and resident and len(wins) > 1))
    inv = None
    if hb.order is not None:  # back to the caller's track order
        inv = np.empty_like(hb.order)
        inv[hb.order] = np.arange(len(hb.order))
    # a window's histories come down while the next windows filter -- when tracks stay in place; a length-bucketed fleet
    # is reordered on the device in one pass at the end (DeviceBatch.download)
    stream_down = bool(outputs) and inv is None and len(wins) > 1
    host = {}
    try:
        up = torch.cuda.Stream(dev) if not resident else None
        down = torch.cuda.Stream(dev) if stream_down else None
        if stream_down:
            for name in outputs:
                attr, width = DeviceBatch._OUT[name]
                host[name] = torch.empty((B, hb.Nmax + 1, width), dtype=torch.float64, pin_memory=True)
        keep = []
        if scheduled and resident and len(wins) > 1:
            ws = [db.window(lo, hi) for lo, hi in wins]
            dones = pipe.submit_sequence(ws, smooth=smooth)
            for (lo, hi), win, done in zip(wins, ws, dones):
                if stream_down:
                    down.wait_event(done)
                    with torch.cuda.stream(down):
                        keep.append(win._download_into(outputs, host, lo, hi))
                keep.append(win)
            wins_left = []
        else:
            wins_left = wins
        for i, (lo, hi) in enumerate(wins_left):
            win = db.window(lo, hi) if len(wins) > 1 else db
            if not resident:
                win._uploaded = db.upload_tracks(lo, hi, up)
            done = pipe.submit(win, final=(i == len(wins) - 1), smooth=smooth)
            if stream_down:
                down.wait_event(done)
                with torch.cuda.stream(down):
                    keep.append(win._download_into(outputs, host, lo, hi))
            keep.append(win)
        pipe.synchronize()
        if down is not None:
            down.synchronize()
        db._pipeline_done = None
    finally:
        if own_pipe:
            pipe.close()
    out = {}
    if stream_down:
        for name in outputs:
            a = host[name].numpy()
            out[name] = a.reshape(a.shape[0], a.shape[1], 4, 4) if DeviceBatch._OUT[name][1] == 16 else a
    elif outputs:
        out = db.download(outputs, None if inv is None else torch.from_numpy(inv).to(dev))
    status = db.status.cpu().numpy()
    if hb.host_status is not None:
        status = status | hb.host_status
    nsteps = hb.nsteps.copy()
    out["status"], out["nsteps"] = (status, nsteps) if inv is None else (status[inv], nsteps[inv])
    out["device_batch"] = db
    return out

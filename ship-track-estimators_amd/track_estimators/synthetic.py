"""
Synthetic geodetic ship tracks for the batched UKF + URTSS path.

This module is *input synthesis* only: it produces the arrays a ``ShipTrack`` carries
(``lon, lat, dts, sog, cog, sog_rate, cog_rate, z``; reference ``ship_track.py:70-83``) for a batch of
independent tracks, with the random stream of track ``i`` depending only on ``seed0 + i`` so that a track's
identity does not depend on the batch it sits in (needed for sharding across ranks).

Recipe (one track):
  T observations at ``gap_h`` hour spacing; truth starts at lon~U(-60,60), lat~U(-50,50), speed~U(10,30) km/h,
  heading~U(0,360) and is advanced with the great-circle model using a per-interval sog_rate~N(0,0.05) and
  cog_rate~N(0,0.5); observations = truth lon/lat + N(0, 0.05 deg); z rows 2-3 = true sog/cog;
  ``sog_rate``/``cog_rate`` handed to the filter are backward differences with a leading 0
  (same convention as reference ``ship_track.py:242-246,296-300``).
"""
from __future__ import annotations

import dataclasses

import numpy as np

EARTH_RADIUS = 6378.137  # km, reference constants.py:1


@dataclasses.dataclass
class SyntheticBatch:
    """Arrays of a batch of B synthetic tracks with T observations each (track-major, like B ShipTracks)."""

    lon: np.ndarray  # (B, T)
    lat: np.ndarray  # (B, T)
    dts: np.ndarray  # (B, T-1) hours between observations
    sog: np.ndarray  # (B, T)
    cog: np.ndarray  # (B, T)
    sog_rate: np.ndarray  # (B, T)
    cog_rate: np.ndarray  # (B, T)
    z: np.ndarray  # (B, 4, T) measurement matrix rows lon, lat, sog, cog

    @property
    def ntracks(self) -> int:
        return self.lon.shape[0]

    @property
    def nobs(self) -> int:
        return self.lon.shape[1]


def _advance(lon, lat, u, alpha, dt):
    """Great-circle dead reckoning of the truth (degrees in/out), vectorised over tracks."""
    lam, phi, a = np.radians(lon), np.radians(lat), np.radians(alpha)
    d = u * dt / EARTH_RADIUS
    sd, cd = np.sin(d), np.cos(d)
    lam2 = lam + np.arctan2(sd * np.sin(a), np.cos(phi) * cd - np.sin(phi) * sd * np.cos(a))
    phi2 = np.arcsin(np.sin(phi) * cd + np.cos(phi) * sd * np.cos(a))
    return np.degrees(lam2), np.degrees(phi2)


def make_batch(ntracks: int, nobs: int = 126, gap_h: float = 1.0, seed0: int = 0) -> SyntheticBatch:
    """Generate ``ntracks`` tracks; track ``i`` uses ``np.random.default_rng(seed0 + i)``."""
    B, T = int(ntracks), int(nobs)
    start = np.empty((B, 4))
    srate = np.empty((B, T - 1))
    crate = np.empty((B, T - 1))
    onoise = np.empty((B, 2, T))
    for i in range(B):
        rng = np.random.default_rng(seed0 + i)
        start[i] = (
            rng.uniform(-60.0, 60.0),
            rng.uniform(-50.0, 50.0),
            rng.uniform(10.0, 30.0),
            rng.uniform(0.0, 360.0),
        )
        srate[i] = rng.normal(0.0, 0.05, T - 1)
        crate[i] = rng.normal(0.0, 0.5, T - 1)
        onoise[i] = rng.normal(0.0, 0.05, (2, T))

    lon = np.empty((B, T))
    lat = np.empty((B, T))
    sog = np.empty((B, T))
    cog = np.empty((B, T))
    lon[:, 0], lat[:, 0], sog[:, 0], cog[:, 0] = start.T
    for k in range(T - 1):
        lon[:, k + 1], lat[:, k + 1] = _advance(lon[:, k], lat[:, k], sog[:, k], cog[:, k], gap_h)
        sog[:, k + 1] = sog[:, k] + srate[:, k] * gap_h
        cog[:, k + 1] = cog[:, k] + crate[:, k] * gap_h
    dts = np.full((B, T - 1), float(gap_h))
    sog_rate = np.zeros((B, T))
    cog_rate = np.zeros((B, T))
    sog_rate[:, 1:] = (sog[:, 1:] - sog[:, :-1]) / dts
    cog_rate[:, 1:] = (cog[:, 1:] - cog[:, :-1]) / dts
    zlon = lon + onoise[:, 0]
    zlat = lat + onoise[:, 1]
    z = np.stack([zlon, zlat, sog, cog], axis=1)
    return SyntheticBatch(
        lon=zlon, lat=zlat, dts=dts, sog=sog, cog=cog, sog_rate=sog_rate, cog_rate=cog_rate, z=z
    )


# Filter matrices of the batch example (reference examples/example_ukf_rts_smoother_batch.py:43-52).
def example_matrices():
    H = np.diag([1.0, 1.0, 0.0, 0.0])
    R = np.diag([0.25, 0.25, 0.0, 0.0])
    Q = np.diag([1e-4, 1e-4, 1e-6, 1e-6])
    P = np.diag([1.0, 1.0, 1.0, 1.0])
    return H, Q, R, P

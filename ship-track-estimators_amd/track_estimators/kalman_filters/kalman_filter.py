"""
Filter driver.  Mirrors reference ``track_estimators.kalman_filters.kalman_filter.KalmanFilterBase``
(/root/reference/src/track_estimators/kalman_filters/kalman_filter.py:8-145): same attributes, same ``run`` /
``run_rts_smoother`` signatures and return shapes.  The time loop itself runs on the GPU: ``run`` packs the track into
a batch of one and launches ``ste_ukf_forward_f64``; subclasses provide ``_launch_forward`` / ``rts_step``.
"""
from __future__ import annotations

from typing import List, Tuple, Union

import numpy as np

from ..ship_track import ShipTrack


class KalmanFilterBase:
    """State shared by the filters: running time, control vector, history lists (kalman_filter.py:20-34)."""

    def __init__(self, *args, **kwargs):
        self.time = 0
        self.c = None
        self.means = []
        self.covariances = []
        self.means_smoothed = []
        self.covariances_smoothed = []
        self.dt = None
        self.nsteps = None

    def run(self, nsteps: int, dt: Union[int, float, List[Union[int, float]], np.ndarray], ship_track: ShipTrack,
            *args, **kwargs) -> Tuple[np.ndarray, np.ndarray]:
        """
        Filter ``ship_track`` over ``nsteps`` steps (kalman_filter.py:36-117).

        History slot 0 is the prior, appended before the initial update with ``z[:, 0]``; every step is a predict,
        then an update whenever the accumulated time equals (float ``==``) one of ``cumsum(ship_track.dts)``.
        Returns ``(means (N+1, n), covariances (N+1, n, n))`` of everything appended so far, squeezed.
        """
        if isinstance(dt, (list, np.ndarray)):
            assert len(dt) == nsteps, "dt must be the same length as nsteps"
        else:
            dt = np.ones(nsteps) * dt
        self.dt = dt
        self.nsteps = nsteps
        means, covs, upd_idx, t_end = self._launch_forward(np.asarray(dt, dtype=np.float64), ship_track)
        # bookkeeping the reference does while stepping
        fired = upd_idx[upd_idx >= 0]
        last = int(fired[-1]) if len(fired) else 0
        self.c = np.asarray([ship_track.sog[last], ship_track.cog[last]])
        self.time = t_end
        for k in range(means.shape[0]):
            self.means.append(means[k].reshape(-1, 1))
            self.covariances.append(covs[k])
        self.x = self.means[-1]
        self.P = self.covariances[-1]
        return np.asarray(self.means).squeeze(), np.asarray(self.covariances).squeeze()

    def run_rts_smoother(self, ship_track: ShipTrack) -> Tuple[np.ndarray, np.ndarray]:
        """Smooth the stored history (kalman_filter.py:119-137)."""
        x, P = self.rts_step(np.asarray(self.means), np.asarray(self.covariances), ship_track)
        return x.squeeze(), P.squeeze()

    def predict(self, *args, **kwargs):
        raise NotImplementedError("Predict not implemented.")

    def update(self, *args, **kwargs):
        raise NotImplementedError("Update not implemented.")

    def _launch_forward(self, dt, ship_track):
        raise NotImplementedError("Forward launch not implemented.")

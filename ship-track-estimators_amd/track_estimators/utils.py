"""
Geodesy helpers and time-step generation.  Mirrors reference ``track_estimators.utils``
(/root/reference/src/track_estimators/utils.py:9-199): same function names, argument order (lon1, lat1, lon2, lat2)
and units (km, degrees).  These are O(T) host-side preparation steps, not part of the GPU hot path.

``geographiclib_distance`` / ``geographiclib_heading`` call ``Geodesic.WGS84.Inverse`` in the reference
(utils.py:36,68).  geographiclib is a third-party dependency (``geographiclib>=2.0``, requirements.txt:4) that may be
absent; when it is, the same inverse problem is solved by this package's restatement of the same algorithm (Karney 2013,
``track_estimators.geodesic``: every pair of points converges, nearly antipodal ones included; it reproduces the one
noise-free WGS84 number the reference holds, row 0 of examples/cli_example/output_01203823_predictions.txt, to the last bit).
"""
from __future__ import annotations

from typing import List, Union

import numpy as np

from . import geodesic as _karney
from .constants import EARTH_RADIUS

try:  # pragma: no cover - depends on the environment
    from geographiclib.geodesic import Geodesic as _Geodesic

    _HAVE_GEOGRAPHICLIB = getattr(_Geodesic, "WGS84", None) is not None
except Exception:  # ModuleNotFoundError in this image
    _Geodesic = None
    _HAVE_GEOGRAPHICLIB = False


def _inverse(lat1, lon1, lat2, lon2):
    if _HAVE_GEOGRAPHICLIB:
        res = _Geodesic.WGS84.Inverse(lat1, lon1, lat2, lon2)
        return res["s12"], res["azi1"]
    s12, azi1, _, _ = _karney.inverse(lat1, lon1, lat2, lon2)
    return s12, azi1


def geographiclib_distance(lon1: float, lat1: float, lon2: float, lat2: float) -> float:
    """WGS84 geodesic distance in km; 0 for (nearly) coincident points (utils.py:9-38)."""
    if np.abs(lat1 - lat2) < 1e-8 and np.abs(lon1 - lon2) < 1e-8:
        return 0.0
    s12, _ = _inverse(lat1, lon1, lat2, lon2)
    return s12 * 1e-3


def geographiclib_heading(lon1: float, lat1: float, lon2: float, lat2: float) -> float:
    """WGS84 forward azimuth at point 1 in [0, 360); 0 for (nearly) coincident points (utils.py:41-72)."""
    if np.abs(lat1 - lat2) < 1e-8 and np.abs(lon1 - lon2) < 1e-8:
        return 0.0
    _, azi1 = _inverse(lat1, lon1, lat2, lon2)
    return (azi1 + 360) % 360


def haversine_formula(lon1: float, lat1: float, lon2: float, lat2: float) -> float:
    """Great-circle distance in km on a sphere of radius EARTH_RADIUS (utils.py:75-113; atan2 form, :109)."""
    lon1, lat1, lon2, lat2 = (np.radians(v) for v in (lon1, lat1, lon2, lat2))
    dlat = lat2 - lat1
    dlon = lon2 - lon1
    a = np.sin(dlat / 2.0) ** 2 + np.cos(lat1) * np.cos(lat2) * np.sin(dlon / 2.0) ** 2
    c = 2 * np.arctan2(np.sqrt(a), np.sqrt(1 - a))
    return c * EARTH_RADIUS


def heading(lon1, lat1, lon2, lat2):
    """Initial great-circle bearing from point 1 to point 2, degrees in [0, 360) (utils.py:116-147)."""
    lon1, lat1, lon2, lat2 = (np.radians(v) for v in (lon1, lat1, lon2, lat2))
    dlon = lon2 - lon1
    east = np.sin(dlon) * np.cos(lat2)
    north = np.cos(lat1) * np.sin(lat2) - np.sin(lat1) * np.cos(lat2) * np.cos(dlon)
    return (np.degrees(np.arctan2(east, north)) + 360) % 360


def smooth(y: np.ndarray, box_pts: int) -> np.ndarray:
    """Centred moving average of width ``box_pts`` with zero padding at the ends (utils.py:150-172)."""
    return np.convolve(y, np.ones(box_pts) / box_pts, mode="same")


def generate_dts(dts: Union[np.ndarray, List[Union[int, float]]], substeps: int) -> np.ndarray:
    """Split every observation gap into ``substeps`` equal steps (utils.py:175-199)."""
    out = [gap / substeps for gap in dts for _ in range(substeps)]
    return np.asarray(out)

"""
Geodesy helpers and time-step generation.  Mirrors reference ``track_estimators.utils``
(/root/reference/src/track_estimators/utils.py:9-199): same function names, argument order (lon1, lat1, lon2, lat2)
and units (km, degrees).  These are O(T) host-side preparation steps, not part of the GPU hot path.

``geographiclib_distance`` / ``geographiclib_heading`` call ``Geodesic.WGS84.Inverse`` in the reference
(utils.py:36,68).  geographiclib is a third-party dependency (``geographiclib>=2.0``, requirements.txt:4) that may be
absent; when it is, the WGS84 inverse problem is solved here with Vincenty's iteration on the same ellipsoid
(a = 6378137 m, f = 1/298.257223563), which agrees with Karney's algorithm to better than 1e-9 relative away from
antipodal pairs.
"""
from __future__ import annotations

import math
import warnings
from typing import List, Union

import numpy as np

from .constants import EARTH_RADIUS

try:  # pragma: no cover - depends on the environment
    from geographiclib.geodesic import Geodesic as _Geodesic

    _HAVE_GEOGRAPHICLIB = getattr(_Geodesic, "WGS84", None) is not None
except Exception:  # ModuleNotFoundError in this image
    _Geodesic = None
    _HAVE_GEOGRAPHICLIB = False

_WGS84_A = 6378137.0
_WGS84_F = 1.0 / 298.257223563


def _wgs84_inverse(lat1, lon1, lat2, lon2):
    """Vincenty inverse on WGS84: returns (s12 metres, azi1 degrees in (-180, 180])."""
    a, f = _WGS84_A, _WGS84_F
    b = a * (1.0 - f)
    phi1, phi2 = math.radians(lat1), math.radians(lat2)
    L = math.radians(lon2 - lon1)
    L = (L + math.pi) % (2.0 * math.pi) - math.pi
    U1 = math.atan((1.0 - f) * math.tan(phi1))
    U2 = math.atan((1.0 - f) * math.tan(phi2))
    sU1, cU1, sU2, cU2 = math.sin(U1), math.cos(U1), math.sin(U2), math.cos(U2)
    lam = L
    done = False
    for _ in range(200):
        sl, cl = math.sin(lam), math.cos(lam)
        sin_sigma = math.hypot(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl)
        if sin_sigma == 0.0:
            return 0.0, 0.0
        cos_sigma = sU1 * sU2 + cU1 * cU2 * cl
        sigma = math.atan2(sin_sigma, cos_sigma)
        sin_alpha = cU1 * cU2 * sl / sin_sigma
        cos2_alpha = 1.0 - sin_alpha * sin_alpha
        cos_2sm = cos_sigma - 2.0 * sU1 * sU2 / cos2_alpha if cos2_alpha != 0.0 else 0.0
        Cc = f / 16.0 * cos2_alpha * (4.0 + f * (4.0 - 3.0 * cos2_alpha))
        lam_new = L + (1.0 - Cc) * f * sin_alpha * (
            sigma + Cc * sin_sigma * (cos_2sm + Cc * cos_sigma * (-1.0 + 2.0 * cos_2sm * cos_2sm)))
        done = abs(lam_new - lam) < 1e-15
        lam = lam_new
        if done:
            break
    if not done:
        # the fixed point does not contract for nearly antipodal points; Karney's solver (geographiclib, what the
        # reference calls) has no such limit.  The last iterate is returned, loudly.
        warnings.warn(f"Vincenty's inverse iteration did not converge for ({lat1}, {lon1}) -> ({lat2}, {lon2}) (nearly "
                      "antipodal points); distance and azimuth are approximate.  Install geographiclib for Karney's "
                      "algorithm.", RuntimeWarning, stacklevel=3)
    sl, cl = math.sin(lam), math.cos(lam)
    u2 = cos2_alpha * (a * a - b * b) / (b * b)
    A = 1.0 + u2 / 16384.0 * (4096.0 + u2 * (-768.0 + u2 * (320.0 - 175.0 * u2)))
    Bc = u2 / 1024.0 * (256.0 + u2 * (-128.0 + u2 * (74.0 - 47.0 * u2)))
    dsig = Bc * sin_sigma * (cos_2sm + Bc / 4.0 * (
        cos_sigma * (-1.0 + 2.0 * cos_2sm ** 2) - Bc / 6.0 * cos_2sm * (-3.0 + 4.0 * sin_sigma ** 2) * (-3.0 + 4.0 * cos_2sm ** 2)))
    s12 = b * A * (sigma - dsig)
    azi1 = math.degrees(math.atan2(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl))
    return s12, azi1


def _inverse(lat1, lon1, lat2, lon2):
    if _HAVE_GEOGRAPHICLIB:
        res = _Geodesic.WGS84.Inverse(lat1, lon1, lat2, lon2)
        return res["s12"], res["azi1"]
    return _wgs84_inverse(float(lat1), float(lon1), float(lat2), float(lon2))


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

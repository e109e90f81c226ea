"""Vincenty's inverse iteration on WGS84 -- TEST INFRASTRUCTURE: an independent cross-check of the product's Karney solver
(track_estimators/geodesic.py, csrc/ste_prep.hip).  It was this package's WGS84 fallback in rounds 1-3; it does not converge
for nearly antipodal points and its series are truncated at ~0.1 mm, which is why it is no longer the product path."""
import math
import warnings

_WGS84_A = 6378137.0
_WGS84_F = 1.0 / 298.257223563


def vincenty_inverse(lat1, lon1, lat2, lon2):
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

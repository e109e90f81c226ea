"""
WGS84 inverse geodesic problem by Karney's algorithm -- what ``Geodesic.WGS84.Inverse`` of geographiclib computes for the
reference's ``geographiclib_distance`` / ``geographiclib_heading`` (/root/reference/src/track_estimators/utils.py:36,68).

geographiclib is a third-party dependency of the reference (requirements.txt:4, ``geographiclib>=2.0``; not vendored
under /root/reference, not installed here), so this module restates its published algorithm:

    C. F. F. Karney, "Algorithms for geodesics", J. Geodesy 87 (2013) 43-55, doi:10.1007/s00190-012-0578-z
        eqs. (7)-(25): the series A1, C1, A2, C2, A3, C3 to sixth order; section 5: the starting guess (incl. the astroid
        problem for nearly antipodal points) and Newton's method on the longitude equation, eqs. (44)-(65)

with the structure of its reference implementation (geodesic.py of geographiclib 2.0: argument canonicalisation by
AngDiff / swap / sign flips, the meridional and equatorial special cases, Newton safeguarded by bisection).  Unlike
Vincenty's iteration -- which this package used before, and keeps as a cross-check in its tests -- it converges for EVERY
pair of points, nearly antipodal ones included.

Parity: without geographiclib on either box this is "parity unpinned" beyond what can be pinned: the reference-held values
(tests/test_utils.py:36-60,87-100 and row 0 of examples/cli_example/output_01203823_predictions.txt), agreement with
Vincenty wherever that converges, symmetry, and the antipodal / meridional cases of the paper's section 5
(tests/test_geodesic_karney.py).  ``csrc/ste_prep.hip`` holds the same algorithm for the device.
"""
from __future__ import annotations

import math

WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563

_a, _f = WGS84_A, WGS84_F
_f1 = 1.0 - _f
_e2 = _f * (2.0 - _f)
_ep2 = _e2 / (_f1 * _f1)
_n = _f / (2.0 - _f)
_b = _a * _f1

_EPS = 2.0 ** -52
_TINY = math.sqrt(2.0 ** -1022)
_TOL0 = _EPS
_TOL1 = 200.0 * _TOL0
_TOL2 = math.sqrt(_TOL0)
_TOLB = _TOL0 * _TOL2
_XTHRESH = 1000.0 * _TOL2
_MAXIT1 = 20
_MAXIT2 = _MAXIT1 + 53 + 10
_ETOL2 = 0.1 * _TOL2 / math.sqrt(max(0.001, abs(_f)) * min(1.0, 1.0 - _f / 2.0) / 2.0)

# A3(eps) = sum_j A3X[j] eps^j and C3_l(eps) = eps^l sum_j C3X[l][j] eps^j: eqs. (24), (25), polynomials in n
_A3X = (1.0, (_n - 1.0) / 2.0, (_n * (3.0 * _n - 1.0) - 2.0) / 8.0, ((-_n - 3.0) * _n - 1.0) / 16.0, (-2.0 * _n - 3.0) / 64.0,
        -3.0 / 128.0)
_C3X = (
    None,
    ((1.0 - _n) / 4.0, (1.0 - _n * _n) / 8.0, ((3.0 - _n) * _n + 3.0) / 64.0, (2.0 * _n + 5.0) / 128.0, 3.0 / 128.0),
    ((_n * (_n - 3.0) + 2.0) / 32.0, ((-3.0 * _n - 2.0) * _n + 3.0) / 64.0, (_n + 3.0) / 128.0, 5.0 / 256.0),
    ((_n * (5.0 * _n - 9.0) + 5.0) / 192.0, (9.0 - 10.0 * _n) / 384.0, 7.0 / 512.0),
    ((7.0 - 14.0 * _n) / 512.0, 7.0 / 512.0),
    (21.0 / 2560.0,),
)


def _sq(x):
    return x * x


def _norm(x, y):
    r = math.hypot(x, y)
    return x / r, y / r


def _sum(u, v):
    """Error-free sum: s = round(u + v), t = u + v - s."""
    s = u + v
    up = s - v
    vpp = s - up
    up -= u
    vpp -= v
    t = s if s == 0 else 0.0 - (up + vpp)
    return s, t


def _ang_round(x):
    z = 1.0 / 16.0
    y = abs(x)
    y = z - (z - y) if y < z else y
    return math.copysign(y, x)


def _ang_diff(x, y):
    """y - x reduced to [-180, 180], with the rounding error of the reduction."""
    d, t = _sum(math.remainder(-x, 360.0), math.remainder(y, 360.0))
    d, t = _sum(math.remainder(d, 360.0), t)
    if d == 0 or abs(d) == 180:
        d = math.copysign(d, -t if t != 0 else y - x)
    return d, t


def _sincosd(x, t=0.0):
    """sin, cos of (x + t) degrees with the quadrant reduced exactly."""
    q = round(x / 90.0) if math.isfinite(x) else 0
    r = x - 90.0 * q if math.isfinite(x) else math.nan
    r = math.radians(_ang_round(r + t))
    s, c = math.sin(r), math.cos(r)
    q &= 3
    if q == 1:
        s, c = c, -s
    elif q == 2:
        s, c = -s, -c
    elif q == 3:
        s, c = -c, s
    c = c + 0.0
    if s == 0:
        s = math.copysign(s, x)
    return s, c


def _atan2d(y, x):
    q = 0
    if abs(y) > abs(x):
        q, x, y = 2, y, x
    if x < 0:
        q += 1
        x = -x
    ang = math.degrees(math.atan2(y, x))
    if q == 1:
        ang = math.copysign(180.0, y) - ang
    elif q == 2:
        ang = 90.0 - ang
    elif q == 3:
        ang = -90.0 + ang
    return ang


def _a1m1f(eps):
    eps2 = eps * eps
    t = eps2 * (eps2 * (eps2 + 4.0) + 64.0) / 256.0
    return (t + eps) / (1.0 - eps)


def _c1f(eps):
    eps2 = eps * eps
    d = eps
    c = [0.0] * 7
    c[1] = d * ((6.0 - eps2) * eps2 - 16.0) / 32.0
    d *= eps
    c[2] = d * ((64.0 - 9.0 * eps2) * eps2 - 128.0) / 2048.0
    d *= eps
    c[3] = d * (9.0 * eps2 - 16.0) / 768.0
    d *= eps
    c[4] = d * (3.0 * eps2 - 5.0) / 512.0
    d *= eps
    c[5] = -7.0 * d / 1280.0
    d *= eps
    c[6] = -7.0 * d / 2048.0
    return c


def _a2m1f(eps):
    eps2 = eps * eps
    t = eps2 * (eps2 * (-11.0 * eps2 - 28.0) - 192.0) / 256.0
    return (t - eps) / (1.0 + eps)


def _c2f(eps):
    eps2 = eps * eps
    d = eps
    c = [0.0] * 7
    c[1] = d * ((eps2 + 2.0) * eps2 + 16.0) / 32.0
    d *= eps
    c[2] = d * ((35.0 * eps2 + 64.0) * eps2 + 384.0) / 2048.0
    d *= eps
    c[3] = d * (15.0 * eps2 + 80.0) / 768.0
    d *= eps
    c[4] = d * (7.0 * eps2 + 35.0) / 512.0
    d *= eps
    c[5] = 63.0 * d / 1280.0
    d *= eps
    c[6] = 77.0 * d / 2048.0
    return c


def _a3f(eps):
    v = 0.0
    for cf in reversed(_A3X):
        v = v * eps + cf
    return v


def _c3f(eps):
    c = [0.0] * 6
    mult = 1.0
    for l in range(1, 6):
        mult *= eps
        v = 0.0
        for cf in reversed(_C3X[l]):
            v = v * eps + cf
        c[l] = mult * v
    return c


def _sin_series(sinx, cosx, c):
    """sum_{k >= 1} c[k] sin(2 k x) by Clenshaw summation."""
    ar = 2.0 * (cosx - sinx) * (cosx + sinx)  # 2 cos 2x
    y0 = y1 = 0.0
    for k in range(len(c) - 1, 0, -1):
        y0, y1 = ar * y0 - y1 + c[k], y0
    return 2.0 * sinx * cosx * y0


def _lengths(eps, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2):
    """(s12 / b, m12 / b): eqs. (7), (40)."""
    c1a, c2a = _c1f(eps), _c2f(eps)
    a1, a2 = _a1m1f(eps), _a2m1f(eps)
    m0x = a1 - a2
    a1, a2 = 1.0 + a1, 1.0 + a2
    b1 = _sin_series(ssig2, csig2, c1a) - _sin_series(ssig1, csig1, c1a)
    s12b = a1 * (sig12 + b1)
    b2 = _sin_series(ssig2, csig2, c2a) - _sin_series(ssig1, csig1, c2a)
    j12 = m0x * sig12 + (a1 * b1 - a2 * b2)
    m12b = dn2 * (csig1 * ssig2) - dn1 * (ssig1 * csig2) - csig1 * csig2 * j12
    return s12b, m12b


def _astroid(x, y):
    """Positive root k of k^4 + 2 k^3 - (x^2 + y^2 - 1) k^2 - 2 y^2 k - y^2 = 0, eq. (55)."""
    p, q = x * x, y * y
    r = (p + q - 1.0) / 6.0
    if q == 0 and r <= 0:
        return 0.0
    S = p * q / 4.0
    r2 = r * r
    r3 = r * r2
    disc = S * (S + 2.0 * r3)
    u = r
    if disc >= 0:
        T3 = S + r3
        T3 += -math.sqrt(disc) if T3 < 0 else math.sqrt(disc)
        T = math.copysign(abs(T3) ** (1.0 / 3.0), T3)
        u += T + (r2 / T if T != 0 else 0.0)
    else:
        ang = math.atan2(math.sqrt(-disc), -(S + r3))
        u += 2.0 * r * math.cos(ang / 3.0)
    v = math.sqrt(u * u + q)
    uv = q / (v - u) if u < 0 else u + v
    w = (uv - q) / (2.0 * v)
    return uv / (math.sqrt(uv + w * w) + w)


def _inverse_start(sbet1, cbet1, dn1, sbet2, cbet2, dn2, lam12, slam12, clam12):
    """Starting azimuth for Newton's method (section 5); sig12 >= 0 marks a short line that needs no iteration."""
    sig12 = -1.0
    salp2 = calp2 = dnm = math.nan
    sbet12 = sbet2 * cbet1 - cbet2 * sbet1
    cbet12 = cbet2 * cbet1 + sbet2 * sbet1
    sbet12a = sbet2 * cbet1
    sbet12a += cbet2 * sbet1
    shortline = cbet12 >= 0 and sbet12 < 0.5 and cbet2 * lam12 < 0.5
    if shortline:
        sbetm2 = _sq(sbet1 + sbet2)
        sbetm2 /= sbetm2 + _sq(cbet1 + cbet2)
        dnm = math.sqrt(1.0 + _ep2 * sbetm2)
        omg12 = lam12 / (_f1 * dnm)
        somg12, comg12 = math.sin(omg12), math.cos(omg12)
    else:
        somg12, comg12 = slam12, clam12
    salp1 = cbet2 * somg12
    calp1 = (sbet12 + cbet2 * sbet1 * _sq(somg12) / (1.0 + comg12) if comg12 >= 0
             else sbet12a - cbet2 * sbet1 * _sq(somg12) / (1.0 - comg12))
    ssig12 = math.hypot(salp1, calp1)
    csig12 = sbet1 * sbet2 + cbet1 * cbet2 * comg12
    if shortline and ssig12 < _ETOL2:
        salp2 = cbet1 * somg12
        calp2 = sbet12 - cbet1 * sbet2 * (_sq(somg12) / (1.0 + comg12) if comg12 >= 0 else 1.0 - comg12)
        salp2, calp2 = _norm(salp2, calp2)
        sig12 = math.atan2(ssig12, csig12)
    elif abs(_n) >= 0.1 or csig12 >= 0 or ssig12 >= 6.0 * abs(_n) * math.pi * _sq(cbet1):
        pass  # the spherical estimate is good enough
    else:
        # nearly antipodal: x, y = scaled longitude / latitude offsets from the antipode, eq. (53)
        lam12x = math.atan2(-slam12, -clam12)  # lam12 - pi
        k2 = _sq(sbet1) * _ep2
        eps = k2 / (2.0 * (1.0 + math.sqrt(1.0 + k2)) + k2)
        lamscale = _f * cbet1 * _a3f(eps) * math.pi
        betscale = lamscale * cbet1
        x = lam12x / lamscale
        y = sbet12a / betscale
        if y > -_TOL1 and x > -1.0 - _XTHRESH:
            salp1 = min(1.0, -x)
            calp1 = -math.sqrt(1.0 - _sq(salp1))
        else:
            k = _astroid(x, y)
            omg12a = lamscale * (-x * k / (1.0 + k))
            somg12, comg12 = math.sin(omg12a), -math.cos(omg12a)
            salp1 = cbet2 * somg12
            calp1 = sbet12a - cbet2 * sbet1 * _sq(somg12) / (1.0 - comg12)
    if not (salp1 <= 0):
        salp1, calp1 = _norm(salp1, calp1)
    else:
        salp1, calp1 = 1.0, 0.0
    return sig12, salp1, calp1, salp2, calp2, dnm


def _lambda12(sbet1, cbet1, dn1, sbet2, cbet2, dn2, salp1, calp1, slam120, clam120, diffp):
    """Longitude difference reached with azimuth alp1, minus the target, and its derivative: eqs. (8), (23), (46)."""
    if sbet1 == 0 and calp1 == 0:
        calp1 = -_TINY
    salp0 = salp1 * cbet1
    calp0 = math.hypot(calp1, salp1 * sbet1)
    ssig1, somg1 = sbet1, salp0 * sbet1
    csig1 = comg1 = calp1 * cbet1
    ssig1, csig1 = _norm(ssig1, csig1)
    salp2 = salp0 / cbet2 if cbet2 != cbet1 else salp1
    if cbet2 != cbet1 or abs(sbet2) != -sbet1:
        calp2 = math.sqrt(_sq(calp1 * cbet1) + ((cbet2 - cbet1) * (cbet1 + cbet2) if cbet1 < -sbet1
                                                else (sbet1 - sbet2) * (sbet1 + sbet2))) / cbet2
    else:
        calp2 = abs(calp1)
    ssig2, somg2 = sbet2, salp0 * sbet2
    csig2 = comg2 = calp2 * cbet2
    ssig2, csig2 = _norm(ssig2, csig2)
    sig12 = math.atan2(max(0.0, csig1 * ssig2 - ssig1 * csig2) + 0.0, csig1 * csig2 + ssig1 * ssig2)
    somg12 = max(0.0, comg1 * somg2 - somg1 * comg2) + 0.0
    comg12 = comg1 * comg2 + somg1 * somg2
    eta = math.atan2(somg12 * clam120 - comg12 * slam120, comg12 * clam120 + somg12 * slam120)
    k2 = _sq(calp0) * _ep2
    eps = k2 / (2.0 * (1.0 + math.sqrt(1.0 + k2)) + k2)
    c3a = _c3f(eps)
    b312 = _sin_series(ssig2, csig2, c3a) - _sin_series(ssig1, csig1, c3a)
    domg12 = -_f * _a3f(eps) * salp0 * (sig12 + b312)
    lam12 = eta + domg12
    dlam12 = math.nan
    if diffp:
        if calp2 == 0:
            dlam12 = -2.0 * _f1 * dn1 / sbet1
        else:
            _, m12b = _lengths(eps, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2)
            dlam12 = m12b * _f1 / (calp2 * cbet2)
    return lam12, salp2, calp2, sig12, ssig1, csig1, ssig2, csig2, eps, dlam12


def inverse(lat1, lon1, lat2, lon2):
    """(s12 metres, azi1 degrees in [-180, 180], azi2 degrees, iterations) of the shortest WGS84 geodesic from point 1 to 2."""
    lat1, lon1, lat2, lon2 = float(lat1), float(lon1), float(lat2), float(lon2)
    lon12, lon12s = _ang_diff(lon1, lon2)
    lonsign = math.copysign(1.0, lon12)
    lon12, lon12s = lonsign * lon12, lonsign * lon12s
    lam12 = math.radians(lon12)
    slam12, clam12 = _sincosd(lon12, lon12s)
    lon12s = (180.0 - lon12) - lon12s
    lat1 = _ang_round(math.nan if abs(lat1) > 90 else lat1)
    lat2 = _ang_round(math.nan if abs(lat2) > 90 else lat2)
    swapp = -1.0 if abs(lat1) < abs(lat2) or math.isnan(lat2) else 1.0
    if swapp < 0:
        lonsign *= -1.0
        lat1, lat2 = lat2, lat1
    latsign = math.copysign(1.0, -lat1)
    lat1 *= latsign
    lat2 *= latsign
    # now 0 <= lon12 <= 180, -90 <= lat1 <= 0, lat1 <= lat2 <= -lat1
    sbet1, cbet1 = _sincosd(lat1)
    sbet1 *= _f1
    sbet1, cbet1 = _norm(sbet1, cbet1)
    cbet1 = max(_TINY, cbet1)
    sbet2, cbet2 = _sincosd(lat2)
    sbet2 *= _f1
    sbet2, cbet2 = _norm(sbet2, cbet2)
    cbet2 = max(_TINY, cbet2)
    if cbet1 < -sbet1:
        if cbet2 == cbet1:
            sbet2 = math.copysign(sbet1, sbet2)
    elif abs(sbet2) == -sbet1:
        cbet2 = cbet1
    dn1 = math.sqrt(1.0 + _ep2 * _sq(sbet1))
    dn2 = math.sqrt(1.0 + _ep2 * _sq(sbet2))
    numit = 0
    s12x = math.nan
    meridian = lat1 == -90 or slam12 == 0
    if meridian:
        calp1, salp1 = clam12, slam12
        calp2, salp2 = 1.0, 0.0
        ssig1, csig1 = sbet1, calp1 * cbet1
        ssig2, csig2 = sbet2, calp2 * cbet2
        sig12 = math.atan2(max(0.0, csig1 * ssig2 - ssig1 * csig2) + 0.0, csig1 * csig2 + ssig1 * ssig2)
        s12x, m12x = _lengths(_n, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2)
        if sig12 < 1 or m12x >= 0:
            if sig12 < 3.0 * _TINY or (sig12 < _TOL0 and (s12x < 0 or m12x < 0)):
                sig12 = m12x = s12x = 0.0
            s12x *= _b
        else:
            meridian = False  # (prolate ellipsoids only)
    if not meridian and sbet1 == 0 and lon12s >= _f * 180.0:
        # along the equator (sbet2 == 0 too)
        calp1 = calp2 = 0.0
        salp1 = salp2 = 1.0
        s12x = _a * lam12
    elif not meridian:
        sig12, salp1, calp1, salp2, calp2, dnm = _inverse_start(sbet1, cbet1, dn1, sbet2, cbet2, dn2, lam12, slam12, clam12)
        if sig12 >= 0:
            s12x = sig12 * _b * dnm  # short line
        else:
            # Newton's method on lam12(alp1) = target, bracketed; bisection when a step leaves the bracket
            tripn = tripb = False
            salp1a, calp1a, salp1b, calp1b = _TINY, 1.0, _TINY, -1.0
            while numit < _MAXIT2:
                (v, salp2, calp2, sig12, ssig1, csig1, ssig2, csig2, eps, dv) = _lambda12(
                    sbet1, cbet1, dn1, sbet2, cbet2, dn2, salp1, calp1, slam12, clam12, numit < _MAXIT1)
                if tripb or not (abs(v) >= (8.0 if tripn else 1.0) * _TOL0):
                    break
                if v > 0 and (numit > _MAXIT1 or calp1 / salp1 > calp1b / salp1b):
                    salp1b, calp1b = salp1, calp1
                elif v < 0 and (numit > _MAXIT1 or calp1 / salp1 < calp1a / salp1a):
                    salp1a, calp1a = salp1, calp1
                numit += 1
                if numit < _MAXIT1 and dv > 0:
                    dalp1 = -v / dv
                    sdalp1, cdalp1 = math.sin(dalp1), math.cos(dalp1)
                    nsalp1 = salp1 * cdalp1 + calp1 * sdalp1
                    if nsalp1 > 0 and abs(dalp1) < math.pi:
                        calp1 = calp1 * cdalp1 - salp1 * sdalp1
                        salp1 = nsalp1
                        salp1, calp1 = _norm(salp1, calp1)
                        tripn = abs(v) <= 16.0 * _TOL0
                        continue
                salp1 = (salp1a + salp1b) / 2.0
                calp1 = (calp1a + calp1b) / 2.0
                salp1, calp1 = _norm(salp1, calp1)
                tripn = False
                tripb = (abs(salp1a - salp1) + (calp1a - calp1) < _TOLB or abs(salp1 - salp1b) + (calp1 - calp1b) < _TOLB)
            s12x, _ = _lengths(eps, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2)
            s12x *= _b
    s12 = 0.0 + s12x
    if swapp < 0:
        salp1, salp2 = salp2, salp1
        calp1, calp2 = calp2, calp1
    salp1 *= swapp * lonsign
    calp1 *= swapp * latsign
    salp2 *= swapp * lonsign
    calp2 *= swapp * latsign
    return s12, _atan2d(salp1, calp1), _atan2d(salp2, calp2), numit

// ste_prep.hip — observation preparation for a batch of tracks: speed / course over ground and their rates.
//
// Device counterpart of ShipTrack.calculate_sog / calculate_cog / calculate_sog_rate / calculate_cog_rate /
// get_measurements (reference src/track_estimators/ship_track.py:197-338) for many tracks at once: one thread per
// (observation, track), track index fastest, every read and write coalesced.  The kernel is pure streaming work
// (24 B in, 64 B out per observation) with ~10 transcendentals per observation on the sphere and a short fixed-point
// Newton iteration on the ellipsoid, so it is HBM/latency trivial next to the filter; it exists so that raw lon/lat/time can
// go to smoothed tracks without a per-ship Python loop.
//
//   model 0  sphere of radius 6378.137 km: haversine_formula + heading        (reference utils.py:75-147)
//   model 1  WGS84 inverse geodesic: geographiclib_distance + _heading         (reference utils.py:9-72)
//            geographiclib itself is a third-party dependency that is not part of the reference tree; the inverse
//            problem is solved by its published algorithm (Karney 2013), exactly as track_estimators/geodesic.py does on
//            the host (it reproduces the reference's CLI fixture to the last bit, tests/test_geodesic_karney.py).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/ste.h"
#include "ste_err.h"
#include "ste_math.h"

// The reference's formulas give exact zeros for coincident points (distance 0, heading atan2(0, 0) = 0) and
// 0 / 0 = NaN on a duplicate timestamp; a fused multiply-add across `lat2 * k - lat1 * k` would turn those zeros into
// rounding residue.  This file is therefore compiled without contraction (the library's build uses -ffp-contract=fast).
#pragma clang fp contract(off)

namespace ste {
namespace {

constexpr double kDeg2Rad = 0.017453292519943295;  // np.radians multiplies by this double
constexpr double kRad2Deg = 57.29577951308232;     // np.degrees
constexpr double kPi = 3.141592653589793;
constexpr double kEarthKm = 6378.137;
constexpr double kWgsA = 6378137.0;
constexpr double kWgsF = 1.0 / 298.257223563;

struct Leg {
    double dist_km;
    double head_deg;
    bool converged;  // always true since 0.3.1 (Karney's solver converges for every pair of points); kept for the status word
};

// utils.py:75-147 — haversine (atan2 form) and the initial great-circle bearing in [0, 360)
__device__ Leg sphere_leg(double lon1, double lat1, double lon2, double lat2) {
    lon1 *= kDeg2Rad;
    lat1 *= kDeg2Rad;
    lon2 *= kDeg2Rad;
    lat2 *= kDeg2Rad;
    const double dlat = lat2 - lat1, dlon = lon2 - lon1;
    const double sh = sin(dlat / 2.0), sl = sin(dlon / 2.0);
    const double c1 = cos(lat1), c2 = cos(lat2);
    const double a = sh * sh + c1 * c2 * sl * sl;
    Leg r;
    r.converged = true;
    r.dist_km = 2.0 * atan2(sqrt(a), sqrt(1.0 - a)) * kEarthKm;
    const double east = sin(dlon) * c2;
    const double north = c1 * sin(lat2) - sin(lat1) * c2 * cos(dlon);
    r.head_deg = floored_mod360(atan2(east, north) * kRad2Deg + 360.0);
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// utils.py:9-72: Geodesic.WGS84.Inverse.  geographiclib is a third-party dependency of the reference (requirements.txt:4,
// not under /root/reference); this is its published algorithm -- C. F. F. Karney, "Algorithms for geodesics", J. Geodesy
// 87 (2013) 43-55: series A1, C1, A2, C2, A3, C3 to sixth order (eqs. 7-25), starting guess incl. the astroid problem for
// nearly antipodal points and Newton's method safeguarded by bisection (section 5) -- the same restatement as
// track_estimators/geodesic.py on the host, function for function.  Every pair of points converges.
// ---------------------------------------------------------------------------------------------------------------
constexpr double kF1 = 1.0 - kWgsF;
constexpr double kE2 = kWgsF * (2.0 - kWgsF);
constexpr double kEp2 = kE2 / (kF1 * kF1);
constexpr double kN = kWgsF / (2.0 - kWgsF);
constexpr double kWgsB = kWgsA * kF1;
constexpr double kEps = 2.220446049250313e-16;   // 2^-52
constexpr double kTiny = 1.4916681462400413e-154;  // sqrt(2^-1022)
constexpr double kTol0 = kEps;
constexpr double kTol1 = 200.0 * kTol0;
constexpr double kTol2 = 1.4901161193847656e-08;  // sqrt(tol0)
constexpr double kTolB = kTol0 * kTol2;
constexpr double kXThresh = 1000.0 * kTol2;
constexpr int kMaxIt1 = 20, kMaxIt2 = kMaxIt1 + 53 + 10;

struct SC {
    double s, c;
};
__device__ __forceinline__ double sq(double x) { return x * x; }
__device__ __forceinline__ SC norm2(double x, double y) {
    const double r = hypot(x, y);
    return {x / r, y / r};
}
__device__ __forceinline__ double ang_round(double x) {
    const double z = 1.0 / 16.0;
    double y = fabs(x);
    y = y < z ? z - (z - y) : y;
    return copysign(y, x);
}
// error-free sum: s = round(u + v), t = u + v - s
__device__ __forceinline__ double two_sum(double u, double v, double& t) {
    const double s = u + v;
    double up = s - v, vpp = s - up;
    up -= u;
    vpp -= v;
    t = s == 0.0 ? s : 0.0 - (up + vpp);
    return s;
}
// y - x reduced to [-180, 180] with the rounding error of the reduction
__device__ __forceinline__ double ang_diff(double x, double y, double& t) {
    double t1;
    double d = two_sum(remainder(-x, 360.0), remainder(y, 360.0), t1);
    d = two_sum(remainder(d, 360.0), t1, t);
    if (d == 0.0 || fabs(d) == 180.0) d = copysign(d, t != 0.0 ? -t : y - x);
    return d;
}
// sin, cos of (x + t) degrees with the quadrant reduced exactly
__device__ __forceinline__ SC sincosd(double x, double t) {
    const bool fin = x * 0.0 == 0.0;
    const double qd = fin ? rint(x / 90.0) : 0.0;
    double r = fin ? x - 90.0 * qd : x * 0.0 + (x - x);
    r = ang_round(r + t) * kDeg2Rad;
    double s = sin(r), c = cos(r);
    const int q = ((int)qd) & 3;
    if (q == 1) {
        const double u = s;
        s = c;
        c = -u;
    } else if (q == 2) {
        s = -s;
        c = -c;
    } else if (q == 3) {
        const double u = s;
        s = -c;
        c = u;
    }
    c = c + 0.0;
    if (s == 0.0) s = copysign(s, x);
    return {s, c};
}
__device__ __forceinline__ double atan2d(double y, double x) {
    int q = 0;
    if (fabs(y) > fabs(x)) {
        q = 2;
        const double u = x;
        x = y;
        y = u;
    }
    if (x < 0.0) {
        q += 1;
        x = -x;
    }
    double ang = atan2(y, x) * kRad2Deg;
    if (q == 1)
        ang = copysign(180.0, y) - ang;
    else if (q == 2)
        ang = 90.0 - ang;
    else if (q == 3)
        ang = -90.0 + ang;
    return ang;
}
__device__ __forceinline__ double a1m1f(double eps) {
    const double e2 = eps * eps, t = e2 * (e2 * (e2 + 4.0) + 64.0) / 256.0;
    return (t + eps) / (1.0 - eps);
}
__device__ __forceinline__ void c1f(double eps, double (&c)[7]) {
    const double e2 = eps * eps;
    double d = eps;
    c[0] = 0.0;
    c[1] = d * ((6.0 - e2) * e2 - 16.0) / 32.0;
    d *= eps;
    c[2] = d * ((64.0 - 9.0 * e2) * e2 - 128.0) / 2048.0;
    d *= eps;
    c[3] = d * (9.0 * e2 - 16.0) / 768.0;
    d *= eps;
    c[4] = d * (3.0 * e2 - 5.0) / 512.0;
    d *= eps;
    c[5] = -7.0 * d / 1280.0;
    d *= eps;
    c[6] = -7.0 * d / 2048.0;
}
__device__ __forceinline__ double a2m1f(double eps) {
    const double e2 = eps * eps, t = e2 * (e2 * (-11.0 * e2 - 28.0) - 192.0) / 256.0;
    return (t - eps) / (1.0 + eps);
}
__device__ __forceinline__ void c2f(double eps, double (&c)[7]) {
    const double e2 = eps * eps;
    double d = eps;
    c[0] = 0.0;
    c[1] = d * ((e2 + 2.0) * e2 + 16.0) / 32.0;
    d *= eps;
    c[2] = d * ((35.0 * e2 + 64.0) * e2 + 384.0) / 2048.0;
    d *= eps;
    c[3] = d * (15.0 * e2 + 80.0) / 768.0;
    d *= eps;
    c[4] = d * (7.0 * e2 + 35.0) / 512.0;
    d *= eps;
    c[5] = 63.0 * d / 1280.0;
    d *= eps;
    c[6] = 77.0 * d / 2048.0;
}
// A3(eps) and C3_l(eps): eqs. (24), (25), polynomials in n = f / (2 - f)
__device__ __forceinline__ double a3f(double eps) {
    const double a1 = (kN - 1.0) / 2.0, a2 = (kN * (3.0 * kN - 1.0) - 2.0) / 8.0, a3 = ((-kN - 3.0) * kN - 1.0) / 16.0,
                 a4 = (-2.0 * kN - 3.0) / 64.0, a5 = -3.0 / 128.0;
    return ((((a5 * eps + a4) * eps + a3) * eps + a2) * eps + a1) * eps + 1.0;
}
__device__ __forceinline__ void c3f(double eps, double (&c)[7]) {
    const double c10 = (1.0 - kN) / 4.0, c11 = (1.0 - kN * kN) / 8.0, c12 = ((3.0 - kN) * kN + 3.0) / 64.0,
                 c13 = (2.0 * kN + 5.0) / 128.0, c14 = 3.0 / 128.0;
    const double c20 = (kN * (kN - 3.0) + 2.0) / 32.0, c21 = ((-3.0 * kN - 2.0) * kN + 3.0) / 64.0, c22 = (kN + 3.0) / 128.0,
                 c23 = 5.0 / 256.0;
    const double c30 = (kN * (5.0 * kN - 9.0) + 5.0) / 192.0, c31 = (9.0 - 10.0 * kN) / 384.0, c32 = 7.0 / 512.0;
    const double c40 = (7.0 - 14.0 * kN) / 512.0, c41 = 7.0 / 512.0, c50 = 21.0 / 2560.0;
    double m = eps;
    c[0] = 0.0;
    c[1] = m * ((((c14 * eps + c13) * eps + c12) * eps + c11) * eps + c10);
    m *= eps;
    c[2] = m * (((c23 * eps + c22) * eps + c21) * eps + c20);
    m *= eps;
    c[3] = m * ((c32 * eps + c31) * eps + c30);
    m *= eps;
    c[4] = m * (c41 * eps + c40);
    m *= eps;
    c[5] = m * c50;
    c[6] = 0.0;
}
// sum_{k = 1..K} c[k] sin(2 k x), Clenshaw
template <int K>
__device__ __forceinline__ double sin_series(double sinx, double cosx, const double (&c)[7]) {
    const double ar = 2.0 * (cosx - sinx) * (cosx + sinx);
    double y0 = 0.0, y1 = 0.0;
#pragma unroll
    for (int k = K; k >= 1; --k) {
        const double y = ar * y0 - y1 + c[k];
        y1 = y0;
        y0 = y;
    }
    return 2.0 * sinx * cosx * y0;
}
// s12 / b and m12 / b: eqs. (7), (40)
__device__ void lengths(double eps, double sig12, double ssig1, double csig1, double dn1, double ssig2, double csig2,
                        double dn2, double& s12b, double& m12b) {
    double ca[7], cb[7];
    c1f(eps, ca);
    c2f(eps, cb);
    double a1 = a1m1f(eps), a2 = a2m1f(eps);
    const double m0x = a1 - a2;
    a1 = 1.0 + a1;
    a2 = 1.0 + a2;
    const double b1 = sin_series<6>(ssig2, csig2, ca) - sin_series<6>(ssig1, csig1, ca);
    s12b = a1 * (sig12 + b1);
    const double b2 = sin_series<6>(ssig2, csig2, cb) - sin_series<6>(ssig1, csig1, cb);
    const double j12 = m0x * sig12 + (a1 * b1 - a2 * b2);
    m12b = dn2 * (csig1 * ssig2) - dn1 * (ssig1 * csig2) - csig1 * csig2 * j12;
}
// positive root of the astroid quartic, eq. (55)
__device__ double astroid(double x, double y) {
    const double p = x * x, q = y * y, r = (p + q - 1.0) / 6.0;
    if (q == 0.0 && r <= 0.0) return 0.0;
    const double S = p * q / 4.0, r2 = r * r, r3 = r * r2, disc = S * (S + 2.0 * r3);
    double u = r;
    if (disc >= 0.0) {
        double T3 = S + r3;
        T3 += T3 < 0.0 ? -sqrt(disc) : sqrt(disc);
        const double T = cbrt(T3);
        u += T + (T != 0.0 ? r2 / T : 0.0);
    } else {
        const double ang = atan2(sqrt(-disc), -(S + r3));
        u += 2.0 * r * cos(ang / 3.0);
    }
    const double v = sqrt(u * u + q);
    const double uv = u < 0.0 ? q / (v - u) : u + v;
    const double w = (uv - q) / (2.0 * v);
    return uv / (sqrt(uv + w * w) + w);
}

struct Start {
    double sig12, salp1, calp1, salp2, calp2, dnm;
};
__device__ Start inverse_start(double sbet1, double cbet1, double sbet2, double cbet2, double lam12, double slam12,
                               double clam12) {
    Start o;
    o.sig12 = -1.0;
    o.salp2 = o.calp2 = o.dnm = 0.0;
    const double sbet12 = sbet2 * cbet1 - cbet2 * sbet1, cbet12 = cbet2 * cbet1 + sbet2 * sbet1;
    double sbet12a = sbet2 * cbet1;
    sbet12a += cbet2 * sbet1;
    const bool shortline = cbet12 >= 0.0 && sbet12 < 0.5 && cbet2 * lam12 < 0.5;
    double somg12, comg12;
    if (shortline) {
        double sbetm2 = sq(sbet1 + sbet2);
        sbetm2 /= sbetm2 + sq(cbet1 + cbet2);
        o.dnm = sqrt(1.0 + kEp2 * sbetm2);
        const double omg12 = lam12 / (kF1 * o.dnm);
        somg12 = sin(omg12);
        comg12 = cos(omg12);
    } else {
        somg12 = slam12;
        comg12 = clam12;
    }
    double salp1 = cbet2 * somg12;
    double calp1 = comg12 >= 0.0 ? sbet12 + cbet2 * sbet1 * sq(somg12) / (1.0 + comg12)
                                 : sbet12a - cbet2 * sbet1 * sq(somg12) / (1.0 - comg12);
    const double ssig12 = hypot(salp1, calp1), csig12 = sbet1 * sbet2 + cbet1 * cbet2 * comg12;
    // etol2 = 0.1 tol2 / sqrt(max(0.001, |f|) min(1, 1 - f/2) / 2)
    const double etol2 = 0.1 * kTol2 / sqrt(fmax(0.001, kWgsF) * fmin(1.0, 1.0 - kWgsF / 2.0) / 2.0);
    if (shortline && ssig12 < etol2) {
        const double sa = cbet1 * somg12;
        const double ca = sbet12 - cbet1 * sbet2 * (comg12 >= 0.0 ? sq(somg12) / (1.0 + comg12) : 1.0 - comg12);
        const SC nn = norm2(sa, ca);
        o.salp2 = nn.s;
        o.calp2 = nn.c;
        o.sig12 = atan2(ssig12, csig12);
    } else if (csig12 >= 0.0 || ssig12 >= 6.0 * kN * kPi * sq(cbet1)) {
        // the spherical estimate is good enough
    } else {
        // nearly antipodal: scaled offsets from the antipode, eq. (53)
        const double lam12x = atan2(-slam12, -clam12);
        const double k2 = sq(sbet1) * kEp2, eps = k2 / (2.0 * (1.0 + sqrt(1.0 + k2)) + k2);
        const double lamscale = kWgsF * cbet1 * a3f(eps) * kPi, betscale = lamscale * cbet1;
        const double x = lam12x / lamscale, y = sbet12a / betscale;
        if (y > -kTol1 && x > -1.0 - kXThresh) {
            salp1 = fmin(1.0, -x);
            calp1 = -sqrt(1.0 - sq(salp1));
        } else {
            const double k = astroid(x, y);
            const double omg12a = lamscale * (-x * k / (1.0 + k));
            somg12 = sin(omg12a);
            comg12 = -cos(omg12a);
            salp1 = cbet2 * somg12;
            calp1 = sbet12a - cbet2 * sbet1 * sq(somg12) / (1.0 - comg12);
        }
    }
    if (!(salp1 <= 0.0)) {
        const SC nn = norm2(salp1, calp1);
        o.salp1 = nn.s;
        o.calp1 = nn.c;
    } else {
        o.salp1 = 1.0;
        o.calp1 = 0.0;
    }
    return o;
}

struct Lam {
    double lam12, salp2, calp2, sig12, ssig1, csig1, ssig2, csig2, eps, dlam12;
};
// longitude difference reached with azimuth alp1 minus the target, and its derivative: eqs. (8), (23), (46)
__device__ Lam lambda12(double sbet1, double cbet1, double dn1, double sbet2, double cbet2, double dn2, double salp1,
                        double calp1, double slam120, double clam120, bool diffp) {
    Lam o;
    if (sbet1 == 0.0 && calp1 == 0.0) calp1 = -kTiny;
    const double salp0 = salp1 * cbet1, calp0 = hypot(calp1, salp1 * sbet1);
    const double somg1 = salp0 * sbet1, comg1 = calp1 * cbet1;
    SC s1 = norm2(sbet1, comg1);
    o.ssig1 = s1.s;
    o.csig1 = s1.c;
    o.salp2 = cbet2 != cbet1 ? salp0 / cbet2 : salp1;
    if (cbet2 != cbet1 || fabs(sbet2) != -sbet1)
        o.calp2 = sqrt(sq(calp1 * cbet1) + (cbet1 < -sbet1 ? (cbet2 - cbet1) * (cbet1 + cbet2) : (sbet1 - sbet2) * (sbet1 + sbet2))) /
                  cbet2;
    else
        o.calp2 = fabs(calp1);
    const double somg2 = salp0 * sbet2, comg2 = o.calp2 * cbet2;
    SC s2 = norm2(sbet2, comg2);
    o.ssig2 = s2.s;
    o.csig2 = s2.c;
    o.sig12 = atan2(fmax(0.0, o.csig1 * o.ssig2 - o.ssig1 * o.csig2) + 0.0, o.csig1 * o.csig2 + o.ssig1 * o.ssig2);
    const double somg12 = fmax(0.0, comg1 * somg2 - somg1 * comg2) + 0.0, comg12 = comg1 * comg2 + somg1 * somg2;
    const double eta = atan2(somg12 * clam120 - comg12 * slam120, comg12 * clam120 + somg12 * slam120);
    const double k2 = sq(calp0) * kEp2;
    o.eps = k2 / (2.0 * (1.0 + sqrt(1.0 + k2)) + k2);
    double c3[7];
    c3f(o.eps, c3);
    const double b312 = sin_series<5>(o.ssig2, o.csig2, c3) - sin_series<5>(o.ssig1, o.csig1, c3);
    const double domg12 = -kWgsF * a3f(o.eps) * salp0 * (o.sig12 + b312);
    o.lam12 = eta + domg12;
    o.dlam12 = 0.0;
    if (diffp) {
        if (o.calp2 == 0.0) {
            o.dlam12 = -2.0 * kF1 * dn1 / sbet1;
        } else {
            double s12b, m12b;
            lengths(o.eps, o.sig12, o.ssig1, o.csig1, dn1, o.ssig2, o.csig2, dn2, s12b, m12b);
            o.dlam12 = m12b * kF1 / (o.calp2 * cbet2);
        }
    }
    return o;
}

__device__ Leg wgs84_leg(double lon1, double lat1, double lon2, double lat2) {
    Leg r{0.0, 0.0, true};
    if (fabs(lat1 - lat2) < 1e-8 && fabs(lon1 - lon2) < 1e-8) return r;  // utils.py:32-33, :64-65
    double lon12s;
    double lon12 = ang_diff(lon1, lon2, lon12s);
    double lonsign = copysign(1.0, lon12);
    lon12 *= lonsign;
    lon12s *= lonsign;
    const double lam12 = lon12 * kDeg2Rad;
    const SC sl = sincosd(lon12, lon12s);
    const double slam12 = sl.s, clam12 = sl.c;
    lon12s = (180.0 - lon12) - lon12s;
    const double nan = lat1 * 0.0 + (lat1 - lat1) + __builtin_nan("");
    lat1 = ang_round(fabs(lat1) > 90.0 ? nan : lat1);
    lat2 = ang_round(fabs(lat2) > 90.0 ? nan : lat2);
    const double swapp = (fabs(lat1) < fabs(lat2) || lat2 != lat2) ? -1.0 : 1.0;
    if (swapp < 0.0) {
        lonsign = -lonsign;
        const double u = lat1;
        lat1 = lat2;
        lat2 = u;
    }
    const double latsign = copysign(1.0, -lat1);
    lat1 *= latsign;
    lat2 *= latsign;
    // now 0 <= lon12 <= 180, -90 <= lat1 <= 0, lat1 <= lat2 <= -lat1
    SC b1 = sincosd(lat1, 0.0);
    b1 = norm2(b1.s * kF1, b1.c);
    double sbet1 = b1.s, cbet1 = fmax(kTiny, b1.c);
    SC b2 = sincosd(lat2, 0.0);
    b2 = norm2(b2.s * kF1, b2.c);
    double sbet2 = b2.s, cbet2 = fmax(kTiny, b2.c);
    if (cbet1 < -sbet1) {
        if (cbet2 == cbet1) sbet2 = copysign(sbet1, sbet2);
    } else if (fabs(sbet2) == -sbet1) {
        cbet2 = cbet1;
    }
    const double dn1 = sqrt(1.0 + kEp2 * sq(sbet1)), dn2 = sqrt(1.0 + kEp2 * sq(sbet2));
    double salp1 = 0.0, calp1 = 0.0, salp2 = 0.0, calp2 = 0.0, s12x = nan;
    bool meridian = lat1 == -90.0 || slam12 == 0.0;
    if (meridian) {
        calp1 = clam12;
        salp1 = slam12;
        calp2 = 1.0;
        salp2 = 0.0;
        const double ssig1 = sbet1, csig1 = calp1 * cbet1, ssig2 = sbet2, csig2 = calp2 * cbet2;
        double sig12 = atan2(fmax(0.0, csig1 * ssig2 - ssig1 * csig2) + 0.0, csig1 * csig2 + ssig1 * ssig2);
        double m12x;
        lengths(kN, sig12, ssig1, csig1, dn1, ssig2, csig2, dn2, s12x, m12x);
        if (sig12 < 1.0 || m12x >= 0.0) {
            if (sig12 < 3.0 * kTiny || (sig12 < kTol0 && (s12x < 0.0 || m12x < 0.0))) s12x = 0.0;
            s12x *= kWgsB;
        } else {
            meridian = false;  // (prolate ellipsoids only)
        }
    }
    if (!meridian && sbet1 == 0.0 && lon12s >= kWgsF * 180.0) {
        calp1 = calp2 = 0.0;  // along the equator
        salp1 = salp2 = 1.0;
        s12x = kWgsA * lam12;
    } else if (!meridian) {
        const Start st = inverse_start(sbet1, cbet1, sbet2, cbet2, lam12, slam12, clam12);
        salp1 = st.salp1;
        calp1 = st.calp1;
        if (st.sig12 >= 0.0) {
            salp2 = st.salp2;
            calp2 = st.calp2;
            s12x = st.sig12 * kWgsB * st.dnm;  // short line
        } else {
            // Newton's method on lam12(alp1) = target, bracketed; bisection when a step leaves the bracket.  At most
            // kMaxIt2 = 83 passes (WGS84, random input: 2.85 on average; Karney 2013 section 5)
            bool tripn = false, tripb = false;
            double salp1a = kTiny, calp1a = 1.0, salp1b = kTiny, calp1b = -1.0;
            Lam L = {};
            for (int numit = 0; numit < kMaxIt2;) {
                L = lambda12(sbet1, cbet1, dn1, sbet2, cbet2, dn2, salp1, calp1, slam12, clam12, numit < kMaxIt1);
                const double v = L.lam12;
                if (tripb || !(fabs(v) >= (tripn ? 8.0 : 1.0) * kTol0)) break;
                if (v > 0.0 && (numit > kMaxIt1 || calp1 / salp1 > calp1b / salp1b)) {
                    salp1b = salp1;
                    calp1b = calp1;
                } else if (v < 0.0 && (numit > kMaxIt1 || calp1 / salp1 < calp1a / salp1a)) {
                    salp1a = salp1;
                    calp1a = calp1;
                }
                ++numit;
                if (numit < kMaxIt1 && L.dlam12 > 0.0) {
                    const double dalp1 = -v / L.dlam12;
                    const double sdalp1 = sin(dalp1), cdalp1 = cos(dalp1);
                    const double nsalp1 = salp1 * cdalp1 + calp1 * sdalp1;
                    if (nsalp1 > 0.0 && fabs(dalp1) < kPi) {
                        const SC nn = norm2(nsalp1, calp1 * cdalp1 - salp1 * sdalp1);
                        salp1 = nn.s;
                        calp1 = nn.c;
                        tripn = fabs(v) <= 16.0 * kTol0;
                        continue;
                    }
                }
                const SC nn = norm2((salp1a + salp1b) / 2.0, (calp1a + calp1b) / 2.0);
                salp1 = nn.s;
                calp1 = nn.c;
                tripn = false;
                tripb = fabs(salp1a - salp1) + (calp1a - calp1) < kTolB || fabs(salp1 - salp1b) + (calp1 - calp1b) < kTolB;
            }
            salp2 = L.salp2;
            calp2 = L.calp2;
            double m12x;
            lengths(L.eps, L.sig12, L.ssig1, L.csig1, dn1, L.ssig2, L.csig2, dn2, s12x, m12x);
            s12x *= kWgsB;
        }
    }
    if (swapp < 0.0) {
        salp1 = salp2;
        calp1 = calp2;
    }
    salp1 *= swapp * lonsign;
    calp1 *= swapp * latsign;
    r.dist_km = (0.0 + s12x) * 1e-3;
    r.head_deg = floored_mod360(atan2d(salp1, calp1) + 360.0);
    return r;
}

struct PrepParams {
    int B, T, model;
    const int32_t* nobs;
    const double *lon, *lat, *gap;
    double *sog, *cog, *sog_rate, *cog_rate, *z;
    int32_t* status;
};

template <int kModel>
__device__ __forceinline__ Leg leg_of(const PrepParams& p, int j, int t) {
    const size_t a = (size_t)j * p.B + t, b = a + p.B;
    const double lon1 = p.lon[a], lat1 = p.lat[a], lon2 = p.lon[b], lat2 = p.lat[b];
    return kModel == 0 ? sphere_leg(lon1, lat1, lon2, lat2) : wgs84_leg(lon1, lat1, lon2, lat2);
}

// Observation i of a track with n observations uses leg j = min(i, n-2) (the last value is repeated,
// ship_track.py:220, :275); its rate is the backward difference against observation i-1 over gap[i-1], 0 for i = 0
// (ship_track.py:242-246, :296-300).  The previous observation's leg is recomputed here rather than read back, so the
// kernel is one pass with no ordering between threads; both evaluations run the same code and agree bit for bit.
template <int kModel>
__global__ void __launch_bounds__(256) track_prep(PrepParams p) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)p.T * p.B) return;
    const int t = (int)(idx % p.B), i = (int)(idx / p.B);
    const int n = p.nobs ? min(p.nobs[t], p.T) : p.T;
    double sog = 0.0, cog = 0.0, sr = 0.0, cr = 0.0;
    if (i < n && n >= 2) {
        const int j = min(i, n - 2);
        const Leg cur = leg_of<kModel>(p, j, t);
        if (kModel == 1 && !cur.converged && p.status) atomicOr(&p.status[t], STE_PREP_STATUS_NOCONV);
        sog = cur.dist_km / p.gap[(size_t)j * p.B + t];
        cog = cur.head_deg;
        if (i >= 1) {
            const int jp = min(i - 1, n - 2);
            double sog_p = sog, cog_p = cog;
            if (jp != j) {
                const Leg prev = leg_of<kModel>(p, jp, t);
                sog_p = prev.dist_km / p.gap[(size_t)jp * p.B + t];
                cog_p = prev.head_deg;
            }
            const double g = p.gap[(size_t)(i - 1) * p.B + t];
            sr = (sog - sog_p) / g;
            cr = (cog - cog_p) / g;
        }
    }
    p.sog[idx] = sog;
    p.cog[idx] = cog;
    p.sog_rate[idx] = sr;
    p.cog_rate[idx] = cr;
    if (p.z) {
        const size_t zb = ((size_t)i * 4) * p.B + t;
        const bool live = i < n;
        p.z[zb] = live ? p.lon[idx] : 0.0;
        p.z[zb + p.B] = live ? p.lat[idx] : 0.0;
        p.z[zb + 2 * (size_t)p.B] = sog;
        p.z[zb + 3 * (size_t)p.B] = cog;
    }
}

}  // namespace
}  // namespace ste

extern "C" int ste_track_prep_f64(const ste_prep_batch_f64* b, void* stream) {
    using namespace ste;
    if (!b) return abi_fail(STE_EINVAL, "prep batch pointer is NULL");
    if (b->B <= 0 || b->Tmax < 1) return abi_fail(STE_EINVAL, "B must be > 0 and Tmax >= 1");
    if (b->model != STE_PREP_SPHERE && b->model != STE_PREP_WGS84)
        return abi_fail(STE_EINVAL, "model must be STE_PREP_SPHERE or STE_PREP_WGS84");
    if (!b->lon || !b->lat || !b->sog || !b->cog || !b->sog_rate || !b->cog_rate)
        return abi_fail(STE_EINVAL, "lon, lat, sog, cog, sog_rate and cog_rate are required");
    if (b->Tmax > 1 && !b->gap) return abi_fail(STE_EINVAL, "gap is required when Tmax > 1");
    PrepParams p{b->B, b->Tmax, b->model, b->nobs, b->lon, b->lat, b->gap, b->sog, b->cog, b->sog_rate, b->cog_rate, b->z,
                 b->status};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (b->status) {
        int rc = abi_check_hip(hipMemsetAsync(b->status, 0, sizeof(int32_t) * (size_t)b->B, s), "track_prep status reset");
        if (rc) return rc;
    }
    const size_t total = (size_t)b->B * b->Tmax;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (b->model == STE_PREP_SPHERE)
        hipLaunchKernelGGL(track_prep<0>, dim3(grid), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(track_prep<1>, dim3(grid), dim3(256), 0, s, p);
    return abi_check_hip(hipGetLastError(), "track_prep launch");
}

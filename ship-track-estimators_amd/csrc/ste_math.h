// ste_math.h — per-lane fp64 building blocks of the UKF / URTSS kernels (gfx950).
//
// Everything here works on small fixed-size arrays whose indices are compile-time constants after unrolling, so the
// whole filter state lives in VGPRs (no LDS; scratch only on the rare out-of-line paths).  Reference semantics being reproduced are cited per function
// (paths relative to /root/reference/src/track_estimators/kalman_filters/).
#pragma once
#include <hip/hip_runtime.h>

namespace ste {

constexpr double kEarthRadius = 6378.137;              // constants.py:1
constexpr double kDeg2Rad = 0.017453292519943295;      // NumPy's NPY_PI / 180.0 (np.radians)
constexpr double kRad2Deg = 57.29577951308232;         // 180.0 / NPY_PI          (np.degrees)
constexpr double kPinvRcond = 1e-15;                   // np.linalg.pinv default rcond (unscented.py:243,333)
constexpr int kMaxSweeps = 12;
// A rotation is skipped when a_pq^2 <= kRotTol2 * |a_pp * a_qq| (relative, Demmel–Veselić style criterion):
// it would change neither diagonal entry in fp64.  An exactly zero a_pq is always skipped.
constexpr double kRotTol2 = 4.930380657631324e-32;     // (2^-52)^2

#define STE_UNROLL _Pragma("unroll")

// Filter constants shared by every track: kernel arguments, so they sit in SGPRs / the scalar cache.
struct Mats {
    double fan_scale, w0, wi;
    double H[16], Q[16], R[16];
    double chi_alpha;  // robust update threshold (unscented.py:357: 50)
    int robust_iters;  // 0 = robustification off (the reference's call site is commented out, unscented.py:228)
};

// NumPy floored modulo by 360 (unscented.py:250,257,340,346): npy_divmod takes fmod(a, b) (exact) and, when that is
// negative, adds b once (one rounding); an exact zero comes out as +0.  Here: q = floor(a/360) estimated with one
// multiply, r = fma(-q, 360, a) -- the product is exact and the difference is representable, so r IS a - 360 q -- and if
// the estimate was off by one the remainder is recomputed with the neighbouring q.  The result equals NumPy's bit for
// bit (for a >= 0 it is the exact remainder; for a < 0 it is the correctly rounded fmod(a, 360) + 360).
__device__ __forceinline__ double floored_mod360(double a) {
    if (__builtin_expect(!(fabs(a) < 1e15), 0)) {  // q * 360 would no longer be exact; also inf / NaN
        double r = fmod(a, 360.0);
        if (r != 0.0) {
            if (r < 0.0) r += 360.0;
        } else {
            r = 0.0;
        }
        return r;
    }
    double q = floor(a * 2.7777777777777778e-03);
    double r = fma(-q, 360.0, a);
    if (r < 0.0) {
        q -= 1.0;
        r = fma(-q, 360.0, a);
    } else if (r >= 360.0) {
        q += 1.0;
        r = fma(-q, 360.0, a);
    }
    return r;
}
__device__ __forceinline__ double floored_mod(double a, double b) {
    (void)b;  // every call site of the reference uses 360
    return floored_mod360(a);
}

// (y + 180) % 360 - 180   (unscented.py:250, :340)
__device__ __forceinline__ double wrap180(double y) { return floored_mod(y + 180.0, 360.0) - 180.0; }

// 1/sqrt(x) for normal positive x: v_rsq_f64 (about 2^-24 relative) plus one third-order correction
// y += y*e*(1/2 + 3/8 e), e = 1 - x*y^2, which leaves ~2^-70 before the final rounding.  No division, no v_sqrt.
__device__ __forceinline__ double rsqrt_fast(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

// ---------------------------------------------------------------------------------------------------------------
// Transcendentals.  The device library's sincos/atan2/asin cost 40/34/14-42 fp64-FMA issue slots each (measured on
// gfx950, one wave per SIMD) and the fan needs 27/9/9 of them per step, so the common cases are inlined here with
// fdlibm's kernels (same minimax polynomials, < 1 ulp on their intervals) and everything unusual drops to the library.
// ---------------------------------------------------------------------------------------------------------------

// sin and cos on |r| <= pi/4 (fdlibm __kernel_sin / __kernel_cos polynomials).
__device__ __forceinline__ void sincos_kernel(double r, double& s, double& c) {
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    s = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    c = fma(z * z, pc, fma(-0.5, z, 1.0));
}

// sincos for |x| < 2^20 by two-constant Cody–Waite reduction with FMA (n*pi/2 is removed exactly in the first FMA,
// the second restores the bits of pi/2 beyond double precision); larger arguments use the library's Payne–Hanek path.
__device__ __forceinline__ void sincos_fast(double x, double& s, double& c) {
    if (__builtin_expect(!(fabs(x) < 1048576.0), 0)) {
        sincos(x, &s, &c);
        return;
    }
    const double n = rint(x * 0.63661977236758134308);  // 2/pi
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    double sk, ck;
    sincos_kernel(r, sk, ck);
    const int q = (int)n;
    const double sv = (q & 1) ? ck : sk;
    const double cv = (q & 1) ? sk : ck;
    s = (q & 2) ? -sv : sv;
    c = ((q + 1) & 2) ? -cv : cv;
}

// atan on |q| <= 7/16 (fdlibm atan's polynomial for its first interval).
__device__ __forceinline__ double atan_small(double q) {
    const double z = q * q, w = z * z;
    double s1 = fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02);
    s1 = fma(w, s1, 6.66107313738753120669e-02);
    s1 = fma(w, s1, 9.09088713343650656196e-02);
    s1 = fma(w, s1, 1.42857142725034663711e-01);
    s1 = fma(w, s1, 3.33333333333329318027e-01);
    double s2 = fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02);
    s2 = fma(w, s2, -7.69187620504482999495e-02);
    s2 = fma(w, s2, -1.11111104054623557880e-01);
    s2 = fma(w, s2, -1.99999999998764832476e-01);
    return fma(-q, fma(z, s1, w * s2), q);
}

// a / b for normal b > 0: v_rcp_f64 (~2^-24) refined once (~2^-48), then one residual correction of the quotient -- the
// residual a - b q is exact in FMA arithmetic, so the corrected quotient is off by ~2^-48 of the first error: within an
// ulp.  (Round 2 refined twice; the second step bought nothing the correction does not.)
__device__ __forceinline__ double div_pos(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// dt / kEarthRadius without the division sequence: product with the rounded reciprocal plus one residual correction
// (the residual is exact in FMA arithmetic, so the result is the correctly rounded quotient except in rare ties).
__device__ __forceinline__ double div_earth_radius(double a) {
    constexpr double kInvR = 1.0 / kEarthRadius;
    const double q = a * kInvR;
    return fma(fma(-q, kEarthRadius, a), kInvR, q);
}

// atan2(a, b): a ship moves a small angle per step, so b (= cos(lat') cos(dlon)) is positive and |a| << b almost always.
__device__ __forceinline__ double atan2_fast(double a, double b) {
    if (__builtin_expect(b > 1e-300 && fabs(a) <= 0.4375 * b, 1)) return atan_small(div_pos(a, b));
    return atan2(a, b);
}

// asin on |x| <= 1/2 (fdlibm asin's rational approximation for that interval).
__device__ __forceinline__ double asin_small(double x) {
    const double t = x * x;
    double pn = fma(t, 3.47933107596021167570e-05, 7.91534994289814532176e-04);
    pn = fma(t, pn, -4.00555345006794114027e-02);
    pn = fma(t, pn, 2.01212532134862925881e-01);
    pn = fma(t, pn, -3.25565818622400915405e-01);
    pn = fma(t, pn, 1.66666666666666657415e-01);
    pn *= t;
    double qd = fma(t, 7.70381505559019352791e-02, -6.88283971605453293030e-01);
    qd = fma(t, qd, 2.02094576023350569471e+00);
    qd = fma(t, qd, -2.40339491173441421878e+00);
    qd = fma(t, qd, 1.0);
    return fma(x, div_pos(pn, qd), x);
}

// The part of the great-circle step after the three sin/cos pairs are known (non_linear_process.py:64-72).
// (lon_r, lat_r) are the point's longitude/latitude in radians; returns lon', lat' in degrees.
//
// Latitude: the reference takes asin(sin(lat')).  Here lat' = lat + asin(sin(lat' - lat)) with
// sin(lat' - lat) = sin(lat') cos(lat) - cos(lat') sin(lat) and cos(lat') = hypot(a, b) (a = cos(lat') sin(dlon),
// b = cos(lat') cos(dlon) are the atan2 operands).  The argument is then of the size of the step, inside asin's
// polynomial interval, and the result keeps its accuracy near the poles.  Steps over 30 degrees use asin(sin(lat')).
// The identity needs |lat' - lat| <= 90 degrees, which |sin(lat' - lat)| <= 1/2 alone does not say (sin 150 = 1/2): the fast
// path also asks for cos(delta) > 0 -- the arc of the step under 90 degrees, and a latitude cannot change by more than the
// arc.  (Round 4: ship WGAE of data/modern_ships, filtered at -2 400 km/h over a 41-hour gap, goes 2.5 times round the
// globe in one step; the reference lands on 67.0 N, the unguarded identity on 33.1 S.)
__device__ __forceinline__ void geodetic_finish(double lon_r, double lat_r, double sp, double cp, double sa, double ca,
                                                double sd, double cd, double& lon_out, double& lat_out) {
    const double a = sd * sa;
    const double sdca = sd * ca;
    const double b = fma(cp, cd, -(sp * sdca));
    const double sl = fma(sp, cd, cp * sdca);  // sin(lat')
    lon_out = (lon_r + atan2_fast(a, b)) * kRad2Deg;
    const double h2 = fma(a, a, b * b);
    const double cl = h2 * rsqrt_fast(h2);  // cos(lat') >= 0
    const double xs = fma(sl, cp, -(cl * sp));
    double lat2;
    if (__builtin_expect(fabs(xs) <= 0.5 && h2 > 1e-300 && cd > 0.0, 1)) {
        lat2 = lat_r + asin_small(xs);
    } else {
        lat2 = asin(sl);
    }
    lat_out = lat2 * kRad2Deg;
}

// sincos of a small increment: the kernels directly when |d| <= pi/4 (always, for a sigma-point deviation of a tracked
// ship), the reducing version otherwise.
__device__ __forceinline__ void sincos_delta(double d, double& s, double& c) {
    if (__builtin_expect(fabs(d) <= 0.78539816339744828, 1)) {
        sincos_kernel(d, s, c);
    } else {
        sincos_fast(d, s, c);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Three-at-a-time forms for the quad kernels (centre point and the +/- pair of one sigma direction).  A lone wave issues
// one instruction every ~4.3 cycles whatever its type, and every fp64 polynomial coefficient has to be brought into a
// register before its FMA (two s_mov_b32 or two v_accvgpr_read: there are more coefficients than registers to keep them
// in), so evaluating the same polynomial for three arguments side by side fetches each coefficient once instead of three
// times.  They are branch-free: `ok` collects the conditions under which the fast paths are valid and the caller redoes
// the rare failing case with the branching scalar functions above (same formulas, so lanes that were fine get the same
// bits either way).
// ---------------------------------------------------------------------------------------------------------------
// Where the sin / cos kernel polynomials take their coefficients from.  As literals (TrigLit) every use needs the 64-bit
// constant in a register pair first -- two s_mov_b32, or, once the kernel has more live scalars than SGPRs (the
// lane-per-track forward kernel: ~33 polynomial coefficients on top of its arguments), two v_readlane_b32 from a spilled
// copy: 83 such reads per step.  TrigReg holds the twelve coefficients of the most used polynomials in VGPRs for the
// whole kernel instead (the lane-per-track kernel has the registers to spare below the 264 it is padded to).
struct TrigLit {
    static constexpr double s1 = 1.58969099521155010221e-10, s2 = -2.50507602534068634195e-08, s3 = 2.75573137070700676789e-06,
                            s4 = -1.98412698298579493134e-04, s5 = 8.33333333332248946124e-03, s6 = -1.66666666666666324348e-01;
    static constexpr double c1 = -1.13596475577881948265e-11, c2 = 2.08757232129817482790e-09, c3 = -2.75573143513906633035e-07,
                            c4 = 2.48015872894767294178e-05, c5 = -1.38888888888741095749e-03, c6 = 4.16666666666666019037e-02;
};
struct TrigReg {
    double s1, s2, s3, s4, s5, s6, c1, c2, c3, c4, c5, c6;
};
__device__ __forceinline__ double in_vgpr(double v) {  // opaque to the optimiser: the value lives in a VGPR from here on
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ void trig_reg_init(TrigLit&) {}
__device__ __forceinline__ void trig_reg_init(TrigReg& k) {
    k.s1 = in_vgpr(TrigLit::s1);
    k.s2 = in_vgpr(TrigLit::s2);
    k.s3 = in_vgpr(TrigLit::s3);
    k.s4 = in_vgpr(TrigLit::s4);
    k.s5 = in_vgpr(TrigLit::s5);
    k.s6 = in_vgpr(TrigLit::s6);
    k.c1 = in_vgpr(TrigLit::c1);
    k.c2 = in_vgpr(TrigLit::c2);
    k.c3 = in_vgpr(TrigLit::c3);
    k.c4 = in_vgpr(TrigLit::c4);
    k.c5 = in_vgpr(TrigLit::c5);
    k.c6 = in_vgpr(TrigLit::c6);
}

template <int N, class K = TrigLit>
__device__ __forceinline__ void sincos_kernel_n(const double (&r)[N], double (&s)[N], double (&c)[N], const K& k = K()) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    double z[N], ps[N], pc[N];
    STE_UNROLL
    for (int i = 0; i < N; ++i) z[i] = r[i] * r[i];
    STE_UNROLL
    for (int i = 0; i < N; ++i) ps[i] = fma(z[i], k.s1, k.s2);
    STE_UNROLL
    for (int i = 0; i < N; ++i) ps[i] = fma(z[i], ps[i], k.s3);
    STE_UNROLL
    for (int i = 0; i < N; ++i) ps[i] = fma(z[i], ps[i], k.s4);
    STE_UNROLL
    for (int i = 0; i < N; ++i) ps[i] = fma(z[i], ps[i], k.s5);
    STE_UNROLL
    for (int i = 0; i < N; ++i) ps[i] = fma(z[i], ps[i], k.s6);
    STE_UNROLL
    for (int i = 0; i < N; ++i) s[i] = fma(z[i] * r[i], ps[i], r[i]);
    STE_UNROLL
    for (int i = 0; i < N; ++i) pc[i] = fma(z[i], k.c1, k.c2);
    STE_UNROLL
    for (int i = 0; i < N; ++i) pc[i] = fma(z[i], pc[i], k.c3);
    STE_UNROLL
    for (int i = 0; i < N; ++i) pc[i] = fma(z[i], pc[i], k.c4);
    STE_UNROLL
    for (int i = 0; i < N; ++i) pc[i] = fma(z[i], pc[i], k.c5);
    STE_UNROLL
    for (int i = 0; i < N; ++i) pc[i] = fma(z[i], pc[i], k.c6);
    STE_UNROLL
    for (int i = 0; i < N; ++i) c[i] = fma(z[i] * z[i], pc[i], fma(-0.5, z[i], 1.0));
}

// sincos_fast for N arguments; ok &= every |x| < 2^20
template <int N, class K = TrigLit>
__device__ __forceinline__ void sincos_fast_n(const double (&x)[N], double (&s)[N], double (&c)[N], bool& ok,
                                              const K& k = K()) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    double n[N], r[N], sk[N], ck[N];
    STE_UNROLL
    for (int i = 0; i < N; ++i) {
        ok = ok && (fabs(x[i]) < 1048576.0);
        n[i] = rint(x[i] * 0.63661977236758134308);
    }
    STE_UNROLL
    for (int i = 0; i < N; ++i) r[i] = fma(-n[i], 1.57079632679489655800e+00, x[i]);
    STE_UNROLL
    for (int i = 0; i < N; ++i) r[i] = fma(-n[i], 6.12323399573676603587e-17, r[i]);
    sincos_kernel_n<N, K>(r, sk, ck, k);
    STE_UNROLL
    for (int i = 0; i < N; ++i) {
        const int q = (int)n[i];
        const double sv = (q & 1) ? ck[i] : sk[i];
        const double cv = (q & 1) ? sk[i] : ck[i];
        s[i] = (q & 2) ? -sv : sv;
        c[i] = ((q + 1) & 2) ? -cv : cv;
    }
}

// sincos_delta for N arguments; ok &= every |d| <= pi/4
template <int N, class K = TrigLit>
__device__ __forceinline__ void sincos_delta_n(const double (&d)[N], double (&s)[N], double (&c)[N], bool& ok,
                                               const K& k = K()) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    STE_UNROLL
    for (int i = 0; i < N; ++i) ok = ok && (fabs(d[i]) <= 0.78539816339744828);
    sincos_kernel_n<N, K>(d, s, c, k);
}

// The same choice for the arctangent and arcsine polynomials of geodetic_finish_n (GeoLit: literals; GeoReg: VGPRs for the
// whole kernel -- the quad kernels, whose lone wave pays an issue slot for every s_mov_b32 and has registers to spare).
struct GeoLit {
    static constexpr double t0 = 1.62858201153657823623e-02, t1 = 4.97687799461593236017e-02, t2 = 6.66107313738753120669e-02,
                            t3 = 9.09088713343650656196e-02, t4 = 1.42857142725034663711e-01, t5 = 3.33333333333329318027e-01,
                            u0 = -3.65315727442169155270e-02, u1 = -5.83357013379057348645e-02, u2 = -7.69187620504482999495e-02,
                            u3 = -1.11111104054623557880e-01, u4 = -1.99999999998764832476e-01;
    static constexpr double p0 = 3.47933107596021167570e-05, p1 = 7.91534994289814532176e-04, p2 = -4.00555345006794114027e-02,
                            p3 = 2.01212532134862925881e-01, p4 = -3.25565818622400915405e-01, p5 = 1.66666666666666657415e-01,
                            q0 = 7.70381505559019352791e-02, q1 = -6.88283971605453293030e-01, q2 = 2.02094576023350569471e+00,
                            q3 = -2.40339491173441421878e+00;
};
struct GeoReg {
    double t0, t1, t2, t3, t4, t5, u0, u1, u2, u3, u4;
    static constexpr double p0 = GeoLit::p0, p1 = GeoLit::p1, p2 = GeoLit::p2, p3 = GeoLit::p3, p4 = GeoLit::p4, p5 = GeoLit::p5,
                            q0 = GeoLit::q0, q1 = GeoLit::q1, q2 = GeoLit::q2, q3 = GeoLit::q3;
};
__device__ __forceinline__ void geo_reg_init(GeoLit&) {}
__device__ __forceinline__ void geo_reg_init(GeoReg& k) {
    k.t0 = in_vgpr(GeoLit::t0);
    k.t1 = in_vgpr(GeoLit::t1);
    k.t2 = in_vgpr(GeoLit::t2);
    k.t3 = in_vgpr(GeoLit::t3);
    k.t4 = in_vgpr(GeoLit::t4);
    k.t5 = in_vgpr(GeoLit::t5);
    k.u0 = in_vgpr(GeoLit::u0);
    k.u1 = in_vgpr(GeoLit::u1);
    k.u2 = in_vgpr(GeoLit::u2);
    k.u3 = in_vgpr(GeoLit::u3);
    k.u4 = in_vgpr(GeoLit::u4);
}

template <int N, class K = GeoLit>
__device__ __forceinline__ void atan_small_n(const double (&q)[N], double (&out)[N], const K& k = K()) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    double z[N], w[N], s1[N], s2[N];
    STE_UNROLL
    for (int i = 0; i < N; ++i) {
        z[i] = q[i] * q[i];
        w[i] = z[i] * z[i];
    }
    STE_UNROLL
    for (int i = 0; i < N; ++i) s1[i] = fma(w[i], k.t0, k.t1);
    STE_UNROLL
    for (int i = 0; i < N; ++i) s1[i] = fma(w[i], s1[i], k.t2);
    STE_UNROLL
    for (int i = 0; i < N; ++i) s1[i] = fma(w[i], s1[i], k.t3);
    STE_UNROLL
    for (int i = 0; i < N; ++i) s1[i] = fma(w[i], s1[i], k.t4);
    STE_UNROLL
    for (int i = 0; i < N; ++i) s1[i] = fma(w[i], s1[i], k.t5);
    STE_UNROLL
    for (int i = 0; i < N; ++i) s2[i] = fma(w[i], k.u0, k.u1);
    STE_UNROLL
    for (int i = 0; i < N; ++i) s2[i] = fma(w[i], s2[i], k.u2);
    STE_UNROLL
    for (int i = 0; i < N; ++i) s2[i] = fma(w[i], s2[i], k.u3);
    STE_UNROLL
    for (int i = 0; i < N; ++i) s2[i] = fma(w[i], s2[i], k.u4);
    STE_UNROLL
    for (int i = 0; i < N; ++i) out[i] = fma(-q[i], fma(z[i], s1[i], w[i] * s2[i]), q[i]);
}

template <int N, class K = GeoLit>
__device__ __forceinline__ void asin_small_n(const double (&x)[N], double (&out)[N], const K& k = K()) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    double t[N], pn[N], qd[N];
    STE_UNROLL
    for (int i = 0; i < N; ++i) t[i] = x[i] * x[i];
    STE_UNROLL
    for (int i = 0; i < N; ++i) pn[i] = fma(t[i], k.p0, k.p1);
    STE_UNROLL
    for (int i = 0; i < N; ++i) pn[i] = fma(t[i], pn[i], k.p2);
    STE_UNROLL
    for (int i = 0; i < N; ++i) pn[i] = fma(t[i], pn[i], k.p3);
    STE_UNROLL
    for (int i = 0; i < N; ++i) pn[i] = fma(t[i], pn[i], k.p4);
    STE_UNROLL
    for (int i = 0; i < N; ++i) pn[i] = fma(t[i], pn[i], k.p5);
    STE_UNROLL
    for (int i = 0; i < N; ++i) pn[i] *= t[i];
    STE_UNROLL
    for (int i = 0; i < N; ++i) qd[i] = fma(t[i], k.q0, k.q1);
    STE_UNROLL
    for (int i = 0; i < N; ++i) qd[i] = fma(t[i], qd[i], k.q2);
    STE_UNROLL
    for (int i = 0; i < N; ++i) qd[i] = fma(t[i], qd[i], k.q3);
    STE_UNROLL
    for (int i = 0; i < N; ++i) qd[i] = fma(t[i], qd[i], 1.0);
    STE_UNROLL
    for (int i = 0; i < N; ++i) out[i] = fma(x[i], div_pos(pn[i], qd[i]), x[i]);
}

// geodetic_finish for N points; ok &= the fast atan2 / asin paths apply to all of them
template <int N, class K = GeoLit>
__device__ __forceinline__ void geodetic_finish_n(const double (&lon_r)[N], const double (&lat_r)[N], const double (&sp)[N],
                                                  const double (&cp)[N], const double (&sa)[N], const double (&ca)[N],
                                                  const double (&sd)[N], const double (&cd)[N], double (&lon_out)[N],
                                                  double (&lat_out)[N], bool& ok, const K& k = K()) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    double a[N], b[N], sl[N], h2[N], xs[N], qa[N], at[N], as[N];
    STE_UNROLL
    for (int i = 0; i < N; ++i) {
        a[i] = sd[i] * sa[i];
        const double sdca = sd[i] * ca[i];
        b[i] = fma(cp[i], cd[i], -(sp[i] * sdca));
        sl[i] = fma(sp[i], cd[i], cp[i] * sdca);
        ok = ok && (b[i] > 1e-300) && (fabs(a[i]) <= 0.4375 * b[i]);
        h2[i] = fma(a[i], a[i], b[i] * b[i]);
    }
    STE_UNROLL
    for (int i = 0; i < N; ++i) qa[i] = div_pos(a[i], b[i]);
    atan_small_n<N, K>(qa, at, k);
    STE_UNROLL
    for (int i = 0; i < N; ++i) {
        const double cl = h2[i] * rsqrt_fast(h2[i]);
        xs[i] = fma(sl[i], cp[i], -(cl * sp[i]));
        ok = ok && (fabs(xs[i]) <= 0.5) && (h2[i] > 1e-300) && (cd[i] > 0.0);  // cd > 0: see geodetic_finish
    }
    asin_small_n<N, K>(xs, as, k);
    STE_UNROLL
    for (int i = 0; i < N; ++i) {
        lon_out[i] = (lon_r[i] + at[i]) * kRad2Deg;
        lat_out[i] = (lat_r[i] + as[i]) * kRad2Deg;
    }
}

// Great-circle dead reckoning of one state (non_linear_process.py:46-85, c = None).
__device__ __forceinline__ void geodetic_step(const double (&x)[4], double dt, double sog_rate, double cog_rate,
                                              double (&out)[4]) {
    const double lon = x[0] * kDeg2Rad;
    const double lat = x[1] * kDeg2Rad;
    const double u = x[2];
    const double alpha = x[3] * kDeg2Rad;
    const double udt_r = u * dt / kEarthRadius;
    double sd, cd, sa, ca, sp, cp;
    sincos_fast(udt_r, sd, cd);
    sincos_fast(alpha, sa, ca);
    sincos_fast(lat, sp, cp);
    geodetic_finish(lon, lat, sp, cp, sa, ca, sd, cd, out[0], out[1]);
    out[2] = u + sog_rate * dt;
    out[3] = alpha * kRad2Deg + cog_rate * dt;
}

// One Jacobi rotation on the (P,Q) plane of the symmetric A (both triangles kept), accumulating into V.
// With delta = a_qq - a_pp and h = sqrt(delta^2 + 4 a_pq^2) the rotation that annihilates a_pq has
//   cos^2 = (1 + |delta|/h)/2,   sin = +-a_pq / (h cos),   tan = sin / cos      (sign of tan = sign(delta) sign(a_pq))
// which needs two reciprocal square roots and no division.
template <int P, int Q>
__device__ __forceinline__ bool jacobi_rot(double (&A)[4][4], double (&V)[4][4]) {
    const double apq = A[P][Q];
    const double app = A[P][P], aqq = A[Q][Q];
    const bool go = apq * apq > kRotTol2 * fabs(app * aqq);  // false for NaN and for apq == 0
    if (go) {
        const double delta = aqq - app;
        const double two_apq = apq + apq;
        const double rh = rsqrt_fast(fma(delta, delta, two_apq * two_apq));
        const double c2 = fma(0.5 * fabs(delta), rh, 0.5);  // in [1/2, 1]
        const double rc = rsqrt_fast(c2);                   // 1 / cos
        const double c = c2 * rc;
        const double s = (delta < 0.0 ? -apq : apq) * rh * rc;
        const double t = s * rc;
        A[P][P] = fma(-t, apq, app);
        A[Q][Q] = fma(t, apq, aqq);
        A[P][Q] = 0.0;
        A[Q][P] = 0.0;
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            if (r != P && r != Q) {
                const double arp = A[r][P], arq = A[r][Q];
                const double np_ = fma(c, arp, -(s * arq));
                const double nq_ = fma(s, arp, c * arq);
                A[r][P] = np_;
                A[P][r] = np_;
                A[r][Q] = nq_;
                A[Q][r] = nq_;
            }
        }
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            const double vrp = V[r][P], vrq = V[r][Q];
            V[r][P] = fma(c, vrp, -(s * vrq));
            V[r][Q] = fma(s, vrp, c * vrq);
        }
    }
    return go;
}

// Cyclic Jacobi sweeps on the symmetric A, accumulating rotations into V (which the caller initialised).
// Returns false if the sweep cap was hit.  The wave leaves the loop together (__any), but each lane's rotations are
// gated by its own data only, so a track's result does not depend on which tracks share its wave.
__device__ __forceinline__ bool jacobi_sweeps(double (&A)[4][4], double (&V)[4][4], double (&w)[4]) {
    bool rotated = true;
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        rotated = jacobi_rot<0, 1>(A, V);
        rotated |= jacobi_rot<2, 3>(A, V);
        rotated |= jacobi_rot<0, 2>(A, V);
        rotated |= jacobi_rot<1, 3>(A, V);
        rotated |= jacobi_rot<0, 3>(A, V);
        rotated |= jacobi_rot<1, 2>(A, V);
        if (!__any(rotated)) break;
    }
    STE_UNROLL
    for (int i = 0; i < 4; ++i) w[i] = A[i][i];
    return !rotated;
}

// Cold start: A = V diag(w) V^T from V = I.  A is destroyed.
__device__ __forceinline__ bool jacobi_eig4(double (&A)[4][4], double (&V)[4][4], double (&w)[4]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) V[r][c] = (r == c) ? 1.0 : 0.0;
    }
    return jacobi_sweeps(A, V, w);
}

// Warm start: V holds the eigenvectors of a nearby matrix (the same quantity one filter step earlier).  A is first
// moved into that basis, B = V^T A V, which is nearly diagonal, so one or two sweeps finish the job; the rotations keep
// accumulating into V.  Same fixed point as the cold start, to rounding.
__device__ __forceinline__ bool jacobi_eig4_warm(double (&A)[4][4], double (&V)[4][4], double (&w)[4]) {
    double M[4][4], Bm[4][4];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            double acc = A[r][0] * V[0][c];
            STE_UNROLL
            for (int i = 1; i < 4; ++i) acc = fma(A[r][i], V[i][c], acc);
            M[r][c] = acc;
        }
    }
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) {
            double acc = V[0][r] * M[0][c];
            STE_UNROLL
            for (int i = 1; i < 4; ++i) acc = fma(V[i][r], M[i][c], acc);
            Bm[r][c] = acc;
            Bm[c][r] = acc;
        }
    }
    return jacobi_sweeps(Bm, V, w);
}

// Eigenvector basis carried from one filter step to the next for one recurring symmetric matrix.
struct EigBasis {
    double V[4][4];
    bool valid;
};

// out = V diag(f) V^T (symmetric, upper triangle computed and mirrored).
__device__ __forceinline__ void recompose(const double (&V)[4][4], const double (&f)[4], double (&out)[4][4]) {
    double Vf[4][4];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int i = 0; i < 4; ++i) Vf[r][i] = V[r][i] * f[i];
    }
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) {
            double acc = Vf[r][0] * V[c][0];
            STE_UNROLL
            for (int i = 1; i < 4; ++i) acc = fma(Vf[r][i], V[c][i], acc);
            out[r][c] = acc;
            out[c][r] = acc;
        }
    }
}

// Principal square root of (scale * P) with negative eigenvalues clamped: the real part of scipy.linalg.sqrtm on a
// symmetric matrix (unscented.py:95-97).  P is symmetrised first.  Returns status bits.
template <bool kWarm>
__device__ __forceinline__ int sym_sqrt4(const double (&P)[4][4], double scale, double (&T)[4][4], EigBasis& basis) {
    double A[4][4], w[4], f[4];
    double (&V)[4][4] = basis.V;
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) {
            const double v = scale * (0.5 * (P[r][c] + P[c][r]));
            A[r][c] = v;
            A[c][r] = v;
        }
    }
    int st = 0;
    double wmax = 0.0;
    STE_UNROLL
    for (int i = 0; i < 4; ++i) wmax = fmax(wmax, fabs(A[i][i]));
    bool ok;
    if (kWarm && basis.valid) {
        ok = jacobi_eig4_warm(A, V, w);
    } else {
        ok = jacobi_eig4(A, V, w);
    }
    if (!ok) st |= 0x4;
    STE_UNROLL
    for (int i = 0; i < 4; ++i) {
        if (w[i] < -1e-12 * wmax) st |= 0x2;
        // sqrt(max(w, 0)) as w * rsqrt(w): a third of the instructions of the correctly rounded sqrt sequence, ~1 ulp
        f[i] = w[i] > 0.0 ? w[i] * rsqrt_fast(w[i]) : 0.0;
    }
    recompose(V, f, T);
    return st;
}

// Moore–Penrose pseudo-inverse of a symmetric 4x4 with NumPy's cutoff: singular values (= |eigenvalues|) not larger
// than rcond * max are dropped (np.linalg.pinv as called at unscented.py:243 and :333).
template <bool kWarm>
__device__ __forceinline__ int sym_pinv4(const double (&S)[4][4], double (&Si)[4][4], EigBasis& basis) {
    double A[4][4], w[4], f[4];
    double (&V)[4][4] = basis.V;
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) {
            const double v = 0.5 * (S[r][c] + S[c][r]);
            A[r][c] = v;
            A[c][r] = v;
        }
    }
    int st = 0;
    bool ok;
    if (kWarm && basis.valid) {
        ok = jacobi_eig4_warm(A, V, w);
    } else {
        ok = jacobi_eig4(A, V, w);
    }
    if (!ok) st |= 0x4;
    double smax = 0.0;
    STE_UNROLL
    for (int i = 0; i < 4; ++i) smax = fmax(smax, fabs(w[i]));
    const double cutoff = kPinvRcond * smax;
    STE_UNROLL
    for (int i = 0; i < 4; ++i) f[i] = (fabs(w[i]) > cutoff) ? 1.0 / w[i] : 0.0;
    recompose(V, f, Si);
    return st;
}

__device__ __forceinline__ int sym_pinv4(const double (&S)[4][4], double (&Si)[4][4]) {
    EigBasis none;
    return sym_pinv4<false>(S, Si, none);
}

// The same for an S that is zero outside its leading 2 x 2 block -- S = H P H^T + R with an H that observes two components
// (the reference's H = diag(1, 1, 0, 0)): its eigen-decomposition is ONE exact rotation, with the formulas of jacobi_rot,
// instead of sweeps over a matrix that is three quarters zeros.  The caller checks the structure (wave-uniformly).
__device__ __forceinline__ void sym_pinv4_block2(const double (&S)[4][4], double (&Si)[4][4]) {
    const double a = S[0][0], d = S[1][1], b = 0.5 * (S[0][1] + S[1][0]);
    const bool go = b * b > kRotTol2 * fabs(a * d);
    const double delta = d - a, two_b = b + b;
    const double rh = rsqrt_fast(go ? fma(delta, delta, two_b * two_b) : 1.0);
    const double c2 = fma(0.5 * fabs(delta), rh, 0.5);
    const double rc = rsqrt_fast(c2);
    const double c = go ? c2 * rc : 1.0;
    const double sn = go ? (delta < 0.0 ? -b : b) * rh * rc : 0.0;
    const double tb = go ? sn * rc * b : 0.0;
    const double w0 = a - tb, w1 = d + tb;
    const double cutoff = kPinvRcond * fmax(fabs(w0), fabs(w1));
    const double f0 = (fabs(w0) > cutoff) ? 1.0 / w0 : 0.0, f1 = (fabs(w1) > cutoff) ? 1.0 / w1 : 0.0;
    const double cc = c * c, ss = sn * sn, cs = c * sn;  // pinv = V diag(f) V^T with V = [[c, s], [-s, c]]
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int q = 0; q < 4; ++q) Si[r][q] = 0.0;
    }
    Si[0][0] = fma(cc, f0, ss * f1);
    Si[1][1] = fma(ss, f0, cc * f1);
    Si[0][1] = cs * (f1 - f0);
    Si[1][0] = Si[0][1];
}

// 1/d for normal d of either sign: v_rcp_f64 (~2^-24) and two Newton steps (below 2^-80 before the final rounding).
__device__ __forceinline__ double rcp_refined(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    return fma(fma(-d, r, 1.0), r, r);
}

// K = D A^-1 for a symmetric 4x4 A given by its upper triangle au[10] (row-major: 00 01 02 03 11 12 13 22 23 33), by an
// unpivoted L Dg L^T factorisation and one pair of triangular solves per row of D: ~110 fp64 instructions against the
// ~1 500 of a cold Jacobi pseudo-inverse.  np.linalg.pinv (unscented.py:333) IS the inverse whenever no singular value
// falls under 1e-15 of the largest, and both routes are backward stable, so they agree to ~cond(A) * 2^-52.  Returns
// true ("bad") when a pivot is not safely away from zero (or anything is NaN): the caller then takes the eigenvalue
// route, which also reproduces pinv's rank decisions.
// Pivot threshold: with pivots no smaller than 1e-7 of the largest diagonal entry cond(A) stays under ~1e7 and the gain
// agrees with pinv's to ~1e-9, three decades inside the 1e-6 parity bound (round 2 used 1e-9: 2e-7 at worst, too close).
constexpr double kLdlPivotTol = 1e-7;
// The factorisation: L (unit lower, l10 l20 l30 l21 l31 l32), the reciprocal pivots, and the "bad" verdict.
struct Ldl4 {
    double l10, l20, l30, l21, l31, l32, i0, i1, i2, i3;
    bool bad;
};
__device__ __forceinline__ void ldl_factor4(const double (&au)[10], Ldl4& f) {
    const double a00 = au[0], a01 = au[1], a02 = au[2], a03 = au[3], a11 = au[4], a12 = au[5], a13 = au[6], a22 = au[7],
                 a23 = au[8], a33 = au[9];
    const double scale = fmax(fmax(fabs(a00), fabs(a11)), fmax(fabs(a22), fabs(a33)));
    const double d0 = a00;
    f.i0 = rcp_refined(d0);
    f.l10 = a01 * f.i0;
    f.l20 = a02 * f.i0;
    f.l30 = a03 * f.i0;
    const double d1 = fma(-f.l10, a01, a11);
    f.i1 = rcp_refined(d1);
    const double t21 = fma(-f.l20, a01, a12), t31 = fma(-f.l30, a01, a13);
    f.l21 = t21 * f.i1;
    f.l31 = t31 * f.i1;
    const double d2 = fma(-f.l21, t21, fma(-f.l20, a02, a22));
    f.i2 = rcp_refined(d2);
    const double t32 = fma(-f.l31, t21, fma(-f.l30, a02, a23));
    f.l32 = t32 * f.i2;
    const double d3 = fma(-f.l32, t32, fma(-f.l31, t31, fma(-f.l30, a03, a33)));
    f.i3 = rcp_refined(d3);
    const double dmin = fmin(fmin(fabs(d0), fabs(d1)), fmin(fabs(d2), fabs(d3)));
    f.bad = !(dmin > kLdlPivotTol * scale);  // also true for NaN anywhere
}
// k = (row of D) A^-1, i.e. A k^T = (row of D)^T
__device__ __forceinline__ void ldl_solve_row4(const Ldl4& f, const double (&d)[4], double (&k)[4]) {
    const double y0 = d[0];
    const double y1 = fma(-f.l10, y0, d[1]);
    const double y2 = fma(-f.l21, y1, fma(-f.l20, y0, d[2]));
    const double y3 = fma(-f.l32, y2, fma(-f.l31, y1, fma(-f.l30, y0, d[3])));
    const double x3 = y3 * f.i3;
    const double x2 = fma(-f.l32, x3, y2 * f.i2);
    const double x1 = fma(-f.l31, x3, fma(-f.l21, x2, y1 * f.i1));
    const double x0 = fma(-f.l30, x3, fma(-f.l20, x2, fma(-f.l10, x1, y0 * f.i0)));
    k[0] = x0;
    k[1] = x1;
    k[2] = x2;
    k[3] = x3;
}
__device__ __forceinline__ bool ldl_right_solve4(const double (&au)[10], const double (&D)[4][4], double (&K)[4][4]) {
    Ldl4 f;
    ldl_factor4(au, f);
    STE_UNROLL
    for (int r = 0; r < 4; ++r) ldl_solve_row4(f, D[r], K[r]);
    return f.bad;
}
__device__ __forceinline__ bool ldl_right_solve_row(const double (&au)[10], const double (&d)[4], double (&k)[4]) {
    Ldl4 f;
    ldl_factor4(au, f);
    ldl_solve_row4(f, d, k);
    return f.bad;
}

// C = A * B (4x4)
__device__ __forceinline__ void mm(const double (&A)[4][4], const double (&B)[4][4], double (&C)[4][4]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            double acc = A[r][0] * B[0][c];
            STE_UNROLL
            for (int i = 1; i < 4; ++i) acc = fma(A[r][i], B[i][c], acc);
            C[r][c] = acc;
        }
    }
}

// C = A * B^T (4x4)
__device__ __forceinline__ void mmt(const double (&A)[4][4], const double (&B)[4][4], double (&C)[4][4]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            double acc = A[r][0] * B[c][0];
            STE_UNROLL
            for (int i = 1; i < 4; ++i) acc = fma(A[r][i], B[c][i], acc);
            C[r][c] = acc;
        }
    }
}

// C = sym(A * B^T): upper triangle computed, mirrored (used where the product is symmetric by construction).
__device__ __forceinline__ void mmt_sym(const double (&A)[4][4], const double (&B)[4][4], double (&C)[4][4]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) {
            double acc = A[r][0] * B[c][0];
            STE_UNROLL
            for (int i = 1; i < 4; ++i) acc = fma(A[r][i], B[c][i], acc);
            C[r][c] = acc;
            C[c][r] = acc;
        }
    }
}

}  // namespace ste

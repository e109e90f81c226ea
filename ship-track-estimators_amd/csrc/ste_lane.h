// ste_lane.h — the UKF forward step with ONE LANE PER TRACK, streamed (gfx950).
//
// Round 3 rebuild of the lane-per-track step.  The round-2 form kept the whole sigma fan (9 points in, 9 points out: 72
// doubles) and every 4x4 temporary of the measurement update live at once: 284 registers with accumulator-register
// shuffling (a quarter of its vector instructions were moves), one wave per SIMD.  Here
//   * symmetric matrices are packed upper triangles (10 doubles);
//   * the +/- pair of one sigma direction is pushed through the process model and folded into running moment sums at
//     once -- deviations are taken from the propagated CENTRE point, so no point outlives its direction;
//   * speed and heading pass through the process model with unit slope (non_linear_process.py:74-75), so their
//     deviations are +/- the columns of T and their share of every moment is formed from T directly;
//   * with the reference's H = diag(1, 1, 0, 0) (and an R confined to the same block) the measurement update is a
//     2-column problem in closed form.
// Same arithmetic as unscented.py:178-265 up to the order of a few additions; parity is checked against the
// reference-run goldens (tests/test_hip_parity.py, tests/test_round3.py).
#pragma once
#include "ste_math.h"

namespace ste {

// index of (r, c) in a packed symmetric 4x4: 00 01 02 03 11 12 13 22 23 33
__host__ __device__ constexpr int tix(int r, int c) {
    return r <= c ? (r * 4 - (r * (r - 1)) / 2 + (c - r)) : (c * 4 - (c * (c - 1)) / 2 + (r - c));
}

// ---- Jacobi on a packed symmetric matrix (same rotation formulas as jacobi_rot in ste_math.h) -----------------------
// Branch-free: a pair that needs no rotation (a_pq^2 <= kRotTol2 |a_pp a_qq|, NaN included) rotates by the identity, so
// every entry is rewritten unconditionally and no value has to be merged with its old self after a skipped branch --
// with `if (go) { ... }` around the body hipcc kept old and new copies of the 26 entries alive side by side and moved
// them back at every join: 25-40 v_mov per rotation, a quarter of the round-2 kernel's vector instructions.
template <int P, int Q>
__device__ __forceinline__ void jacobi_rot_p(double (&a)[10], double (&V)[4][4]) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    const double apq = a[tix(P, Q)];
    const double app = a[tix(P, P)], aqq = a[tix(Q, Q)];
    const bool go = apq * apq > kRotTol2 * fabs(app * aqq);  // false for NaN and for apq == 0
    const double delta = aqq - app;
    const double two_apq = apq + apq;
    const double rh = rsqrt_fast(go ? fma(delta, delta, two_apq * two_apq) : 1.0);
    const double c2 = fma(0.5 * fabs(delta), rh, 0.5);  // in [1/2, 1]
    const double rc = rsqrt_fast(c2);                   // 1 / cos
    const double c = go ? c2 * rc : 1.0;
    const double s = go ? (delta < 0.0 ? -apq : apq) * rh * rc : 0.0;
    const double t = s * rc;
    a[tix(P, P)] = fma(-t, apq, app);
    a[tix(Q, Q)] = fma(t, apq, aqq);
    a[tix(P, Q)] = go ? 0.0 : apq;
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        if (r != P && r != Q) {
            const double arp = a[tix(r, P)], arq = a[tix(r, Q)];
            a[tix(r, P)] = fma(c, arp, -(s * arq));
            a[tix(r, Q)] = fma(s, arp, c * arq);
        }
    }
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        const double vrp = V[r][P], vrq = V[r][Q];
        V[r][P] = fma(c, vrp, -(s * vrq));
        V[r][Q] = fma(s, vrp, c * vrq);
    }
}

__device__ __forceinline__ bool jacobi_needs_sweep(const double (&a)[10]) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    bool need = false;
    STE_UNROLL
    for (int p = 0; p < 3; ++p) {
        STE_UNROLL
        for (int q = p + 1; q < 4; ++q) {
            const double apq = a[tix(p, q)];
            need = need || (apq * apq > kRotTol2 * fabs(a[tix(p, p)] * a[tix(q, q)]));
        }
    }
    return need;
}

#ifdef STE_DEBUG_SWEEPS
__device__ unsigned long long g_dbg[64];
__device__ __forceinline__ void dbg_sweep_probe(const double (&a)[10], int sweep) {
    const bool need = jacobi_needs_sweep(a);
    int pairs_wave = 0, pairs_lane = 0;
    STE_UNROLL
    for (int p = 0; p < 3; ++p) {
        STE_UNROLL
        for (int q = p + 1; q < 4; ++q) {
            const double apq = a[tix(p, q)];
            const bool go = apq * apq > kRotTol2 * fabs(a[tix(p, p)] * a[tix(q, q)]);
            pairs_wave += __any(go) ? 1 : 0;
            pairs_lane += go ? 1 : 0;
        }
    }
    // off-diagonal size (max |apq| / sqrt|app aqq|) of this lane, as a decade
    double rel = 0.0;
    STE_UNROLL
    for (int p = 0; p < 3; ++p) {
        STE_UNROLL
        for (int q = p + 1; q < 4; ++q) rel = fmax(rel, fabs(a[tix(p, q)]) * rsqrt(fabs(a[tix(p, p)] * a[tix(q, q)]) + 1e-300));
    }
    int dec = rel > 0.0 ? (int)floor(-log10(rel)) : 20;
    dec = dec < 0 ? 0 : (dec > 20 ? 20 : dec);
    atomicAdd(&g_dbg[32 + dec], 1ull);  // histogram over (lane, probe) of -log10(rel)
    atomicAdd(&g_dbg[3], (unsigned long long)pairs_lane);
    if ((threadIdx.x & 63) == 0) {
        if (sweep == 0) atomicAdd(&g_dbg[0], 1ull);          // eigen-solves (wave level)
        if (__any(need)) atomicAdd(&g_dbg[1], 1ull);         // sweeps executed (wave level)
        atomicAdd(&g_dbg[2], (unsigned long long)pairs_wave);  // rotations some lane of the wave needed
        if (!__any(need)) atomicAdd(&g_dbg[8 + (sweep < 15 ? sweep : 15)], 1ull);  // sweeps per solve
    }
}
#endif

// Cyclic sweeps until no lane of the wave asks for one (the criterion of the rotations, evaluated up front, so that a
// sweep in which nothing would rotate is not run at all); a lane's rotations are gated by its own data only, so a
// track's result does not depend on which tracks share its wave.
__device__ __forceinline__ bool jacobi_sweeps_p(double (&a)[10], double (&V)[4][4]) {
    for (int sweep = 0; sweep <= kMaxSweeps; ++sweep) {
#ifdef STE_DEBUG_SWEEPS  // profiles/tools/jacobi_sweep_probe.py: how many sweeps a fan costs, and how many of their rotations any lane needed
        dbg_sweep_probe(a, sweep);
#endif
        // The first two sweeps run unasked: measured on the bench batch, 99.6 % of the solves need at least two (a warm
        // start leaves off-diagonals of 1e-3 ... 1e-1 of the diagonal, one sweep 1e-6 ... 1e-4), and a sweep over an
        // already diagonal matrix only rotates by the identity.
        if (sweep >= 2 && !__any(jacobi_needs_sweep(a))) return true;
        if (sweep == kMaxSweeps) break;
        jacobi_rot_p<0, 1>(a, V);
        jacobi_rot_p<2, 3>(a, V);
        jacobi_rot_p<0, 2>(a, V);
        jacobi_rot_p<1, 3>(a, V);
        jacobi_rot_p<0, 3>(a, V);
        jacobi_rot_p<1, 2>(a, V);
    }
    return !jacobi_needs_sweep(a);
}

// T = principal square root of (scale * P), negative eigenvalues clamped (unscented.py:95-97), both packed.
// V: eigenvectors of the previous step's matrix when `warm` (B = V^T A V is then nearly diagonal), else overwritten.
__device__ __forceinline__ int sym_sqrt_p(const double (&P)[10], double scale, double (&T)[10], double (&V)[4][4], bool warm) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    double a[10];
    double wmax = 0.0;
    STE_UNROLL
    for (int e = 0; e < 10; ++e) a[e] = scale * P[e];
    STE_UNROLL
    for (int i = 0; i < 4; ++i) wmax = fmax(wmax, fabs(a[tix(i, i)]));
    if (warm) {
        double M[4][4];
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = 0; c < 4; ++c) {
                double acc = a[tix(r, 0)] * V[0][c];
                STE_UNROLL
                for (int i = 1; i < 4; ++i) acc = fma(a[tix(r, i)], V[i][c], acc);
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
                a[tix(r, c)] = acc;
            }
        }
    } else {
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = 0; c < 4; ++c) V[r][c] = (r == c) ? 1.0 : 0.0;
        }
    }
    int st = jacobi_sweeps_p(a, V) ? 0 : 0x4;
    double Vf[4][4];
    STE_UNROLL
    for (int i = 0; i < 4; ++i) {
        const double w = a[tix(i, i)];
        if (w < -1e-12 * wmax) st |= 0x2;
        const double f = w > 0.0 ? w * rsqrt_fast(w) : 0.0;  // sqrt(max(w, 0)), ~1 ulp
        STE_UNROLL
        for (int r = 0; r < 4; ++r) Vf[r][i] = V[r][i] * f;
    }
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) {
            double acc = Vf[r][0] * V[c][0];
            STE_UNROLL
            for (int i = 1; i < 4; ++i) acc = fma(Vf[r][i], V[c][i], acc);
            T[tix(r, c)] = acc;
        }
    }
    return st;
}

// Running moments of the propagated fan about the propagated centre point c' (see file header).  With y = chi' - c':
//   s[c]     sum over the 8 outer points of y[c]                         (c = 0, 1; components 2, 3 cancel)
//   S        sum of y y^T, packed; rows/columns 2-3 from T:  y[2] = +-T[2][i], y[3] = +-T[3][i]
//   Dn[r][c] sum_i T[r][i] (chi'_{i+} - chi'_{i-})[c], r = 0, 1, c = 0, 1   (rows 2-3 of it are S[c][2], S[c][3])
struct FanMoments {
    double s0, s1;
    double S[10];
    double Dn[2][2];
    double TT[2][2];  // sum_i T[r][i] T[c][i], r = 0, 1, c = 2, 3: columns 2-3 of D over 2 wi (rows 2-3 of those are S[2:4, 2:4])
};

__device__ __forceinline__ void moments_clear(FanMoments& f) {
    f.s0 = 0.0;
    f.s1 = 0.0;
    STE_UNROLL
    for (int e = 0; e < 10; ++e) f.S[e] = 0.0;
    f.Dn[0][0] = f.Dn[0][1] = f.Dn[1][0] = f.Dn[1][1] = 0.0;
    f.TT[0][0] = f.TT[0][1] = f.TT[1][0] = f.TT[1][1] = 0.0;
}

// Fold the +/- pair of direction I: (lonp, latp), (lonm, latm) = propagated positions, c0, c1 = the centre's.
template <int I, bool kGains>
__device__ __forceinline__ void moments_add(FanMoments& f, const double (&T)[10], double c0, double c1, double lonp,
                                            double latp, double lonm, double latm) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    const double yp0 = lonp - c0, ym0 = lonm - c0, yp1 = latp - c1, ym1 = latm - c1;
    const double t0 = T[tix(0, I)], t1 = T[tix(1, I)], t2 = T[tix(2, I)], t3 = T[tix(3, I)];
    f.s0 += yp0 + ym0;
    f.s1 += yp1 + ym1;
    const double dl0 = yp0 - ym0, dl1 = yp1 - ym1;
    f.S[tix(0, 0)] = fma(yp0, yp0, fma(ym0, ym0, f.S[tix(0, 0)]));
    f.S[tix(0, 1)] = fma(yp0, yp1, fma(ym0, ym1, f.S[tix(0, 1)]));
    f.S[tix(1, 1)] = fma(yp1, yp1, fma(ym1, ym1, f.S[tix(1, 1)]));
    f.S[tix(0, 2)] = fma(t2, dl0, f.S[tix(0, 2)]);
    f.S[tix(0, 3)] = fma(t3, dl0, f.S[tix(0, 3)]);
    f.S[tix(1, 2)] = fma(t2, dl1, f.S[tix(1, 2)]);
    f.S[tix(1, 3)] = fma(t3, dl1, f.S[tix(1, 3)]);
    f.S[tix(2, 2)] = fma(t2, t2, f.S[tix(2, 2)]);  // doubled at the end (+ and - point)
    f.S[tix(2, 3)] = fma(t2, t3, f.S[tix(2, 3)]);
    f.S[tix(3, 3)] = fma(t3, t3, f.S[tix(3, 3)]);
    if (kGains) {
        f.Dn[0][0] = fma(t0, dl0, f.Dn[0][0]);
        f.Dn[0][1] = fma(t0, dl1, f.Dn[0][1]);
        f.Dn[1][0] = fma(t1, dl0, f.Dn[1][0]);
        f.Dn[1][1] = fma(t1, dl1, f.Dn[1][1]);
        f.TT[0][0] = fma(t0, t2, f.TT[0][0]);
        f.TT[0][1] = fma(t0, t3, f.TT[0][1]);
        f.TT[1][0] = fma(t1, t2, f.TT[1][0]);
        f.TT[1][1] = fma(t1, t3, f.TT[1][1]);
    }
}

// What the centre point contributes to every direction: its three sin/cos pairs and its propagated image.
struct FanCentre {
    double sp0, cp0, sa0, ca0, sd0, cd0;
    double c[4];  // geodetic_dynamics(x)
    double dt_r, du, da;
};

__device__ __forceinline__ void fan_centre(const double (&x)[4], double dt, double sr, double cr, FanCentre& g, bool& ok,
                                           const TrigReg& tk) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    g.dt_r = div_earth_radius(dt);
    g.du = sr * dt;
    g.da = cr * dt;
    const double lat0 = x[1] * kDeg2Rad, alpha0 = x[3] * kDeg2Rad, delta0 = x[2] * g.dt_r;
    const double a0[3] = {lat0, alpha0, delta0};
    double s_0[3], c_0[3];
    sincos_fast_n<3, TrigReg>(a0, s_0, c_0, ok, tk);
    g.sp0 = s_0[0];
    g.cp0 = c_0[0];
    g.sa0 = s_0[1];
    g.ca0 = c_0[1];
    g.sd0 = s_0[2];
    g.cd0 = c_0[2];
    const double lo[1] = {x[0] * kDeg2Rad}, la[1] = {lat0}, vsp[1] = {g.sp0}, vcp[1] = {g.cp0}, vsa[1] = {g.sa0},
                 vca[1] = {g.ca0}, vsd[1] = {g.sd0}, vcd[1] = {g.cd0};
    double lon_o[1], lat_o[1];
    geodetic_finish_n<1>(lo, la, vsp, vcp, vsa, vca, vsd, vcd, lon_o, lat_o, ok);
    g.c[0] = lon_o[0];
    g.c[1] = lat_o[0];
    g.c[2] = x[2] + g.du;
    g.c[3] = fma(alpha0, kRad2Deg, g.da);
}

// Positions of the +/- pair along column I of T after the great-circle step (non_linear_process.py:64-72): their angles
// are the centre's +- a small increment, so sin/cos follow from one sincos per increment by angle addition.
template <int I>
__device__ __forceinline__ void fan_pair(const double (&x)[4], const double (&T)[10], const FanCentre& g, double& lonp,
                                         double& latp, double& lonm, double& latm, bool& ok, const TrigReg& tk) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    const double t0 = T[tix(0, I)], t1 = T[tix(1, I)], t2 = T[tix(2, I)], t3 = T[tix(3, I)];
    const double dl[3] = {t1 * kDeg2Rad, t3 * kDeg2Rad, t2 * g.dt_r};
    double s_d[3], c_d[3];
    sincos_delta_n<3, TrigReg>(dl, s_d, c_d, ok, tk);
    const double p2 = g.cp0 * s_d[0], p4 = g.sp0 * s_d[0], a2 = g.ca0 * s_d[1], a4 = g.sa0 * s_d[1], d2 = g.cd0 * s_d[2],
                 d4 = g.sd0 * s_d[2];
    const double lo[2] = {(x[0] + t0) * kDeg2Rad, (x[0] - t0) * kDeg2Rad};
    const double la[2] = {(x[1] + t1) * kDeg2Rad, (x[1] - t1) * kDeg2Rad};
    const double vsp[2] = {fma(g.sp0, c_d[0], p2), fma(g.sp0, c_d[0], -p2)};
    const double vcp[2] = {fma(g.cp0, c_d[0], -p4), fma(g.cp0, c_d[0], p4)};
    const double vsa[2] = {fma(g.sa0, c_d[1], a2), fma(g.sa0, c_d[1], -a2)};
    const double vca[2] = {fma(g.ca0, c_d[1], -a4), fma(g.ca0, c_d[1], a4)};
    const double vsd[2] = {fma(g.sd0, c_d[2], d2), fma(g.sd0, c_d[2], -d2)};
    const double vcd[2] = {fma(g.cd0, c_d[2], -d4), fma(g.cd0, c_d[2], d4)};
    double lon_o[2], lat_o[2];
    geodetic_finish_n<2>(lo, la, vsp, vcp, vsa, vca, vsd, vcd, lon_o, lat_o, ok);
    lonp = lon_o[0];
    latp = lat_o[0];
    lonm = lon_o[1];
    latm = lat_o[1];
}

// Moore-Penrose pseudo-inverse of a packed symmetric 4x4 with NumPy's cutoff (np.linalg.pinv, unscented.py:333): the
// eigenvalue route of sym_pinv4, on the packed branch-free Jacobi.  Cold start; a rare path (see smoother_gain).
__device__ __forceinline__ int sym_pinv_p(const double (&A)[10], double (&Ai)[10]) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    double a[10], V[4][4];
    STE_UNROLL
    for (int e = 0; e < 10; ++e) a[e] = A[e];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) V[r][c] = (r == c) ? 1.0 : 0.0;
    }
    const int st = jacobi_sweeps_p(a, V) ? 0 : 0x4;
    double smax = 0.0;
    STE_UNROLL
    for (int i = 0; i < 4; ++i) smax = fmax(smax, fabs(a[tix(i, i)]));
    const double cutoff = kPinvRcond * smax;
    double Vf[4][4];
    STE_UNROLL
    for (int i = 0; i < 4; ++i) {
        const double w = a[tix(i, i)];
        const double f = (fabs(w) > cutoff) ? 1.0 / w : 0.0;
        STE_UNROLL
        for (int r = 0; r < 4; ++r) Vf[r][i] = V[r][i] * f;
    }
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) {
            double acc = Vf[r][0] * V[c][0];
            STE_UNROLL
            for (int i = 1; i < 4; ++i) acc = fma(Vf[r][i], V[c][i], acc);
            Ai[tix(r, c)] = acc;
        }
    }
    return st;
}

// K = D pinv(P_b) by the eigenvalue route, out of line: it is the rare path of smoother_gain (P_b close to singular, or a
// test asking for it), and inlined its Jacobi working set would set the register allocation of the whole smoother
// kernel.  Arguments live in memory (the caller's stack), which is what keeps the hot path's arrays in registers.
__device__ __noinline__ int smoother_gain_eig(const double* pb, const double* d, double* k) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    double Pb[10], Pbi[10];
    STE_UNROLL
    for (int e = 0; e < 10; ++e) Pb[e] = pb[e];
    const int st = sym_pinv_p(Pb, Pbi);
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            double acc = d[r * 4 + 0] * Pbi[tix(0, c)];
            STE_UNROLL
            for (int i = 1; i < 4; ++i) acc = fma(d[r * 4 + i], Pbi[tix(i, c)], acc);
            k[r * 4 + c] = acc;
        }
    }
    return st;
}

// The smoother's gain K = D pinv(P_b) (unscented.py:333) for one step: by an unpivoted L D L^T solve where P_b is safely
// invertible (pinv is then the inverse; agreement ~ cond(P_b) 2^-52), by the eigenvalue route -- with NumPy's rank
// cutoff -- for the lanes where a pivot falls under kLdlPivotTol of the largest diagonal entry, and for every lane when
// `all_eig` (tuning bit 8: tests).  Returns status bits of the eigenvalue route for the lanes that used it.
__device__ __forceinline__ int smoother_gain(const double (&Pb)[10], const double (&D)[4][4], bool all_eig,
                                             double (&K)[4][4]) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    const bool bad = ldl_right_solve4(Pb, D, K) || all_eig;
    int st = 0;
    // a lane whose P_b is already non-finite (first pivot NaN: every factor and K are NaN on either route) does not send
    // its wave through the eigenvalue route -- a dead track must not slow the live ones beside it at every step
    if (__builtin_expect(__any(bad && Pb[0] == Pb[0]), 0)) {
        double pbm[10], dm[16], km[16];
        STE_UNROLL
        for (int e = 0; e < 10; ++e) pbm[e] = Pb[e];
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = 0; c < 4; ++c) dm[r * 4 + c] = D[r][c];
        }
        const int pst = smoother_gain_eig(pbm, dm, km);
        if (bad) {
            st = pst;
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) K[r][c] = km[r * 4 + c];
            }
        }
    }
    return st;
}

// The same for ONE row of D (the quad kernels: a lane holds the whole packed P_b and row q of D).
__device__ __forceinline__ int quad_smoother_gain(const double (&Pb)[10], const double (&Drow)[4], bool all_eig,
                                                  double (&Krow)[4]) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    const bool bad = ldl_right_solve_row(Pb, Drow, Krow) || all_eig;
    int st = 0;
    if (__builtin_expect(__any(bad && Pb[0] == Pb[0]), 0)) {
        double Pbi[10];
        const int pst = sym_pinv_p(Pb, Pbi);
        if (bad) {
            st = pst;
            STE_UNROLL
            for (int c = 0; c < 4; ++c) {
                double acc = Drow[0] * Pbi[tix(0, c)];
                STE_UNROLL
                for (int i = 1; i < 4; ++i) acc = fma(Drow[i], Pbi[tix(i, c)], acc);
                Krow[c] = acc;
            }
        }
    }
    return st;
}

// Opt-in robustification (check_robustness, unscented.py:353-387) for the same H = diag(1, 1, 0, 0) / block-R case, in
// closed form: S = H P H^T + R lives in the leading 2 x 2 block, so with y = z - x (the reference's innovation here, :420)
//   gamma = |y^T S^+ y| = |y01^T Sb^+ y01|      criterion_index, :420-426
//   denom = y^T S^+ R S^+ y = u^T Rb u, u = Sb^+ y01   update_lambda_factor, :468-478
// and while gamma > chi_alpha: lambda += (gamma - chi_alpha) / denom, R <- lambda R (compounding, as written there).
// Rescales (r00, r01, r11) in place; returns STE_STATUS_ROBUST_CAP when the criterion is still above chi_alpha after
// robust_iters rescalings.  Branch-free per lane inside a wave-uniform loop: a lane that is done keeps its values.  The
// unobserved components of y enter the reference's products through exact zeros of S^+ (0 * NaN = NaN): kept as a poison term.
__device__ __forceinline__ int robust_rescale_sel2(const Mats& p, const double (&x)[4], const double (&P)[10],
                                                    const double (&z)[4], double& r00, double& r01, double& r11) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    const double y0 = z[0] - x[0], y1 = z[1] - x[1];
    const double poison = fma(0.0, z[2] - x[2], 0.0 * (z[3] - x[3]));
    auto terms = [&](double q00, double q01, double q11, double& gamma, double& denom) {
        double Sm[4][4], Si[4][4];
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = 0; c < 4; ++c) Sm[r][c] = 0.0;
        }
        Sm[0][0] = P[tix(0, 0)] + q00;
        Sm[0][1] = P[tix(0, 1)] + q01;
        Sm[1][0] = Sm[0][1];
        Sm[1][1] = P[tix(1, 1)] + q11;
        sym_pinv4_block2(Sm, Si);
        const double u0 = fma(Si[0][1], y1, Si[0][0] * y0) + poison, u1 = fma(Si[1][1], y1, Si[0][1] * y0) + poison;
        gamma = fabs(fma(y1, u1, y0 * u0));
        const double v0 = fma(q01, u1, q00 * u0), v1 = fma(q11, u1, q01 * u0);
        denom = fma(u1, v1, u0 * v0);
    };
    double gamma, denom, lambda = 1.0;
    terms(r00, r01, r11, gamma, denom);
    for (int it = 0; it < p.robust_iters; ++it) {
        const bool active = gamma > p.chi_alpha;
        if (!__any(active)) break;
        const double l2 = lambda + (gamma - p.chi_alpha) / denom;
        lambda = active ? l2 : lambda;
        r00 = active ? r00 * lambda : r00;
        r01 = active ? r01 * lambda : r01;
        r11 = active ? r11 * lambda : r11;
        double g2, d2;
        terms(r00, r01, r11, g2, d2);
        gamma = active ? g2 : gamma;
        denom = active ? d2 : denom;
    }
    return gamma > p.chi_alpha ? STE_STATUS_ROBUST_CAP : 0;
}

// Measurement update for H = diag(1, 1, 0, 0) and an R that is zero outside its leading 2 x 2 block (unscented.py:219-265
// with the matrices every example and the CLI of the reference use): S = H P H^T + R lives in that block, its
// pseudo-inverse is sym_pinv4_block2's single rotation, K = P H^T S^+ has two columns, and the Joseph form
// (I - K H) P (I - K H)^T + K R K^T needs the products with those two columns only.  r00, r01, r11: the block of R.
// The unobserved components of the innovation still reach the state through exact zeros of K (0 * NaN = NaN in the
// reference when an observation carries a non-finite speed or course): kept as a poison term.
__device__ __forceinline__ void lane_update_sel2(double r00, double r01, double r11, double (&x)[4], double (&P)[10],
                                                 const double (&z)[4]) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    double Sm[4][4], Si[4][4];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) Sm[r][c] = 0.0;
    }
    Sm[0][0] = P[tix(0, 0)] + r00;
    Sm[0][1] = P[tix(0, 1)] + r01;
    Sm[1][0] = Sm[0][1];
    Sm[1][1] = P[tix(1, 1)] + r11;
    sym_pinv4_block2(Sm, Si);
    const double i00 = Si[0][0], i01 = Si[0][1], i11 = Si[1][1];
    double K[4][2];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        const double pr0 = P[tix(r, 0)], pr1 = P[tix(r, 1)];
        K[r][0] = fma(pr1, i01, pr0 * i00);
        K[r][1] = fma(pr1, i11, pr0 * i01);
    }
    const double y0 = z[0] - x[0], y1 = z[1] - x[1];
    const double poison = fma(0.0, z[2], 0.0 * z[3]);  // K[:, 2:4] y[2:4] with K[:, 2:4] = 0: NaN iff z[2] or z[3] is not finite
    STE_UNROLL
    for (int r = 0; r < 4; ++r) x[r] = fma(K[r][1], y1, fma(K[r][0], y0, x[r])) + poison;
    x[3] = floored_mod(x[3], 360.0);
    // AP = (I - K H) P: every entry but (3, 2)
    double AP[4][4];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            if (r == 3 && c == 2) continue;
            AP[r][c] = fma(-K[r][1], P[tix(1, c)], fma(-K[r][0], P[tix(0, c)], P[tix(r, c)]));
        }
    }
    double KR[4][2];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        KR[r][0] = fma(K[r][1], r01, K[r][0] * r00);
        KR[r][1] = fma(K[r][1], r11, K[r][0] * r01);
    }
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) {
            const double p1 = fma(-AP[r][1], K[c][1], fma(-AP[r][0], K[c][0], AP[r][c]));  // (A P A^T)[r][c]
            const double p2 = fma(KR[r][1], K[c][1], KR[r][0] * K[c][0]);                   // (K R K^T)[r][c]
            P[tix(r, c)] = p1 + p2;
        }
    }
}

}  // namespace ste

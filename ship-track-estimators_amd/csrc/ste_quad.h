// ste_quad.h — the UKF forward step with ONE DPP QUAD (4 adjacent lanes) PER TRACK, 16 tracks per wave (gfx950).
//
// Why: a 10 000-track batch is only 157 waves when a lane owns a track, i.e. 15 % of the chip's 1024 SIMDs, each issuing
// one fp64 instruction per ~9.5 cycles.  Spreading a track over the four lanes of a quad puts 625 waves on the chip and
// shortens the per-wave instruction stream: each lane propagates one +- pair of sigma points (plus the shared centre),
// the two disjoint Jacobi rotations of a round run in the two lane pairs, and 4x4 matrix work is one row per lane.
// Cross-lane traffic is DPP quad_perm moves (two v_mov_b32_dpp per double, no LDS).
//
// Layout, for lane q in {0,1,2,3} of a quad:
//   x[4]      state mean, replicated in the four lanes
//   Px[4]     row q of the covariance in XOR order: Px[k] = P[q][q ^ k]   (slot 0 is the diagonal)
//   Vx[4]     row q of the eigenvector matrix, same XOR order
// The XOR order makes every register index in the Jacobi sweeps, the similarity transform and the moment reductions a
// compile-time constant: the partner of lane q in round m is lane q ^ m, the pivot a_pq sits in slot m of both, and the
// column that partner p holds in slot k is the one this lane holds in slot k ^ m.
#pragma once
#include "ste_math.h"

namespace ste {

// ---- DPP helpers ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_move_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
// value held by lane (q ^ K) of the same quad; K = 0 is the lane itself
template <int K>
__device__ __forceinline__ double fetch(double v) {
    static_assert(K >= 0 && K <= 3, "quad xor");
    if (K == 0) return v;
    return dpp_move<(K == 1) ? 0xB1 : (K == 2) ? 0x4E : 0x1B>(v);  // quad_perm [1,0,3,2] / [2,3,0,1] / [3,2,1,0]
}
// value held by lane L of the quad
template <int L>
__device__ __forceinline__ double bcast(double v) {
    return dpp_move<L * 0x55>(v);
}

// out[k] = v[k ^ q]  (its own inverse): two conditional-swap stages.
__device__ __forceinline__ void xorperm(const double (&v)[4], int q, double (&out)[4]) {
    const bool b0 = q & 1, b1 = q & 2;
    const double t0 = b0 ? v[1] : v[0], t1 = b0 ? v[0] : v[1], t2 = b0 ? v[3] : v[2], t3 = b0 ? v[2] : v[3];
    out[0] = b1 ? t2 : t0;
    out[1] = b1 ? t3 : t1;
    out[2] = b1 ? t0 : t2;
    out[3] = b1 ? t1 : t3;
}
__device__ __forceinline__ double sel4(const double (&v)[4], int q) {
    const double a = (q & 1) ? v[1] : v[0], b = (q & 1) ? v[3] : v[2];
    return (q & 2) ? b : a;
}
__device__ __forceinline__ double quad_sum(double v) {
    v += fetch<1>(v);
    return v + fetch<2>(v);
}
__device__ __forceinline__ double quad_max(double v) {
    v = fmax(v, fetch<1>(v));
    return fmax(v, fetch<2>(v));
}

// Per-lane constants derived from the shared matrices once per kernel.
struct QuadCtx {
    int q;
    bool lo[4];        // lo[m]: this lane is the smaller index of its pair {q, q ^ m}
    double HTx[4][4];  // HTx[c][k] = H[c][q ^ k]
    double Hrow[4];    // H[q][l]
    double Rrow[4];    // R[q][c]
    double Qx[4];      // Q[q][q ^ k]
};

__device__ __forceinline__ void quad_ctx_init(const Mats& m, int q, QuadCtx& cx) {
    cx.q = q;
    cx.lo[0] = true;
    cx.lo[1] = !(q & 1);
    cx.lo[2] = !(q & 2);
    cx.lo[3] = !(q & 2);
    STE_UNROLL
    for (int c = 0; c < 4; ++c) {
        const double hr[4] = {m.H[c * 4 + 0], m.H[c * 4 + 1], m.H[c * 4 + 2], m.H[c * 4 + 3]};
        xorperm(hr, q, cx.HTx[c]);
    }
    STE_UNROLL
    for (int l = 0; l < 4; ++l) {
        const double hc[4] = {m.H[0 * 4 + l], m.H[1 * 4 + l], m.H[2 * 4 + l], m.H[3 * 4 + l]};
        const double rc[4] = {m.R[0 * 4 + l], m.R[1 * 4 + l], m.R[2 * 4 + l], m.R[3 * 4 + l]};
        const double qc[4] = {m.Q[0 * 4 + l], m.Q[1 * 4 + l], m.Q[2 * 4 + l], m.Q[3 * 4 + l]};
        cx.Hrow[l] = sel4(hc, q);
        cx.Rrow[l] = sel4(rc, q);
        cx.Qx[l] = sel4(qc, q);  // Q[q][l], natural for now
    }
    double qn[4] = {cx.Qx[0], cx.Qx[1], cx.Qx[2], cx.Qx[3]};
    xorperm(qn, q, cx.Qx);
}

// ---- Jacobi in the quad ---------------------------------------------------------------------------------------
// One round: the two disjoint rotations {q, q^M} run at once, one in each lane pair.  Each lane works in its own
// convention (itself = "p", its partner = "q" of jacobi_rot in ste_math.h); the partner then holds the same cosine and
// the opposite sine, which is exactly the transposed role.
template <int M>
__device__ __forceinline__ bool quad_jacobi_round(double (&A)[4], double (&V)[4], const QuadCtx& cx) {
    constexpr int K1 = (M == 1) ? 2 : 1;
    constexpr int K2 = K1 ^ M;
    const bool is_lo = cx.lo[M];
    const double e0 = A[0], em = A[M];
    const double pe0 = fetch<M>(e0), pem = fetch<M>(em);
    const double apq = is_lo ? em : pem;  // both lanes of the pair use the value the smaller lane holds
    const bool go = apq * apq > kRotTol2 * fabs(e0 * pe0);
    const double delta = pe0 - e0;
    const double two_apq = apq + apq;
    const double h2 = fma(delta, delta, two_apq * two_apq);
    const double rh = rsqrt_fast(go ? h2 : 1.0);
    const double c2 = fma(0.5 * fabs(delta), rh, 0.5);
    const double rc = rsqrt_fast(c2);
    const bool neg = (delta < 0.0) || (delta == 0.0 && !is_lo);
    const double c = go ? c2 * rc : 1.0;
    const double s = go ? (neg ? -apq : apq) * rh * rc : 0.0;
    // the other pair's rotation, seen from its lane q ^ K1 (whose "own" column is this lane's slot K1)
    const double c1 = fetch<K1>(c), s1 = fetch<K1>(s);
    // column mixing of this lane's row of A and of V
    const double a0 = fma(c, A[0], -(s * A[M])), am = fma(s, A[0], c * A[M]);
    const double ak1 = fma(c1, A[K1], -(s1 * A[K2])), ak2 = fma(s1, A[K1], c1 * A[K2]);
    const double v0 = fma(c, V[0], -(s * V[M])), vm = fma(s, V[0], c * V[M]);
    const double vk1 = fma(c1, V[K1], -(s1 * V[K2])), vk2 = fma(s1, V[K1], c1 * V[K2]);
    V[0] = v0;
    V[M] = vm;
    V[K1] = vk1;
    V[K2] = vk2;
    // row mixing with the partner's (column-mixed) row: its slot k ^ M holds the column this lane has in slot k
    const double f0 = fetch<M>(a0), fm = fetch<M>(am), fk1 = fetch<M>(ak1), fk2 = fetch<M>(ak2);
    A[0] = fma(c, a0, -(s * fm));
    const double newm = fma(c, am, -(s * f0));
    A[M] = go ? 0.0 : newm;
    A[K1] = fma(c, ak1, -(s * fk2));
    A[K2] = fma(c, ak2, -(s * fk1));
    return go;
}

// Sweeps until no pair of any track in the wave needs a rotation.  A: row q of the symmetric matrix (XOR order),
// destroyed (slot 0 ends as eigenvalue q).  V: row q of the accumulated eigenvectors (XOR order), initialised by the
// caller.  The rounds are branch-free (a converged pair rotates by the identity), so a sweep in which nothing rotates
// would cost as much as a productive one; the same criterion the rounds apply is therefore evaluated up front -- three
// fetches of the partners' diagonal entries -- and the loop ends as soon as no lane of the wave asks for a sweep.
__device__ __forceinline__ bool quad_jacobi_sweeps(double (&A)[4], double (&V)[4], const QuadCtx& cx) {
    for (int sweep = 0; sweep <= kMaxSweeps; ++sweep) {
        const double d1 = fetch<1>(A[0]), d2 = fetch<2>(A[0]), d3 = fetch<3>(A[0]);
        const bool need = (A[1] * A[1] > kRotTol2 * fabs(A[0] * d1)) || (A[2] * A[2] > kRotTol2 * fabs(A[0] * d2)) ||
                          (A[3] * A[3] > kRotTol2 * fabs(A[0] * d3));
        if (!__any(need)) return true;
        if (sweep == kMaxSweeps) break;
        quad_jacobi_round<1>(A, V, cx);
        quad_jacobi_round<2>(A, V, cx);
        quad_jacobi_round<3>(A, V, cx);
    }
    return false;
}

// B = V^T A V in XOR order (A, V XOR-order rows); A is overwritten with B.
__device__ __forceinline__ void quad_similarity(double (&A)[4], const double (&V)[4]) {
    double pv[4][4], pm[4][4], M[4];
    STE_UNROLL
    for (int s = 0; s < 4; ++s) {
        pv[0][s] = V[s];
        pv[1][s] = fetch<1>(V[s]);
        pv[2][s] = fetch<2>(V[s]);
        pv[3][s] = fetch<3>(V[s]);
    }
    STE_UNROLL
    for (int s = 0; s < 4; ++s) {
        double acc = A[0] * pv[0][s];
        STE_UNROLL
        for (int k = 1; k < 4; ++k) acc = fma(A[k], pv[k][k ^ s], acc);
        M[s] = acc;
    }
    STE_UNROLL
    for (int s = 0; s < 4; ++s) {
        pm[0][s] = M[s];
        pm[1][s] = fetch<1>(M[s]);
        pm[2][s] = fetch<2>(M[s]);
        pm[3][s] = fetch<3>(M[s]);
    }
    STE_UNROLL
    for (int s = 0; s < 4; ++s) {
        double acc = pv[0][0] * pm[0][s];
        STE_UNROLL
        for (int k = 1; k < 4; ++k) acc = fma(pv[k][k], pm[k][k ^ s], acc);
        A[s] = acc;
    }
}

// Row q (XOR order) of V diag(f) V^T, f = this lane's spectral value (lane l owns eigenvalue l).
__device__ __forceinline__ void quad_recompose(const double (&V)[4], double f, double (&out)[4]) {
    double g[4];
    g[0] = V[0] * f;
    g[1] = V[1] * fetch<1>(f);
    g[2] = V[2] * fetch<2>(f);
    g[3] = V[3] * fetch<3>(f);
    {
        double acc = g[0] * V[0];
        STE_UNROLL
        for (int k = 1; k < 4; ++k) acc = fma(g[k], V[k], acc);
        out[0] = acc;
    }
    {  // s = 1
        const double p0 = fetch<1>(V[0]), p1 = fetch<1>(V[1]), p2 = fetch<1>(V[2]), p3 = fetch<1>(V[3]);
        out[1] = fma(g[3], p2, fma(g[2], p3, fma(g[1], p0, g[0] * p1)));
    }
    {  // s = 2
        const double p0 = fetch<2>(V[0]), p1 = fetch<2>(V[1]), p2 = fetch<2>(V[2]), p3 = fetch<2>(V[3]);
        out[2] = fma(g[3], p1, fma(g[2], p0, fma(g[1], p3, g[0] * p2)));
    }
    {  // s = 3
        const double p0 = fetch<3>(V[0]), p1 = fetch<3>(V[1]), p2 = fetch<3>(V[2]), p3 = fetch<3>(V[3]);
        out[3] = fma(g[3], p0, fma(g[2], p1, fma(g[1], p2, g[0] * p3)));
    }
}

struct QuadBasis {
    double V[4];  // XOR-order row of the eigenvectors of the previous step's matrix
    bool valid;
};

// Row q (natural order) of sqrtm(scale * P) with negative eigenvalues clamped (unscented.py:95-97).
__device__ __forceinline__ int quad_sym_sqrt(const double (&Px)[4], double scale, const QuadCtx& cx, QuadBasis& basis,
                                             double (&Tn)[4]) {
    double A[4];
    A[0] = scale * Px[0];
    A[1] = scale * (0.5 * (Px[1] + fetch<1>(Px[1])));  // symmetrise like sym_sqrt4: the partner holds the transposed entry
    A[2] = scale * (0.5 * (Px[2] + fetch<2>(Px[2])));
    A[3] = scale * (0.5 * (Px[3] + fetch<3>(Px[3])));
    const double wmax = quad_max(fabs(A[0]));
    if (basis.valid) {
        quad_similarity(A, basis.V);
    } else {
        basis.V[0] = 1.0;
        basis.V[1] = 0.0;
        basis.V[2] = 0.0;
        basis.V[3] = 0.0;
    }
    int st = quad_jacobi_sweeps(A, basis.V, cx) ? 0 : 0x4;
    basis.valid = true;
    const double w = A[0];
    if (w < -1e-12 * wmax) st |= 0x2;
    double Tx[4];
    // sqrt(max(w, 0)) as w * rsqrt(w): a third of the instructions of the correctly rounded sqrt sequence
    quad_recompose(basis.V, w > 0.0 ? w * rsqrt_fast(w) : 0.0, Tx);
    xorperm(Tx, cx.q, Tn);
    return st;
}

// Row q (natural order) of pinv(S) for a symmetric S given as natural-order rows (np.linalg.pinv cutoff, unscented.py:243).
__device__ __forceinline__ int quad_sym_pinv(const double (&Sn)[4], const QuadCtx& cx, double (&Sin)[4]) {
    // Fast path: S = H P H^T + R with an H that observes two components (the reference's H = diag(1, 1, 0, 0)) is zero
    // outside its leading 2 x 2 block.  Its eigen-decomposition is then ONE exact rotation, evaluated on every lane with
    // the formulas of quad_jacobi_round, instead of a three-round sweep plus a confirming pass over a 4 x 4 matrix that
    // is three quarters zeros.  Taken only when every track of the wave has that structure.
    {
        const int q = cx.q;
        // (a quad whose S is already non-finite does not veto the fast path for its wave: its result is NaN either way)
        const bool fin = (Sn[0] + Sn[1] + Sn[2] + Sn[3]) * 0.0 == 0.0;
        const bool blk = !fin || ((q < 2) ? (Sn[2] == 0.0 && Sn[3] == 0.0)
                                          : (Sn[0] == 0.0 && Sn[1] == 0.0 && Sn[2] == 0.0 && Sn[3] == 0.0));
        if (__all(blk)) {
            const double a = bcast<0>(Sn[0]), d = bcast<1>(Sn[1]);
            const double b = 0.5 * (bcast<0>(Sn[1]) + bcast<1>(Sn[0]));
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
            // pinv = V diag(f) V^T with V = [[c, s], [-s, c]]
            const double cc = c * c, ss = sn * sn, cs = c * sn;
            const double i00 = fma(cc, f0, ss * f1), i11 = fma(ss, f0, cc * f1), i01 = cs * (f1 - f0);
            Sin[0] = (q == 0) ? i00 : (q == 1 ? i01 : 0.0);
            Sin[1] = (q == 0) ? i01 : (q == 1 ? i11 : 0.0);
            Sin[2] = 0.0;
            Sin[3] = 0.0;
            return 0;
        }
    }
    double A[4], V[4] = {1.0, 0.0, 0.0, 0.0};
    xorperm(Sn, cx.q, A);
    // symmetrise like sym_pinv4: average with the transposed entry held by the partner lanes
    A[1] = 0.5 * (A[1] + fetch<1>(A[1]));
    A[2] = 0.5 * (A[2] + fetch<2>(A[2]));
    A[3] = 0.5 * (A[3] + fetch<3>(A[3]));
    const int st = quad_jacobi_sweeps(A, V, cx) ? 0 : 0x4;
    const double w = A[0];
    const double cutoff = kPinvRcond * quad_max(fabs(w));
    const double f = (fabs(w) > cutoff) ? 1.0 / w : 0.0;
    double Six[4];
    quad_recompose(V, f, Six);
    xorperm(Six, cx.q, Sin);
    return st;
}

// out_r[c] = sum_l a_r[l] * (row l of the matrix whose rows live one per lane, natural order) [c]
__device__ __forceinline__ void quad_mm_rows(const double (&a)[4], const double (&rows)[4], double (&out)[4]) {
    STE_UNROLL
    for (int c = 0; c < 4; ++c) {
        double acc = a[0] * bcast<0>(rows[c]);
        acc = fma(a[1], bcast<1>(rows[c]), acc);
        acc = fma(a[2], bcast<2>(rows[c]), acc);
        acc = fma(a[3], bcast<3>(rows[c]), acc);
        out[c] = acc;
    }
}
// out_r[c] = sum_l a_r[l] * B[c][l]  where row c of B lives in lane c (natural order)
__device__ __forceinline__ void quad_mm_rows_t(const double (&a)[4], const double (&rows)[4], double (&out)[4]) {
    double acc0 = a[0] * bcast<0>(rows[0]), acc1 = a[0] * bcast<1>(rows[0]), acc2 = a[0] * bcast<2>(rows[0]),
           acc3 = a[0] * bcast<3>(rows[0]);
    STE_UNROLL
    for (int l = 1; l < 4; ++l) {
        acc0 = fma(a[l], bcast<0>(rows[l]), acc0);
        acc1 = fma(a[l], bcast<1>(rows[l]), acc1);
        acc2 = fma(a[l], bcast<2>(rows[l]), acc2);
        acc3 = fma(a[l], bcast<3>(rows[l]), acc3);
    }
    out[0] = acc0;
    out[1] = acc1;
    out[2] = acc2;
    out[3] = acc3;
}

// Weighted scatter of the deviations around `centre`, as XOR-order row q (plus Q):
//   P[q][q^s] = w0 d0[q] d0[q^s] + wi sum_{lanes l} ( d+_l[q] d+_l[q^s] + d-_l[q] d-_l[q^s] ) + Q[q][q^s]
// Every lane forms the products of ITS pair for all four target rows in XOR order; row r then collects its four
// contributions with three fetches per slot (a reduce-scatter over the quad).
__device__ __forceinline__ void quad_scatter(const double (&s0)[4], const double (&sp)[4], const double (&sm)[4],
                                             const double (&centre)[4], double w0, double wi, const QuadCtx& cx,
                                             double (&Pout)[4]) {
    double d0[4], dp[4], dm[4], d0x[4], dpx[4], dmx[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) {
        d0[c] = s0[c] - centre[c];
        dp[c] = sp[c] - centre[c];
        dm[c] = sm[c] - centre[c];
    }
    xorperm(d0, cx.q, d0x);
    xorperm(dp, cx.q, dpx);
    xorperm(dm, cx.q, dmx);
    // u(a,b) = d+[a] d+[b] + d-[a] d-[b] on XOR slots; N[k][s] = u(k, k ^ s)
    double u[4][4];
    STE_UNROLL
    for (int a = 0; a < 4; ++a) {
        STE_UNROLL
        for (int b = a; b < 4; ++b) {
            const double v = fma(dpx[a], dpx[b], dmx[a] * dmx[b]);
            u[a][b] = v;
            u[b][a] = v;
        }
    }
    STE_UNROLL
    for (int s = 0; s < 4; ++s) {
        double acc = u[0][s];
        acc += fetch<1>(u[1][1 ^ s]);
        acc += fetch<2>(u[2][2 ^ s]);
        acc += fetch<3>(u[3][3 ^ s]);
        Pout[s] = fma(w0 * d0x[0], d0x[s], fma(wi, acc, cx.Qx[s]));
    }
}

// Propagated sigma points this lane is responsible for: the centre (same in every lane) and the +- pair along column q
// of T.  Tn = row q of T (natural order) = column q.
__device__ __forceinline__ void quad_propagate_branching(const double (&x)[4], const double (&Tn)[4], double dt, double sr,
                                               double cr, double (&s0)[4], double (&sp)[4], double (&sm)[4]) {
    const double dt_r = dt / kEarthRadius;
    const double du = sr * dt, da = cr * dt;
    const double lat0 = x[1] * kDeg2Rad, alpha0 = x[3] * kDeg2Rad, delta0 = x[2] * dt_r;
    double sp0, cp0, sa0, ca0, sd0, cd0;
    sincos_fast(lat0, sp0, cp0);
    sincos_fast(alpha0, sa0, ca0);
    sincos_fast(delta0, sd0, cd0);
    geodetic_finish(x[0] * kDeg2Rad, lat0, sp0, cp0, sa0, ca0, sd0, cd0, s0[0], s0[1]);
    s0[2] = x[2] + du;
    s0[3] = alpha0 * kRad2Deg + da;
    double sdp, cdp, sda, cda, sdd, cdd;
    sincos_delta(Tn[1] * kDeg2Rad, sdp, cdp);
    sincos_delta(Tn[3] * kDeg2Rad, sda, cda);
    sincos_delta(Tn[2] * dt_r, sdd, cdd);
    const double p1 = sp0 * cdp, p2 = cp0 * sdp, p3 = cp0 * cdp, p4 = sp0 * sdp;
    const double a1 = sa0 * cda, a2 = ca0 * sda, a3 = ca0 * cda, a4 = sa0 * sda;
    const double d1 = sd0 * cdd, d2 = cd0 * sdd, d3 = cd0 * cdd, d4 = sd0 * sdd;
    {
        double pt[4];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) pt[c] = x[c] + Tn[c];
        geodetic_finish(pt[0] * kDeg2Rad, pt[1] * kDeg2Rad, p1 + p2, p3 - p4, a1 + a2, a3 - a4, d1 + d2, d3 - d4, sp[0],
                        sp[1]);
        sp[2] = pt[2] + du;
        sp[3] = (pt[3] * kDeg2Rad) * kRad2Deg + da;
    }
    {
        double pt[4];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) pt[c] = x[c] - Tn[c];
        geodetic_finish(pt[0] * kDeg2Rad, pt[1] * kDeg2Rad, p1 - p2, p3 + p4, a1 - a2, a3 + a4, d1 - d2, d3 + d4, sm[0],
                        sm[1]);
        sm[2] = pt[2] + du;
        sm[3] = (pt[3] * kDeg2Rad) * kRad2Deg + da;
    }
}


// The same three points with the three-at-a-time, branch-free helpers of ste_math.h; the rare wave in which some lane
// leaves their validity range (a pole, a giant step, non-finite data) redoes the step with the branching version.
// (KT, KG: where the polynomial coefficients come from -- literals, or registers set up once per kernel: ste_math.h)
template <class KT = TrigLit, class KG = GeoLit>
__device__ __forceinline__ void quad_propagate(const double (&x)[4], const double (&Tn)[4], double dt, double sr,
                                               double cr, double (&s0)[4], double (&sp)[4], double (&sm)[4],
                                               const KT& tk = KT(), const KG& gk = KG()) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    const double dt_r = div_earth_radius(dt);
    const double du = sr * dt, da = cr * dt;
    const double lat0 = x[1] * kDeg2Rad, alpha0 = x[3] * kDeg2Rad, delta0 = x[2] * dt_r;
    bool ok = true;
    const double a0[3] = {lat0, alpha0, delta0};
    double s_0[3], c_0[3];
    sincos_fast_n<3, KT>(a0, s_0, c_0, ok, tk);
    const double dl[3] = {Tn[1] * kDeg2Rad, Tn[3] * kDeg2Rad, Tn[2] * dt_r};
    double s_d[3], c_d[3];
    sincos_delta_n<3, KT>(dl, s_d, c_d, ok, tk);
    const double sp0 = s_0[0], cp0 = c_0[0], sa0 = s_0[1], ca0 = c_0[1], sd0 = s_0[2], cd0 = c_0[2];
    // angle addition for the +/- pair: sin(a +- d) = sin a cos d +- cos a sin d, cos(a +- d) = cos a cos d -+ sin a sin d
    const double p2 = cp0 * s_d[0], p4 = sp0 * s_d[0], a2 = ca0 * s_d[1], a4 = sa0 * s_d[1], d2 = cd0 * s_d[2],
                 d4 = sd0 * s_d[2];
    double ptp[4], ptm[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) {
        ptp[c] = x[c] + Tn[c];
        ptm[c] = x[c] - Tn[c];
    }
    const double lon_r[3] = {x[0] * kDeg2Rad, ptp[0] * kDeg2Rad, ptm[0] * kDeg2Rad};
    const double lat_r[3] = {lat0, ptp[1] * kDeg2Rad, ptm[1] * kDeg2Rad};
    const double vsp[3] = {sp0, fma(sp0, c_d[0], p2), fma(sp0, c_d[0], -p2)};
    const double vcp[3] = {cp0, fma(cp0, c_d[0], -p4), fma(cp0, c_d[0], p4)};
    const double vsa[3] = {sa0, fma(sa0, c_d[1], a2), fma(sa0, c_d[1], -a2)};
    const double vca[3] = {ca0, fma(ca0, c_d[1], -a4), fma(ca0, c_d[1], a4)};
    const double vsd[3] = {sd0, fma(sd0, c_d[2], d2), fma(sd0, c_d[2], -d2)};
    const double vcd[3] = {cd0, fma(cd0, c_d[2], -d4), fma(cd0, c_d[2], d4)};
    double lon_o[3], lat_o[3];
    geodetic_finish_n<3, KG>(lon_r, lat_r, vsp, vcp, vsa, vca, vsd, vcd, lon_o, lat_o, ok, gk);
    // a lane whose inputs are already non-finite fails every range test but ends in NaN on either path: it must not send
    // its wave through the slow one (see lane_predict)
    const double fin = (dt + sr + cr + x[0] + x[1] + x[2] + x[3] + Tn[0] + Tn[1] + Tn[2] + Tn[3]) * 0.0;
    if (__builtin_expect(__any(!ok && fin == 0.0), 0)) {
        quad_propagate_branching(x, Tn, dt, sr, cr, s0, sp, sm);
        return;
    }
    s0[0] = lon_o[0];
    s0[1] = lat_o[0];
    s0[2] = x[2] + du;
    s0[3] = fma(alpha0, kRad2Deg, da);
    sp[0] = lon_o[1];
    sp[1] = lat_o[1];
    sp[2] = ptp[2] + du;
    sp[3] = fma(ptp[3] * kDeg2Rad, kRad2Deg, da);
    sm[0] = lon_o[2];
    sm[1] = lat_o[2];
    sm[2] = ptm[2] + du;
    sm[3] = fma(ptm[3] * kDeg2Rad, kRad2Deg, da);
}

}  // namespace ste

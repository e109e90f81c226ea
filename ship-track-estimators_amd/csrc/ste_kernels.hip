// ste_kernels.hip — batched UKF forward pass and unscented RTS smoother for gfx950 (MI355X), plus the C ABI of
// include/ste.h.  fp64 throughout; no MFMA (4x4 contractions); the whole per-track state lives in VGPRs.  No kernel of this
// file declares LDS or synchronises through it (the quad forward kernel exchanges rows by DPP); the quad kernels with the
// smoother's rows or the robust loop do carry 4 096 B of compiler-allocated LDS each -- hipcc promotes a small
// dynamically indexed private array (the q-dependent row stores) to LDS instead of scratch (SQ_INSTS_LDS in
// profiles/r03_pmc_counters_per_launch.csv).
//
// Kernels:
//   ukf_forward_l1 / ukf_forward_q4      forward filter, one lane or one DPP quad per track (chosen by batch size or by
//                                        flag); with rts_work they also leave the smoother's cross-covariance D and,
//                                        where it does not follow from the history, its x_b and P_b
//   ukf_forward_sched + sched_gate       the forward passes of MANY batches / fleet windows as one launch of resident waves
//     + sched_upload                     working through a host-made schedule of (64-track tile, time slice) items -- the
//                                        body is ukf_forward_l1's --, the one-wave gate a smoother's stream waits behind,
//                                        and the upload of the schedule's table from page-locked memory
//   urtss_recur_l1                       smoother from those rows, one lane per track: gain K = D pinv(P_b), recurrence
//                                        (batches that fill the chip)
//   urtss_recur_sched                    the same body for every window of a scheduled forward launch as one launch: a wave
//                                        per tile, waiting for its own tile's forward pass
//   urtss_gains_all + urtss_recur_lean   the same smoother in two kernels for batches of <= 4 096 tracks: every gain of
//     / urtss_recur_lean_q4              every (step, track) at once, written back into the work rows, then the bare
//                                        recurrence, one lane or one DPP quad per track -- bit-identical to urtss_recur_l1
//   urtss_backward_l1                    stand-alone smoother that recomputes everything (rts_work == NULL)
//   predict / update / robust_terms / geodetic / sigma_points kernels   single-step API parity
// All per-step inputs/outputs are SoA with the track index fastest, so a wave's accesses are contiguous runs.
//
// Reference semantics (paths relative to /root/reference/src/track_estimators/kalman_filters/):
//   forward  : kalman_filter.py:61-117 (driver), unscented.py:178-207 (predict), :219-265 (update)
//   backward : unscented.py:285-351 (rts_step)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/ste.h"
#include "ste_err.h"
#include "ste_math.h"
#include "ste_lane.h"
#include "ste_quad.h"

namespace ste {

struct KParams {
    int B, Nmax, Tmax;
    unsigned flags;
    int tuning;
    int fast_upd;  // H = diag(1, 1, 0, 0), R confined to the same block: closed-form update (and closed-form robust rescaling)
    Mats m;
    const int32_t* nsteps;
    const double* x0;
    const double* P0;
    const double* dt;
    const double* sog_rate;
    const double* cog_rate;
    const double* sog_rate_rts;
    const double* cog_rate_rts;
    const int32_t* upd_idx;
    const double* z;
    const double* noise_pred;
    const double* noise_upd;
    const double* noise_rts;
    double* fwd_mean;
    double* fwd_cov;
    double* sm_mean;
    double* sm_cov;
    int32_t* status;
    double* rts_work;  // [Nmax][kWorkElems][ld] smoother gains produced by the forward pass, or nullptr
    int ld;            // tracks per row of every per-track array (ste.h: track_stride; = B for a batch of its own)
    int k0;            // forward pass, time slices: absolute index of this launch's step 0 (0 for a whole pass); every
                       // per-step pointer above already names row k0, Nmax is the slice's length (see slice_params)
    int qpw;           // quad forward kernel: quads (tracks) per wave, 1 .. 16 (launch_forward: fewer when waves are scarce)
    double* first_bad; // [ld] the last row of rts_work (first bad square root per track), or nullptr
    double* sm_pos;    // [Nmax+1][2][ld] smoothed lon / lat beside sm_mean, or nullptr
};

// ste.h flags are 8 bits wide; this one is set by slice_params only: the launch continues a forward pass at step k0 > 0
constexpr unsigned kFlagContinue = 0x10000u;

// A kernel argument fetched where it is used (volatile: neither merged with an earlier load of the same word nor hoisted).
// The forward kernels sit at the edge of both register files; an argument kept live across the step loop for one rare use
// costs spills inside the loop (with the slice offset held in a scalar register: 62 instead of 12 lane reads per step).
__device__ __forceinline__ int late_k0() {
    typedef const char __attribute__((address_space(4))) * kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    return *(const volatile int __attribute__((address_space(4)))*)(ka + offsetof(KParams, k0));
}

// The scheduled forward kernel (ukf_forward_sched) has no KParams in its argument segment: the work item's slice offset is
// parked in a word of LDS (one wave per workgroup) and read back where it is used, for the same reason.
__shared__ int g_sched_word[2];  // [0] the slice offset k0 of the item being run, [1] the wave's position in its item list
__device__ __forceinline__ int sched_k0() { return *(volatile int*)&g_sched_word[0]; }

constexpr int kColdEvery = 64;  // power of two

// rts_work row layout (round 3): columns 0-1 of D, row-major 4 x 2 (8) | x_b (4) | P_b upper triangle, row-major (10) |
// columns 2-3 of D, row-major 4 x 2 (8).
// The smoother's step k (unscented.py:297-333) starts from the same filtered state as the forward predict of step k, with
// the same dt and rates, so its fan, its back-prediction x_b, its P_b and its cross-covariance D are values the predict
// already holds: the forward kernels leave them here and the backward pass is the gain solve K = D pinv(P_b) (:333) plus
// the recurrence (:337-349).  What is written is only what cannot be had cheaper:
//   * x_b and P_b only for the steps where they do not follow from the filtered history -- steps followed by a
//     measurement update, row 0 when the run starts with an update, every step of a run with recorded noise; elsewhere
//     x_b = fwd_mean[k + 1] and P_b = fwd_cov[k + 1] + b b^T with b = fwd_mean[k + 1] - fwd_mean[k];
//   * columns 2-3 of D only at and after a track's first clamped / unconverged square root: speed and heading pass
//     through the process model with unit slope (non_linear_process.py:74-75), so D[:, 2:4] = 2 wi (T T)[:, 2:4], which for
//     an exact T = sqrtm(scale P_k) is (2 wi scale) P_k[:, 2:4] -- the filtered covariance the smoother reads anyway.
//     The step index of that first bad square root is kept, as a double, in the B words that follow the Nmax rows
//     (kNeverBad when there is none): the smoother never looks at status[], one kernel smooths every track.
constexpr int kWorkD = 0, kWorkXb = 8, kWorkPb = 12, kWorkD23 = 22, kWorkElems = STE_RTS_WORK_ROWS;
static_assert(kWorkElems == 30, "include/ste.h: STE_RTS_WORK_ROWS");
constexpr double kNeverBad = 1e300;

__device__ __forceinline__ void load_mat(const double* base, size_t row, size_t B, size_t t, double (&M)[4][4]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) M[r][c] = base[(row * 16 + r * 4 + c) * B + t];
    }
}
// Histories and work rows are written once and read by another kernel milliseconds later: nontemporal stores (no change at
// round 2's 1.01 ms per step; +1.2 % now that the pipeline moves 3.7 TB/s: 7.48 -> 7.57e9 track-steps/s, same box, twice;
// nontemporal loads on the smoother's side: nothing).
__device__ __forceinline__ void st_stream(double* p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void store_mat(double* base, size_t row, size_t B, size_t t, const double (&M)[4][4]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) st_stream(&base[(row * 16 + r * 4 + c) * B + t], M[r][c]);
    }
}
__device__ __forceinline__ void load_vec(const double* base, size_t row, size_t B, size_t t, double (&v)[4]) {
    STE_UNROLL
    for (int c = 0; c < 4; ++c) v[c] = base[(row * 4 + c) * B + t];
}
__device__ __forceinline__ void store_vec(double* base, size_t row, size_t B, size_t t, const double (&v)[4]) {
    STE_UNROLL
    for (int c = 0; c < 4; ++c) st_stream(&base[(row * 4 + c) * B + t], v[c]);
}

// Covariance histories (fwd_cov, sm_cov) come in two layouts: [row][16][B] full 4 x 4 matrices, the reference's return
// shape, or -- STE_FLAG_PACKED_COV -- [row][10][B] upper triangles (row-major: 00 01 02 03 11 12 13 22 23 33).  The
// matrices are symmetric by construction, so the packed form loses nothing; it takes 48 of 128 bytes off every history
// row written and lets the host expand on its way out (DeviceBatch.download).
__device__ __forceinline__ size_t cov_at(bool packed, size_t row, int r, int c) {  // r <= c
    return packed ? row * 10 + (size_t)(r * 4 - (r * (r - 1)) / 2 + (c - r)) : row * 16 + (size_t)(r * 4 + c);
}
// packed symmetric -> history row
__device__ __forceinline__ void store_cov_p(double* base, bool packed, size_t row, size_t B, size_t t, const double (&P)[10]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) st_stream(&base[cov_at(packed, row, r, c) * B + t], P[tix(r, c)]);
    }
    if (!packed) {
        STE_UNROLL
        for (int r = 1; r < 4; ++r) {
            STE_UNROLL
            for (int c = 0; c < r; ++c) st_stream(&base[(row * 16 + r * 4 + c) * B + t], P[tix(r, c)]);
        }
    }
}
__device__ __forceinline__ void load_cov_p(const double* base, bool packed, size_t row, size_t B, size_t t, double (&P)[10]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) P[tix(r, c)] = base[cov_at(packed, row, r, c) * B + t];
    }
}
// full 4 x 4 <-> history row (the literal kernels); the packed layout keeps the upper triangle
__device__ __forceinline__ void store_cov_m(double* base, bool packed, size_t row, size_t B, size_t t, const double (&M)[4][4]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            if (c >= r)
                st_stream(&base[cov_at(packed, row, r, c) * B + t], M[r][c]);
            else if (!packed)
                st_stream(&base[(row * 16 + r * 4 + c) * B + t], M[r][c]);
        }
    }
}
__device__ __forceinline__ void load_cov_m(const double* base, bool packed, size_t row, size_t B, size_t t, double (&M)[4][4]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            if (packed)
                M[r][c] = base[cov_at(true, row, r < c ? r : c, r < c ? c : r) * B + t];
            else
                M[r][c] = base[(row * 16 + r * 4 + c) * B + t];
        }
    }
}

__device__ __forceinline__ bool all_finite(const double (&x)[4], const double (&P)[4][4]) {
    double acc = 0.0;
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        acc += x[r] * 0.0;
        STE_UNROLL
        for (int c = 0; c < 4; ++c) acc += P[r][c] * 0.0;
    }
    return acc == 0.0;  // inf*0 and nan*0 are NaN
}

// Sigma fan of (x, P): dev[i] = column i of sqrtm(scale*P) (unscented.py:95-105).  chi_{i+1} = x + dev[i],
// chi_{i+1+n} = x - dev[i], chi_0 = x.  T is symmetric, so column i == row i.
template <bool kWarm>
__device__ __forceinline__ int sigma_fan(const double (&x)[4], const double (&P)[4][4], double scale,
                                         double (&sig)[9][4], EigBasis& basis);

__device__ __forceinline__ int sigma_fan(const double (&x)[4], const double (&P)[4][4], double scale,
                                         double (&sig)[9][4]) {
    EigBasis none;
    return sigma_fan<false>(x, P, scale, sig, none);
}

template <bool kWarm>
__device__ __forceinline__ int sigma_fan(const double (&x)[4], const double (&P)[4][4], double scale,
                                         double (&sig)[9][4], EigBasis& basis) {
    double T[4][4];
    const int st = sym_sqrt4<kWarm>(P, scale, T, basis);
    if (kWarm) basis.valid = true;
    STE_UNROLL
    for (int c = 0; c < 4; ++c) sig[0][c] = x[c];
    STE_UNROLL
    for (int i = 0; i < 4; ++i) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            sig[1 + i][c] = x[c] + T[c][i];
            sig[5 + i][c] = x[c] - T[c][i];
        }
    }
    return st;
}

// Sigma fan of (x, P) pushed through the process model (unscented.py:183-191 = :301-312):
//   sig0[j]  the fan itself: x, x + col_i(T), x - col_i(T) with T = sqrtm(scale * P)
//   sig[j]   geodetic_dynamics(sig0[j], dt, sog_rate, cog_rate)
// The 2n points come in +- pairs around the centre, so their angles are (centre angle) +- (small increment) and
// sin/cos of all of them follow from the centre's three sincos and one sincos per increment by the angle-addition
// formulas: 15 sincos evaluations instead of 27, 12 of them on small arguments that need no range reduction.
__device__ __noinline__ void propagate_fan_branching(const double (&x)[4], const double (&T)[4][4], double dt, double sr,
                                                        double cr, double (&sig0)[9][4], double (&sig)[9][4]) {
    const double dt_r = dt / kEarthRadius;
    const double du = sr * dt, da = cr * dt;
    // centre
    const double lat0 = x[1] * kDeg2Rad, alpha0 = x[3] * kDeg2Rad, delta0 = x[2] * dt_r;
    double sp0, cp0, sa0, ca0, sd0, cd0;
    sincos_fast(lat0, sp0, cp0);
    sincos_fast(alpha0, sa0, ca0);
    sincos_fast(delta0, sd0, cd0);
    STE_UNROLL
    for (int c = 0; c < 4; ++c) sig0[0][c] = x[c];
    geodetic_finish(x[0] * kDeg2Rad, lat0, sp0, cp0, sa0, ca0, sd0, cd0, sig[0][0], sig[0][1]);
    sig[0][2] = x[2] + du;
    sig[0][3] = alpha0 * kRad2Deg + da;
    STE_UNROLL
    for (int i = 0; i < 4; ++i) {
        // increments of the three angles along column i of T (T symmetric: T[c][i] == T[i][c])
        double sdp, cdp, sda, cda, sdd, cdd;
        sincos_delta(T[1][i] * kDeg2Rad, sdp, cdp);
        sincos_delta(T[3][i] * kDeg2Rad, sda, cda);
        sincos_delta(T[2][i] * dt_r, sdd, cdd);
        const double p1 = sp0 * cdp, p2 = cp0 * sdp, p3 = cp0 * cdp, p4 = sp0 * sdp;
        const double a1 = sa0 * cda, a2 = ca0 * sda, a3 = ca0 * cda, a4 = sa0 * sda;
        const double d1 = sd0 * cdd, d2 = cd0 * sdd, d3 = cd0 * cdd, d4 = sd0 * sdd;
        STE_UNROLL
        for (int sgn = 0; sgn < 2; ++sgn) {
            const int j = 1 + i + 4 * sgn;
            double pt[4];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) pt[c] = sgn ? x[c] - T[c][i] : x[c] + T[c][i];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) sig0[j][c] = pt[c];
            const double sp = sgn ? p1 - p2 : p1 + p2, cp = sgn ? p3 + p4 : p3 - p4;
            const double sa = sgn ? a1 - a2 : a1 + a2, ca = sgn ? a3 + a4 : a3 - a4;
            const double sd = sgn ? d1 - d2 : d1 + d2, cd = sgn ? d3 + d4 : d3 - d4;
            geodetic_finish(pt[0] * kDeg2Rad, pt[1] * kDeg2Rad, sp, cp, sa, ca, sd, cd, sig[j][0], sig[j][1]);
            sig[j][2] = pt[2] + du;
            sig[j][3] = (pt[3] * kDeg2Rad) * kRad2Deg + da;
        }
    }
}

// The same fan with the several-at-a-time, branch-free helpers of ste_math.h (each polynomial coefficient is fetched once
// for the three angles of the centre, once per direction for its three increments, and once per direction for its +/-
// pair); a wave in which some lane leaves their validity range redoes the step with the branching version above.
__device__ __forceinline__ void propagate_points(const double (&x)[4], const double (&T)[4][4], double dt, double sr,
                                                 double cr, double (&sig0)[9][4], double (&sig)[9][4]) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    const double dt_r = div_earth_radius(dt);

    const double du = sr * dt, da = cr * dt;
    bool ok = true;
    const double lat0 = x[1] * kDeg2Rad, alpha0 = x[3] * kDeg2Rad, delta0 = x[2] * dt_r;
    const double a0[3] = {lat0, alpha0, delta0};
    double s_0[3], c_0[3];
    sincos_fast_n<3>(a0, s_0, c_0, ok);
    const double sp0 = s_0[0], cp0 = c_0[0], sa0 = s_0[1], ca0 = c_0[1], sd0 = s_0[2], cd0 = c_0[2];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) sig0[0][c] = x[c];
    {
        const double lo[1] = {x[0] * kDeg2Rad}, la[1] = {lat0}, vsp[1] = {sp0}, vcp[1] = {cp0}, vsa[1] = {sa0},
                     vca[1] = {ca0}, vsd[1] = {sd0}, vcd[1] = {cd0};
        double lon_o[1], lat_o[1];
        geodetic_finish_n<1>(lo, la, vsp, vcp, vsa, vca, vsd, vcd, lon_o, lat_o, ok);
        sig[0][0] = lon_o[0];
        sig[0][1] = lat_o[0];
    }
    sig[0][2] = x[2] + du;
    sig[0][3] = fma(alpha0, kRad2Deg, da);
    STE_UNROLL
    for (int i = 0; i < 4; ++i) {
        const double dl[3] = {T[1][i] * kDeg2Rad, T[3][i] * kDeg2Rad, T[2][i] * dt_r};
        double s_d[3], c_d[3];
        sincos_delta_n<3>(dl, s_d, c_d, ok);
        const double p2 = cp0 * s_d[0], p4 = sp0 * s_d[0], a2 = ca0 * s_d[1], a4 = sa0 * s_d[1], d2 = cd0 * s_d[2],
                     d4 = sd0 * s_d[2];
        double ptp[4], ptm[4];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            ptp[c] = x[c] + T[c][i];
            ptm[c] = x[c] - T[c][i];
            sig0[1 + i][c] = ptp[c];
            sig0[5 + i][c] = ptm[c];
        }
        const double lo[2] = {ptp[0] * kDeg2Rad, ptm[0] * kDeg2Rad}, la[2] = {ptp[1] * kDeg2Rad, ptm[1] * kDeg2Rad};
        const double vsp[2] = {fma(sp0, c_d[0], p2), fma(sp0, c_d[0], -p2)};
        const double vcp[2] = {fma(cp0, c_d[0], -p4), fma(cp0, c_d[0], p4)};
        const double vsa[2] = {fma(sa0, c_d[1], a2), fma(sa0, c_d[1], -a2)};
        const double vca[2] = {fma(ca0, c_d[1], -a4), fma(ca0, c_d[1], a4)};
        const double vsd[2] = {fma(sd0, c_d[2], d2), fma(sd0, c_d[2], -d2)};
        const double vcd[2] = {fma(cd0, c_d[2], -d4), fma(cd0, c_d[2], d4)};
        double lon_o[2], lat_o[2];
        geodetic_finish_n<2>(lo, la, vsp, vcp, vsa, vca, vsd, vcd, lon_o, lat_o, ok);
        sig[1 + i][0] = lon_o[0];
        sig[1 + i][1] = lat_o[0];
        sig[1 + i][2] = ptp[2] + du;
        sig[1 + i][3] = fma(ptp[3] * kDeg2Rad, kRad2Deg, da);
        sig[5 + i][0] = lon_o[1];
        sig[5 + i][1] = lat_o[1];
        sig[5 + i][2] = ptm[2] + du;
        sig[5 + i][3] = fma(ptm[3] * kDeg2Rad, kRad2Deg, da);
    }
    double fin = dt + sr + cr;  // non-finite lanes end in NaN on either path: they do not ask for the slow one
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        fin += x[r];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) fin += T[r][c];
    }
    if (__builtin_expect(__any(!ok && fin * 0.0 == 0.0), 0)) {
        // Out of line, through copies: the branching version calls the device library (Payne-Hanek sincos, atan2, asin) and
        // is ~55 KB of code; inlined it tripled this kernel's size and register pressure.  Its arguments are address-taken,
        // so they are private copies here and the hot path's arrays stay in registers.
        double xc[4], Tc[4][4], s0c[9][4], sc[9][4];
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            xc[r] = x[r];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) Tc[r][c] = T[r][c];
        }
        propagate_fan_branching(xc, Tc, dt, sr, cr, s0c, sc);
        STE_UNROLL
        for (int j = 0; j < 9; ++j) {
            STE_UNROLL
            for (int c = 0; c < 4; ++c) {
                sig0[j][c] = s0c[j][c];
                sig[j][c] = sc[j][c];
            }
        }
    }
}


template <bool kWarm>
__device__ __forceinline__ int propagate_fan(const double (&x)[4], const double (&P)[4][4], double scale, double dt,
                                             double sr, double cr, double (&sig0)[9][4], double (&sig)[9][4],
                                             EigBasis& basis) {
    double T[4][4];
    const int st = sym_sqrt4<kWarm>(P, scale, T, basis);
    if (kWarm) basis.valid = true;
    propagate_points(x, T, dt, sr, cr, sig0, sig);
    return st;
}

// sum_j W_j a_j b_j^T  for two sets of 9 deviation vectors.
template <bool kSym>
__device__ __forceinline__ void weighted_outer(const double (&a)[9][4], const double (&b)[9][4], double w0, double wi,
                                               double (&out)[4][4]) {
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = kSym ? r : 0; c < 4; ++c) {
            double acc = 0.0;
            STE_UNROLL
            for (int j = 1; j < 9; ++j) acc = fma(a[j][r], b[j][c], acc);
            acc = fma(w0 * a[0][r], b[0][c], wi * acc);
            out[r][c] = acc;
            if (kSym) out[c][r] = acc;
        }
    }
}

// UKF predict (unscented.py:178-207), the literal form: the whole fan, the weighted mean, the weighted outer products.
// x, P updated in place.  Used by the single-step entry point (ste_ukf_predict_f64); the forward kernels run the streamed
// form of the same arithmetic (lane_predict / quad_predict).
__device__ __forceinline__ int ukf_predict(const Mats& p, double (&x)[4], double (&P)[4][4], double dt, double sr,
                                           double cr, const double* noise, size_t nrow, size_t B, size_t t) {
    double sig[9][4], sig0[9][4], T[4][4];
    EigBasis none;
    const int st = sym_sqrt4<false>(P, p.fan_scale, T, none);
    propagate_points(x, T, dt, sr, cr, sig0, sig);
    double xp[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) {
        double acc = 0.0;
        STE_UNROLL
        for (int j = 1; j < 9; ++j) acc += sig[j][c];
        xp[c] = fma(p.w0, sig[0][c], p.wi * acc);
    }
    if (noise) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) xp[c] += noise[(nrow * 4 + c) * B + t];
    }
    STE_UNROLL
    for (int j = 0; j < 9; ++j) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) sig[j][c] -= xp[c];
    }
    double Pn[4][4];
    weighted_outer<true>(sig, sig, p.w0, p.wi, Pn);
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        x[r] = xp[r];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) P[r][c] = Pn[r][c] + p.Q[r * 4 + c];
    }
    return st;
}

// Chang (2014) Mahalanobis terms of the robustification helpers (unscented.py:389-483), with the reference's y = z - x:
//   gamma = | y^T (H P H^T + R)^+ y |          criterion_index, :420-426
//   denom =   y^T (S^+ R S^+) y                  denominator of update_lambda_factor, :468-478
__device__ __forceinline__ int robust_terms(const double (&H)[4][4], const double (&R)[4][4], const double (&x)[4],
                                            const double (&P)[4][4], const double (&z)[4], double& gamma,
                                            double& denom) {
    double HP[4][4], S[4][4], Si[4][4], y[4], u[4], v[4];
    mm(H, P, HP);
    mmt(HP, H, S);
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        y[r] = z[r] - x[r];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) S[r][c] += R[r][c];
    }
    const int st = sym_pinv4(S, Si);
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        double acc = 0.0;
        STE_UNROLL
        for (int c = 0; c < 4; ++c) acc = fma(Si[r][c], y[c], acc);
        u[r] = acc;  // S^+ y
    }
    double g = 0.0, d = 0.0;
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        g = fma(y[r], u[r], g);
        double acc = 0.0;
        STE_UNROLL
        for (int c = 0; c < 4; ++c) acc = fma(R[r][c], u[c], acc);
        v[r] = acc;  // R S^+ y
    }
    STE_UNROLL
    for (int r = 0; r < 4; ++r) d = fma(u[r], v[r], d);  // (S^+ y)^T R (S^+ y) = y^T S^+ R S^+ y  (S^+ symmetric)
    gamma = fabs(g);
    denom = d;
    return st;
}

// Linear Kalman update with pseudo-inverse gain and Joseph-form covariance (unscented.py:219-265).
__device__ __forceinline__ int ukf_update(const Mats& p, double (&x)[4], double (&P)[4][4], const double (&zin)[4],
                                          const double* noise, size_t nrow, size_t B, size_t t) {
    double H[4][4], R[4][4];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            H[r][c] = p.H[r * 4 + c];
            R[r][c] = p.R[r * 4 + c];
        }
    }
    double z[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) z[c] = zin[c];
    int rst = 0;
    if (p.robust_iters > 0) {
        // Opt-in robustification (check_robustness, unscented.py:353-387) on the un-noised observation: while the
        // criterion exceeds chi_alpha, lambda += (gamma - chi)/denom and R <- lambda R (compounding, as written there).
        double gamma, denom, lambda = 1.0;
        rst |= robust_terms(H, R, x, P, z, gamma, denom);
        for (int it = 0; it < p.robust_iters && gamma > p.chi_alpha; ++it) {
            lambda += (gamma - p.chi_alpha) / denom;
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) R[r][c] *= lambda;
            }
            rst |= robust_terms(H, R, x, P, z, gamma, denom);
        }
        if (gamma > p.chi_alpha) rst |= STE_STATUS_ROBUST_CAP;
    }
    if (noise) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) z[c] += noise[(nrow * 4 + c) * B + t];
    }
    double HP[4][4], S[4][4], Si[4][4], PHt[4][4], K[4][4];
    mm(H, P, HP);
    mmt(HP, H, S);
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) S[r][c] += R[r][c];
    }
    int st = rst;
    {
        bool blk = true;  // S confined to its leading 2 x 2 block (exact zeros elsewhere), for every track of the wave
        double fin = 0.0;
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = 0; c < 4; ++c) {
                if (r >= 2 || c >= 2) blk = blk && (S[r][c] == 0.0);
                fin += S[r][c];
            }
        }
        blk = blk || !(fin * 0.0 == 0.0);  // a non-finite S is NaN on either route: no veto against the fast one
        if (__all(blk))
            sym_pinv4_block2(S, Si);
        else
            st |= sym_pinv4(S, Si);
    }
    mmt(P, H, PHt);
    mm(PHt, Si, K);
    double y[4];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        double hx = H[r][0] * x[0];
        STE_UNROLL
        for (int c = 1; c < 4; ++c) hx = fma(H[r][c], x[c], hx);
        y[r] = z[r] - hx;
    }
    y[3] = wrap180(y[3]);
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        double acc = x[r];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) acc = fma(K[r][c], y[c], acc);
        x[r] = acc;
    }
    x[3] = floored_mod(x[3], 360.0);
    double A[4][4], AP[4][4], KR[4][4], P1[4][4], P2[4][4];
    mm(K, H, A);
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) A[r][c] = ((r == c) ? 1.0 : 0.0) - A[r][c];
    }
    mm(A, P, AP);
    mmt_sym(AP, A, P1);
    mm(K, R, KR);
    mmt_sym(KR, K, P2);
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) P[r][c] = P1[r][c] + P2[r][c];
    }
    return st;
}

// ---------------------------------------------------------------------------------------------------------------
// forward pass, one lane per track (ste_lane.h)
// ---------------------------------------------------------------------------------------------------------------
// UKF predict (unscented.py:178-207) on packed (x, P), the sigma pairs streamed through running moments; with `work` it
// also leaves the smoother's row of this step (see kWorkD).  x, P are replaced by the predicted mean (+ recorded noise)
// and covariance.  `flagged` is sticky: set once a square root of this track was clamped or did not converge; from that
// step on columns 2-3 of D are stored as well.
template <bool kGains, bool kSched = false>
__device__ __forceinline__ int lane_predict(const Mats& p, double (&x)[4], double (&P)[10], double (&V)[4][4], bool warm,
                                            double dt, double sr, double cr, const double* noise,
                                            const double* noise_rts, size_t nrow, size_t B, size_t t, double* work,
                                            bool full_row, bool noise_mode, bool& flagged, double* first_bad,
                                            const TrigReg& tk, const double (&Qv)[10]) {
#pragma clang fp contract(off)  // explicit fma() only: the same bits in every kernel this is inlined into
    double T[10];
    const int st = sym_sqrt_p(P, p.fan_scale, T, V, warm);
    bool ok = true;
    FanCentre g;
    fan_centre(x, dt, sr, cr, g, ok, tk);
    FanMoments f;
    moments_clear(f);
    {
        double lonp, latp, lonm, latm;
        fan_pair<0>(x, T, g, lonp, latp, lonm, latm, ok, tk);
        moments_add<0, kGains>(f, T, g.c[0], g.c[1], lonp, latp, lonm, latm);
        fan_pair<1>(x, T, g, lonp, latp, lonm, latm, ok, tk);
        moments_add<1, kGains>(f, T, g.c[0], g.c[1], lonp, latp, lonm, latm);
        fan_pair<2>(x, T, g, lonp, latp, lonm, latm, ok, tk);
        moments_add<2, kGains>(f, T, g.c[0], g.c[1], lonp, latp, lonm, latm);
        fan_pair<3>(x, T, g, lonp, latp, lonm, latm, ok, tk);
        moments_add<3, kGains>(f, T, g.c[0], g.c[1], lonp, latp, lonm, latm);
    }
    // A lane whose inputs are already non-finite fails every range test, but its outcome is NaN on either path: it must not
    // send its whole wave through the slow one (BASELINE configs[3]: five of seven ships go non-finite early -- duplicate
    // timestamps -- and dragged the two healthy ones along at a tenth of the speed, step after step).
    double fin = dt + sr + cr;
    STE_UNROLL
    for (int c = 0; c < 4; ++c) fin += x[c];
    STE_UNROLL
    for (int e = 0; e < 10; ++e) fin += T[e];
    const bool finite_in = fin * 0.0 == 0.0;
    if (__builtin_expect(__any(!ok && finite_in), 0)) {
        // Some lane left the validity range of the branch-free transcendentals (a pole, a step of tens of degrees):
        // the whole fan again with the branching version (device-library sincos / atan2 / asin, out
        // of line, through copies so that the hot path's arrays stay in registers), same formulas for the lanes that
        // were fine, then the same moment sums.
        double xc[4], Tc[4][4], s0c[9][4], sc[9][4];
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            xc[r] = x[r];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) Tc[r][c] = T[tix(r, c)];
        }
        propagate_fan_branching(xc, Tc, dt, sr, cr, s0c, sc);
        STE_UNROLL
        for (int c = 0; c < 4; ++c) g.c[c] = sc[0][c];
        moments_clear(f);
        moments_add<0, kGains>(f, T, g.c[0], g.c[1], sc[1][0], sc[1][1], sc[5][0], sc[5][1]);
        moments_add<1, kGains>(f, T, g.c[0], g.c[1], sc[2][0], sc[2][1], sc[6][0], sc[6][1]);
        moments_add<2, kGains>(f, T, g.c[0], g.c[1], sc[3][0], sc[3][1], sc[7][0], sc[7][1]);
        moments_add<3, kGains>(f, T, g.c[0], g.c[1], sc[4][0], sc[4][1], sc[8][0], sc[8][1]);
    }
    // weighted mean and covariance about it: the weights sum to one (make_params refuses anything else), the centre's
    // deviation is zero, so mean = c' + wi s and cov = wi S - (wi s)(wi s)^T
    const double d0 = p.wi * f.s0, d1 = p.wi * f.s1;
    const double m[4] = {g.c[0] + d0, g.c[1] + d1, g.c[2], g.c[3]};
    double xp[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) xp[c] = m[c];
    if (noise) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) xp[c] += noise[(nrow * 4 + c) * B + t];
    }
    const double two_wi = p.wi + p.wi;
    double Pn[10];
    Pn[tix(0, 0)] = fma(-d0, d0, p.wi * f.S[tix(0, 0)]);
    Pn[tix(0, 1)] = fma(-d0, d1, p.wi * f.S[tix(0, 1)]);
    Pn[tix(1, 1)] = fma(-d1, d1, p.wi * f.S[tix(1, 1)]);
    Pn[tix(0, 2)] = p.wi * f.S[tix(0, 2)];
    Pn[tix(0, 3)] = p.wi * f.S[tix(0, 3)];
    Pn[tix(1, 2)] = p.wi * f.S[tix(1, 2)];
    Pn[tix(1, 3)] = p.wi * f.S[tix(1, 3)];
    Pn[tix(2, 2)] = two_wi * f.S[tix(2, 2)];
    Pn[tix(2, 3)] = two_wi * f.S[tix(2, 3)];
    Pn[tix(3, 3)] = two_wi * f.S[tix(3, 3)];
    if (kGains && work) {
        // The smoother's cross-covariance D = wi sum_i T_i (chi'_{i+} - chi'_{i-})^T (the centre's deviation from x_k is
        // zero and x_b cancels in the difference): rows 2-3 of its columns 0-1 are the cross moments S[c][2], S[c][3] just
        // formed, i.e. Pn before Q is added.
        double* w = work + (nrow * kWorkElems) * B + t;
        st_stream(&w[(kWorkD + 0) * B], p.wi * f.Dn[0][0]);
        st_stream(&w[(kWorkD + 1) * B], p.wi * f.Dn[0][1]);
        st_stream(&w[(kWorkD + 2) * B], p.wi * f.Dn[1][0]);
        st_stream(&w[(kWorkD + 3) * B], p.wi * f.Dn[1][1]);
        if (noise_mode) {  // otherwise the smoother takes rows 2-3 from P_b: D[2:4, 0:2] = (P_b - b b^T - Q)[0:2, 2:4]^T
            st_stream(&w[(kWorkD + 4) * B], Pn[tix(0, 2)]);
            st_stream(&w[(kWorkD + 5) * B], Pn[tix(1, 2)]);
            st_stream(&w[(kWorkD + 6) * B], Pn[tix(0, 3)]);
            st_stream(&w[(kWorkD + 7) * B], Pn[tix(1, 3)]);
        }
        const bool bad_now = (st & (STE_STATUS_CLAMPED | STE_STATUS_NOCONV)) != 0;
        if (bad_now && !flagged) *first_bad = (double)((long long)nrow + (kSched ? sched_k0() : late_k0()));  // absolute step index
        flagged = flagged || bad_now;
        if (flagged) {  // columns 2-3 of D = 2 wi (T T)[:, 2:4]: (2 wi scale) P_k[:, 2:4] only for an exact square root
            st_stream(&w[(kWorkD23 + 0) * B], two_wi * f.TT[0][0]);
            st_stream(&w[(kWorkD23 + 1) * B], two_wi * f.TT[0][1]);
            st_stream(&w[(kWorkD23 + 2) * B], two_wi * f.TT[1][0]);
            st_stream(&w[(kWorkD23 + 3) * B], two_wi * f.TT[1][1]);
            st_stream(&w[(kWorkD23 + 4) * B], Pn[tix(2, 2)]);
            st_stream(&w[(kWorkD23 + 5) * B], Pn[tix(2, 3)]);
            st_stream(&w[(kWorkD23 + 6) * B], Pn[tix(2, 3)]);
            st_stream(&w[(kWorkD23 + 7) * B], Pn[tix(3, 3)]);
        }
    }
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) Pn[tix(r, c)] += Qv[tix(r, c)];
    }
    if (noise) {  // the covariance is taken about the noised mean (unscented.py:203-205): + e e^T, e = mean - x^-
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = r; c < 4; ++c) Pn[tix(r, c)] = fma(m[r] - xp[r], m[c] - xp[c], Pn[tix(r, c)]);
        }
    }
    if (kGains && work && full_row) {
        // P_b is centred on the filtered mean x_k, not on the predicted one (unscented.py:324-325): with b = x^- - x_k,
        // e = (weighted mean) - x^-,  P_b = P^- + e b^T + b e^T + b b^T
        double* w = work + (nrow * kWorkElems) * B + t;
        double bv[4];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) bv[c] = xp[c] - x[c];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            double xb = m[c];
            if (noise_rts) xb += noise_rts[(nrow * 4 + c) * B + t];
            st_stream(&w[(kWorkXb + c) * B], xb);
        }
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = r; c < 4; ++c) {
                double v = fma(bv[r], bv[c], Pn[tix(r, c)]);
                if (noise) v += fma(m[r] - xp[r], bv[c], bv[r] * (m[c] - xp[c]));
                st_stream(&w[(kWorkPb + tix(r, c)) * B], v);
            }
        }
    }
    STE_UNROLL
    for (int c = 0; c < 4; ++c) x[c] = xp[c];
    STE_UNROLL
    for (int e = 0; e < 10; ++e) P[e] = Pn[e];
    return st;
}

// Measurement update on packed (x, P): the closed form for H = diag(1, 1, 0, 0) (kFastUpd, chosen by launch_forward from
// the matrices), otherwise the general 4x4 route (any H, R; the opt-in robust rescaling).
template <bool kFastUpd, bool kRobust = false>
__device__ __forceinline__ int lane_update(const Mats& p, double (&x)[4], double (&P)[10], const double (&zin)[4],
                                           const double* noise, size_t nrow, size_t B, size_t t) {
    if (kFastUpd) {
        double z[4];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) z[c] = zin[c];
        double r00 = p.R[0], r01 = p.R[1], r11 = p.R[5];
        int st = 0;
        if (kRobust) st = robust_rescale_sel2(p, x, P, z, r00, r01, r11);  // on the un-noised observation (unscented.py:228)
        if (noise) {
            STE_UNROLL
            for (int c = 0; c < 4; ++c) z[c] += noise[(nrow * 4 + c) * B + t];
        }
        lane_update_sel2(r00, r01, r11, x, P, z);
        return st;
    } else {
        double Pf[4][4];
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = 0; c < 4; ++c) Pf[r][c] = P[tix(r, c)];
        }
        const int st = ukf_update(p, x, Pf, zin, noise, nrow, B, t);
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = r; c < 4; ++c) P[tix(r, c)] = Pf[r][c];
        }
        return st;
    }
}

__device__ __forceinline__ void store_hist(const KParams& p, size_t row, size_t B, size_t t, const double (&x)[4],
                                           const double (&P)[10]) {
    STE_UNROLL
    for (int c = 0; c < 4; ++c) st_stream(&p.fwd_mean[(row * 4 + c) * B + t], x[c]);
    store_cov_p(p.fwd_cov, (p.flags & STE_FLAG_PACKED_COV) != 0, row, B, t, P);
}

// One wave per SIMD, on purpose: the step loop issues a vector instruction in ~80 % of its cycles, so a second forward
// wave on the same SIMD gains next to nothing (measured: two per SIMD run 1.55x slower each), while waves of several
// passes that the dispatcher doubles up on some SIMDs leave others empty.  The kernel holds 254 registers (its state,
// plus the sin / cos coefficients and Q kept in VGPRs) and amdgpu_waves_per_eu(1, 1) pads the allocation to 264: a second
// forward wave never fits on a SIMD, a smoother wave (234, allocation 240) does (264 + 240 <= 512).
// kRobust: the closed-form update with the closed-form robust rescaling in front of it (a template parameter, so that the
// default instantiation's step loop is exactly the one measured without it); with kFastUpd false the general route reads
// robust_iters itself.
template <bool kGains, bool kFastUpd, bool kRobust, bool kSched>
__device__ __forceinline__ void forward_tile_l1(const KParams& p, const size_t t) {
    const size_t B = (size_t)p.ld;  // row pitch of every per-track array (= the batch's own width unless it is a window)
    if (t >= (size_t)p.B) return;
    // A later time slice of a forward pass (slice_params): x0 / P0 name history row k0 -- the state the previous launch
    // left, bit for bit -- and every per-step pointer row k0; the slice boundary is a multiple of kColdEvery, where the
    // eigenvector basis restarts from the identity anyway, so nothing else has to be carried.
    const bool cont = (p.flags & kFlagContinue) != 0;
    const int ns = (p.nsteps ? p.nsteps[t] - p.k0 : p.Nmax);
    if (cont && ns <= 0) return;  // this track ended in an earlier slice: its rows and its status are final

    double x[4], P[10];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) x[c] = p.x0[c * B + t];
    if (cont && (p.flags & STE_FLAG_PACKED_COV)) {
        load_cov_p(p.P0, true, 0, B, t, P);
    } else {
        // the prior enters as (P0 + P0^T) / 2: the host refuses a P0 that is not symmetric (batch._as44)
        double Pf[4][4];
        if (p.flags & STE_FLAG_SHARED_P0) {
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) Pf[r][c] = p.P0[r * 4 + c];
            }
        } else {
            load_mat(p.P0, 0, B, t, Pf);
        }
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = r; c < 4; ++c) P[tix(r, c)] = 0.5 * (Pf[r][c] + Pf[c][r]);
        }
    }
    store_hist(p, 0, B, t, x, P);  // slot 0 = prior (kalman_filter.py:76-77); a later slice rewrites row k0 with its own bits

    // (a later slice may run in the same launch, on the same wave, as the one that wrote these two words -- the scheduled
    //  kernel -- without a cache invalidate in between: read them past the CU's L1)
    int st = cont ? (__hip_atomic_load(&p.status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~STE_STATUS_NAN) : 0;  // the NaN bit is taken from the final state, as in a whole pass
    const bool initial_update = !(p.flags & STE_FLAG_NO_INITIAL_UPDATE);
    const bool noise_mode = p.noise_pred || p.noise_upd || p.noise_rts;
    bool flagged = false;  // sticky: a square root of this track was clamped or did not converge
    double* first_bad = (kGains && p.rts_work) ? p.first_bad + t : nullptr;
    if (first_bad) {
        if (cont)
            flagged = __hip_atomic_load(first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kNeverBad;
        else
            *first_bad = kNeverBad;
    }
    TrigReg tk;  // sin / cos polynomial coefficients, in VGPRs for the whole kernel (ste_math.h)
    trig_reg_init(tk);
    double Qv[10];  // Q in VGPRs too: as kernel arguments its 20 words were spilled to VGPR lanes and read back every step
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = r; c < 4; ++c) Qv[tix(r, c)] = in_vgpr(p.m.Q[r * 4 + c]);
    }
    double V[4][4];
    if (kGains && initial_update && ns > 0) {
        // History row 0 is the PRIOR (kalman_filter.py:76-77) while the first predict starts from the state after the
        // initial update, so the smoother's step 0 (which reads row 0, unscented.py:301) needs its own fan: run the
        // predict arithmetic once on a copy of the prior, keep only the smoother's rows.
        double xc[4], Pc[10];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) xc[c] = x[c];
        STE_UNROLL
        for (int e = 0; e < 10; ++e) Pc[e] = P[e];
        st |= lane_predict<true, kSched>(p.m, xc, Pc, V, false, p.dt[t], p.sog_rate[t], p.cog_rate[t], nullptr, p.noise_rts, 0, B, t,
                                         p.rts_work, true, noise_mode, flagged, first_bad, tk, Qv);
    }
    if (initial_update) {
        double z0[4];
        load_vec(p.z, 0, B, t, z0);
        st |= lane_update<kFastUpd, kRobust>(p.m, x, P, z0, p.noise_upd, 0, B, t);  // kalman_filter.py:81
    }

    // The eigenvectors of the fan matrix carry over from step to step (warm start); restarted from the identity every
    // kColdEvery steps so that rounding in the accumulated rotations cannot build up over a long track.
    double dt_n = 0.0, sr_n = 0.0, cr_n = 0.0;
    int ui_n = -1;
    if (ns > 0) {
        dt_n = p.dt[t];
        sr_n = p.sog_rate[t];
        cr_n = p.cog_rate[t];
        ui_n = p.upd_idx[t];
    }
    for (int k = 0; k < p.Nmax; ++k) {
        const bool live = k < ns;
        if (!__any(live)) break;
        if (live) {
            const double dt = dt_n, sr = sr_n, cr = cr_n;
            const int ui = ui_n;
            // Unconditional loads with clamped indices: a load inside an `if` makes hipcc drain the queue with
            // s_waitcnt vmcnt(0) where the branch rejoins.  The observation row of a step without an update is row 0
            // (cache-resident); the inputs of step k + 1 stay in flight during this step's arithmetic.
            const bool ui_ok = ui < p.Tmax;  // an observation column past the padded batch: flagged, update skipped
            const bool upd = ui >= 0 && ui_ok;
            double zk[4];
            load_vec(p.z, (size_t)(upd ? ui : 0), B, t, zk);
            {
                const size_t o = (size_t)(k + 1 < ns ? k + 1 : k) * B + t;
                dt_n = p.dt[o];
                sr_n = p.sog_rate[o];
                cr_n = p.cog_rate[o];
                ui_n = p.upd_idx[o];
            }
            // row 0's smoother rows were taken from the prior above; every later row k is the state this predict starts from
            double* work = (kGains && !(k == 0 && initial_update)) ? p.rts_work : nullptr;
            const bool warm = (k & (kColdEvery - 1)) != 0;
            st |= lane_predict<kGains, kSched>(p.m, x, P, V, warm, dt, sr, cr, p.noise_pred, p.noise_rts, (size_t)k, B, t, work,
                                               upd || noise_mode, noise_mode, flagged, first_bad, tk, Qv);
            if (upd) st |= lane_update<kFastUpd, kRobust>(p.m, x, P, zk, p.noise_upd, (size_t)k + 1, B, t);
            if (!ui_ok) st |= STE_STATUS_BAD_INDEX;
            store_hist(p, (size_t)k + 1, B, t, x, P);
        }
    }
    double chk = 0.0;
    STE_UNROLL
    for (int c = 0; c < 4; ++c) chk += x[c] * 0.0;
    STE_UNROLL
    for (int e = 0; e < 10; ++e) chk += P[e] * 0.0;
    if (!(chk == 0.0)) st |= STE_STATUS_NAN;  // inf*0 and nan*0 are NaN
    p.status[t] = st;
}

template <bool kGains, bool kFastUpd, bool kRobust = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void ukf_forward_l1(const KParams p) {
    forward_tile_l1<kGains, kFastUpd, kRobust, false>(p, (size_t)blockIdx.x * 64 + threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// The forward passes of MANY windows as one launch of resident waves working through a host-made schedule
// (ste_ukf_forward_sched_f64; DESIGN.md section 5, "Scheduled forward pass").
//
// A forward wave is indivisible for a whole pass (3.6-4.5 ms at 500 steps) when every window is a launch of its own: W
// windows of T tiles on S SIMDs then cost ceil(W T / S) pass times although they hold only W T / S pass times of work --
// the driver's 20 steps (3 140 tiles on 1 024 SIMDs) pay 4 generations for 3.07, a 100 000-track fleet 2 for 1.53.  Here the
// unit of work is a (tile, time slice) ITEM -- 64 tracks x one STE_SLICE_ALIGN-aligned step range, the slices of
// ste_ukf_forward_f64, bit-identical because a slice starts from the history row the previous one left -- and the launch
// is one wave per SIMD, each walking its own column of a [rounds][waves] item table.  Slices of one tile may run on
// different waves: a finished slice publishes the tile's slice count (release, agent scope: its history rows, work rows and
// status are in memory before the count moves) and the next slice acquires it.  The table comes from the host, which checks
// that every tile's slices appear in order, at most one per round -- so a wait only ever looks at an EARLIER round, and with
// all waves resident (the launch is sized to the SIMDs its stream may use) it is satisfied without spinning in the common
// case.  Every wait is bounded (s_memrealtime): a schedule that cannot progress raises the launch's error word and every
// wave leaves, instead of hanging the device.
// A window whose tiles have all finished their last slice bumps window_done[w]; smoothers wait for that count
// (ste_stream_wait_counter: a one-wave gate kernel on the smoother's stream).
// ---------------------------------------------------------------------------------------------------------------
struct SchedItem {
    int kp;    // index into the KParams table (one entry per (window, run of slices)), < 0: nothing to do this round
    int tile;  // 64-track tile of that window | slices of the tile done once this item has run << 24
    int prog;  // index of the tile's progress counter
    int meta;  // slice | last-slice flag << 8 | window << 9 (22 bits) | hand-over flags: bit 31 = the slice before this one ran
               // on ANOTHER wave (wait for the tile's count and acquire), bit 30 = the next slice runs on another wave, or there is
               // none (release and publish).  A tile that stays on its wave -- the rule -- needs neither: a wave sees its own stores.
};
struct SchedParams {
    const KParams* kps;
    const SchedItem* items;  // [nrounds][nwaves]
    int nrounds, nwaves;
    int* progress;     // [tiles of all windows] slices finished, zeroed before the launch
    int* window_done;  // [nwindows] tiles finished, zeroed before the launch
    int* error;        // one word, zeroed before the launch: 1 = a wait timed out (the schedule could not progress)
    unsigned long long timeout_ticks;  // bound of a single wait, in s_memrealtime ticks (100 MHz)
    int* started;      // optional, zeroed before the launch: waves that have begun (== nwaves: the launch is resident)
};

// wave-uniform: spin (sleeping) until *flag >= need or the bound is reached
__device__ __forceinline__ bool sched_wait(const int* flag, int need, unsigned long long timeout_ticks) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) return false;
            __builtin_amdgcn_s_sleep(64);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

template <bool kGains, bool kFastUpd, bool kRobust>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void ukf_forward_sched(const SchedParams sp) {
    typedef const KParams __attribute__((address_space(4))) ConstKParams;  // the table is constant for the launch: scalar loads,
                                                                           // re-materialised where they are used like kernel arguments
    if (sp.started && threadIdx.x == 0) __hip_atomic_fetch_add(sp.started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int r = 0; r < sp.nrounds; ++r) {
        const SchedItem* ip = sp.items + ((size_t)r * sp.nwaves + blockIdx.x);
        const int kp = __builtin_amdgcn_readfirstlane(ip->kp);
        if (kp < 0) continue;
        const int tile_word = __builtin_amdgcn_readfirstlane(ip->tile), prog = __builtin_amdgcn_readfirstlane(ip->prog);
        const int tile = tile_word & 0xffffff, publish = (tile_word >> 24) & 0xff;  // publish: slices done once this item has run
        const int meta = __builtin_amdgcn_readfirstlane(ip->meta);
        const int slice = meta & 0xff;
        // (the error word is looked at only where a wave waits: a launch that cannot progress ends through its bounded waits)
        if (meta < 0 && (__hip_atomic_load(sp.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                         !sched_wait(sp.progress + prog, slice, sp.timeout_ticks))) {
            if (threadIdx.x == 0) __hip_atomic_store(sp.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const KParams& p = *(const KParams*)(ConstKParams*)(uintptr_t)(sp.kps + kp);
        if (threadIdx.x == 0) *(volatile int*)&g_sched_word[0] = p.k0;
        forward_tile_l1<kGains, kFastUpd, kRobust, true>(p, (size_t)tile * 64 + threadIdx.x);
        if (meta & 0x40000000) {
            // publish: this wave's stores have been acknowledged, then made visible at agent scope, before the count moves
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) {
                __hip_atomic_store(sp.progress + prog, publish, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (meta & 0x100)
                    __hip_atomic_fetch_add(sp.window_done + ((meta >> 9) & 0x1fffff), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Table upload of a scheduled launch as a kernel on the launch's own stream: copies `ncopy` 16-byte words from page-locked host
// memory (read over the host link) into the device workspace and clears the `nzero` words behind them (the progress counters).
// Nothing but kernels of this stream between two launches: no copy-engine transfer (an engine, and a queue, shared with
// whatever else the node is doing) in front of every scheduled launch.
__global__ __launch_bounds__(256) void sched_upload(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t ncopy, size_t nzero) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncopy + nzero; i += stride)
        dst[i] = i < ncopy ? src[i] : make_uint4(0, 0, 0, 0);
}

// One wave that leaves when *counter >= need (or after the bound, raising *error): what a smoother's stream runs in front of
// the smoother of a window whose forward pass is part of a scheduled launch on another stream.
__global__ __launch_bounds__(64) void sched_gate(const int* counter, int need, int* error, unsigned long long timeout_ticks) {
    if (!sched_wait(counter, need, timeout_ticks) && threadIdx.x == 0 && error)
        __hip_atomic_store(error, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ---------------------------------------------------------------------------------------------------------------
// forward pass, one DPP quad per track (ste_quad.h)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tri_index(int r, int c) { return r * 4 - (r * (r - 1)) / 2 + (c - r); }  // r <= c

// Predict for a quad (unscented.py:178-207) and, when `work` is given, the smoother's x_b, P_b, D of this step.
// With T = sqrtm(scale P) the cross-covariance is D = wi sum_i T_i (chi'_{i+} - chi'_{i-})^T (the centre's deviation is
// zero and x_b cancels in the difference), and P_b, which is centred on x_k instead of on the predicted mean, follows
// from the predicted covariance by P_b = P^- + e b^T + b e^T + b b^T with b = x^- - x_k and e = (weighted mean) - x^-
// (= minus the injected predict noise; zero in noise-free runs), because the weights sum to one.
template <class KT, class KG>
__device__ __forceinline__ int quad_predict(const Mats& p, const QuadCtx& cx, double (&x)[4], double (&Px)[4],
                                            QuadBasis& basis, double dt, double sr, double cr, const double* noise,
                                            const double* noise_rts, double* work, size_t nrow, size_t B, size_t t,
                                            bool full_row, bool noise_mode, bool& flagged, double* first_bad,
                                            const KT& tk, const KG& gk) {
    double Tn[4], s0[4], sp[4], sm[4], m[4], xp[4];
    int st = quad_sym_sqrt(Px, p.fan_scale, cx, basis, Tn);
    quad_propagate<KT, KG>(x, Tn, dt, sr, cr, s0, sp, sm, tk, gk);
    STE_UNROLL
    for (int c = 0; c < 4; ++c) m[c] = fma(p.w0, s0[c], p.wi * quad_sum(sp[c] + sm[c]));
    STE_UNROLL
    for (int c = 0; c < 4; ++c) xp[c] = m[c];
    if (noise) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) xp[c] += noise[(nrow * 4 + c) * B + t];
    }
    double Pn[4];
    quad_scatter(s0, sp, sm, xp, p.w0, p.wi, cx, Pn);
    if (work) {
        const int q = cx.q;
        double xb[4], D[2], bv[4], bx[4];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            xb[c] = m[c];
            bv[c] = xp[c] - x[c];
        }
        if (noise_rts) {
            STE_UNROLL
            for (int c = 0; c < 4; ++c) xb[c] += noise_rts[(nrow * 4 + c) * B + t];
        }
        STE_UNROLL
        for (int c = 0; c < 2; ++c) {  // columns 0-1 of D (row q): sum_l T[q][l] (chi'_{l+} - chi'_{l-})[c]
            const double dlt = sp[c] - sm[c];
            double acc = Tn[0] * bcast<0>(dlt);
            acc = fma(Tn[1], bcast<1>(dlt), acc);
            acc = fma(Tn[2], bcast<2>(dlt), acc);
            acc = fma(Tn[3], bcast<3>(dlt), acc);
            D[c] = acc;
        }
        xorperm(bv, q, bx);
        double Pb[4];
        STE_UNROLL
        for (int s = 0; s < 4; ++s) Pb[s] = fma(bx[0], bx[s], Pn[s]);
        if (noise) {
            double ev[4], ex[4];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) ev[c] = m[c] - xp[c];
            xorperm(ev, q, ex);
            STE_UNROLL
            for (int s = 0; s < 4; ++s) Pb[s] += fma(ex[0], bx[s], bx[0] * ex[s]);
        }
        double* w = work + (nrow * kWorkElems) * B + t;
        if (q < 2 || noise_mode) {  // rows 2-3: only with recorded noise, else the smoother takes them from P_b (see kWorkD)
            STE_UNROLL
            for (int c = 0; c < 2; ++c) w[(kWorkD + q * 2 + c) * B] = p.wi * D[c];  // row q of D, columns 0-1
        }
        if (full_row) {  // elsewhere the smoother rebuilds x_b and P_b from history rows k and k + 1 (see kWorkD)
            w[(kWorkXb + q) * B] = sel4(xb, q);
            STE_UNROLL
            for (int s = 0; s < 4; ++s) {
                const int c = q ^ s;
                if (c >= q) w[(kWorkPb + tri_index(q, c)) * B] = Pb[s];
            }
        }
        int stq = st & (STE_STATUS_CLAMPED | STE_STATUS_NOCONV);  // each lane saw its own eigenvalue: combine over the quad
        stq |= dpp_move_i<0xB1>(stq);
        stq |= dpp_move_i<0x4E>(stq);
        const bool bad_now = stq != 0;
        if (bad_now && !flagged && q == 0) *first_bad = (double)((long long)nrow + late_k0());  // absolute step index
        flagged = flagged || bad_now;
        if (flagged) {
            // columns 2-3 of D (row q): 2 wi (T T)[q][2:4] -- (2 wi scale) P_k[q][2:4] only for an exact square root
            STE_UNROLL
            for (int c = 2; c < 4; ++c) {
                double acc = Tn[0] * bcast<0>(Tn[c]);
                acc = fma(Tn[1], bcast<1>(Tn[c]), acc);
                acc = fma(Tn[2], bcast<2>(Tn[c]), acc);
                acc = fma(Tn[3], bcast<3>(Tn[c]), acc);
                w[(kWorkD23 + q * 2 + (c - 2)) * B] = (p.wi + p.wi) * acc;
            }
        }
    }
    STE_UNROLL
    for (int c = 0; c < 4; ++c) {
        x[c] = xp[c];
        Px[c] = Pn[c];
    }
    return st;
}

// Measurement update for a quad (unscented.py:219-265): every 4x4 product is one row per lane, rows of the other
// operand arrive by quad broadcasts.
template <bool kRobust>
__device__ __forceinline__ int quad_update(const Mats& p, const QuadCtx& cx, double (&x)[4], double (&Px)[4],
                                           const double (&zin)[4], const double* noise, size_t nrow, size_t B,
                                           size_t t) {
    const int q = cx.q;
    double z[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) z[c] = zin[c];
    if (noise) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) z[c] += noise[(nrow * 4 + c) * B + t];
    }
    // G = P H^T (row q): G[c] = sum_k P[q][q^k] H[c][q^k]
    double G[4], Sn[4], Sin[4], K[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) {
        double acc = Px[0] * cx.HTx[c][0];
        STE_UNROLL
        for (int k = 1; k < 4; ++k) acc = fma(Px[k], cx.HTx[c][k], acc);
        G[c] = acc;
    }
    quad_mm_rows(cx.Hrow, G, Sn);  // H G = H P H^T (row q)
    double Rrow[4];                // row q of the measurement covariance this update runs with
    STE_UNROLL
    for (int c = 0; c < 4; ++c) Rrow[c] = cx.Rrow[c];
    int rst = 0;
    if (kRobust) {
        // Opt-in robustification (check_robustness, unscented.py:353-387) on the un-noised observation with the
        // reference's y = z - x: while gamma = |y^T S^+ y| exceeds chi_alpha, lambda += (gamma - chi)/(y^T S^+ R S^+ y) and
        // R <- lambda R (compounding, as written there).  A quad's four lanes see the same gamma; tracks that are done
        // keep their values while others in the wave iterate.
        double y0[4];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) y0[c] = zin[c] - x[c];
        const double yq = sel4(y0, q);
        auto terms = [&](double& gamma, double& denom) -> int {
            double S[4], Si[4];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) S[c] = Sn[c] + Rrow[c];
            const int pst = quad_sym_pinv(S, cx, Si);
            double uq = 0.0;  // (S^+ y)[q]
            STE_UNROLL
            for (int c = 0; c < 4; ++c) uq = fma(Si[c], y0[c], uq);
            gamma = fabs(quad_sum(yq * uq));
            double vq = 0.0;  // (R S^+ y)[q]
            vq = fma(Rrow[0], bcast<0>(uq), vq);
            vq = fma(Rrow[1], bcast<1>(uq), vq);
            vq = fma(Rrow[2], bcast<2>(uq), vq);
            vq = fma(Rrow[3], bcast<3>(uq), vq);
            denom = quad_sum(uq * vq);
            return pst;
        };
        double gamma, denom, lambda = 1.0;
        rst |= terms(gamma, denom);
        for (int it = 0; it < p.robust_iters; ++it) {
            const bool active = gamma > p.chi_alpha;
            if (!__any(active)) break;
            lambda = active ? lambda + (gamma - p.chi_alpha) / denom : lambda;
            STE_UNROLL
            for (int c = 0; c < 4; ++c) Rrow[c] = active ? Rrow[c] * lambda : Rrow[c];
            double g2, d2;
            const int pst = terms(g2, d2);
            if (active) {
                rst |= pst;
                gamma = g2;
                denom = d2;
            }
        }
        if (gamma > p.chi_alpha) rst |= STE_STATUS_ROBUST_CAP;
    }
    STE_UNROLL
    for (int c = 0; c < 4; ++c) Sn[c] += Rrow[c];  // S = H P H^T + R
    const int st = quad_sym_pinv(Sn, cx, Sin) | rst;
    quad_mm_rows(G, Sin, K);  // K = G S^+  (row q)
    double y[4];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        double hx = p.H[r * 4 + 0] * x[0];
        STE_UNROLL
        for (int c = 1; c < 4; ++c) hx = fma(p.H[r * 4 + c], x[c], hx);
        y[r] = z[r] - hx;
    }
    y[3] = wrap180(y[3]);
    double xq = sel4(x, q);
    STE_UNROLL
    for (int c = 0; c < 4; ++c) xq = fma(K[c], y[c], xq);
    x[0] = bcast<0>(xq);
    x[1] = bcast<1>(xq);
    x[2] = bcast<2>(xq);
    x[3] = floored_mod(bcast<3>(xq), 360.0);
    // Joseph form: A = I - K H, P = A P A^T + K R K^T
    double A[4], KR[4], Pnat[4], AP[4], P1[4], P2[4], Pnew[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) {
        double kh = K[0] * p.H[0 * 4 + c], kr = K[0] * p.R[0 * 4 + c];
        STE_UNROLL
        for (int l = 1; l < 4; ++l) {
            kh = fma(K[l], p.H[l * 4 + c], kh);
            kr = fma(K[l], p.R[l * 4 + c], kr);
        }
        A[c] = ((c == q) ? 1.0 : 0.0) - kh;
        KR[c] = kr;
    }
    if (kRobust) quad_mm_rows(K, Rrow, KR);  // K R with this update's rescaled R (rows live one per lane)
    xorperm(Px, q, Pnat);
    quad_mm_rows(A, Pnat, AP);
    quad_mm_rows_t(AP, A, P1);
    quad_mm_rows_t(KR, K, P2);
    STE_UNROLL
    for (int c = 0; c < 4; ++c) Pnew[c] = P1[c] + P2[c];
    xorperm(Pnew, q, Px);
    return st;
}

// The same update for H = diag(1, 1, 0, 0) and an R confined to that block (every example and the CLI of the reference), in
// closed form like lane_update_sel2: S = H P H^T + R is 2 x 2, K = P[:, 0:2] S^+ has two columns, and the Joseph form needs the
// products with those two columns only.  Row q of every matrix in lane q; rows 0 and 1 of P and the two columns of K reach
// the other lanes by quad broadcasts.  ~150 instead of ~600 instructions per update; with kRobust the rescaling loop in closed
// form too (robust_rescale_sel2 on the three entries of the block, the same on every lane of the quad).
template <bool kRobust>
__device__ __forceinline__ int quad_update_sel2(const Mats& p, const QuadCtx& cx, double (&x)[4], double (&Px)[4],
                                                const double (&zin)[4], const double* noise, size_t nrow, size_t B,
                                                size_t t) {
#pragma clang fp contract(off)  // explicit fma() only
    const int q = cx.q;
    double z[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) z[c] = zin[c];
    double Pn[4];  // row q of P, natural order
    xorperm(Px, q, Pn);
    const double p00 = bcast<0>(Pn[0]), p11 = bcast<1>(Pn[1]);
    const double p01 = 0.5 * (bcast<0>(Pn[1]) + bcast<1>(Pn[0]));  // the quad keeps both triangles: symmetrised like quad_sym_pinv
    double r00 = p.R[0], r01 = p.R[1], r11 = p.R[5];
    int st = 0;
    if (kRobust) {
        double Pblk[10];
        STE_UNROLL
        for (int e = 0; e < 10; ++e) Pblk[e] = 0.0;
        Pblk[tix(0, 0)] = p00;
        Pblk[tix(0, 1)] = p01;
        Pblk[tix(1, 1)] = p11;
        st = robust_rescale_sel2(p, x, Pblk, z, r00, r01, r11);  // on the un-noised observation (unscented.py:228)
    }
    if (noise) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) z[c] += noise[(nrow * 4 + c) * B + t];
    }
    double Sm[4][4], Si[4][4];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) Sm[r][c] = 0.0;
    }
    Sm[0][0] = p00 + r00;
    Sm[0][1] = p01 + r01;
    Sm[1][0] = Sm[0][1];
    Sm[1][1] = p11 + r11;
    sym_pinv4_block2(Sm, Si);
    const double i00 = Si[0][0], i01 = Si[0][1], i11 = Si[1][1];
    const double k0 = fma(Pn[1], i01, Pn[0] * i00), k1 = fma(Pn[1], i11, Pn[0] * i01);  // K[q][0], K[q][1]
    const double y0 = z[0] - x[0], y1 = z[1] - x[1];
    const double poison = fma(0.0, z[2], 0.0 * z[3]);  // K[:, 2:4] y[2:4] with K[:, 2:4] = 0: NaN iff z[2] or z[3] is not finite
    const double xq = fma(k1, y1, fma(k0, y0, sel4(x, q))) + poison;
    x[0] = bcast<0>(xq);
    x[1] = bcast<1>(xq);
    x[2] = bcast<2>(xq);
    x[3] = floored_mod(bcast<3>(xq), 360.0);
    // AP = (I - K H) P, row q:  P[q][c] - K[q][0] P[0][c] - K[q][1] P[1][c]
    double AP[4], K0c[4], K1c[4], Pnew[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) {
        AP[c] = fma(-k1, bcast<1>(Pn[c]), fma(-k0, bcast<0>(Pn[c]), Pn[c]));
    }
    K0c[0] = bcast<0>(k0);
    K0c[1] = bcast<1>(k0);
    K0c[2] = bcast<2>(k0);
    K0c[3] = bcast<3>(k0);
    K1c[0] = bcast<0>(k1);
    K1c[1] = bcast<1>(k1);
    K1c[2] = bcast<2>(k1);
    K1c[3] = bcast<3>(k1);
    const double kr0 = fma(k1, r01, k0 * r00), kr1 = fma(k1, r11, k0 * r01);  // (K R)[q][0:2]
    STE_UNROLL
    for (int c = 0; c < 4; ++c) {
        const double p1 = fma(-AP[1], K1c[c], fma(-AP[0], K0c[c], AP[c]));  // (A P A^T)[q][c]
        const double p2 = fma(kr1, K1c[c], kr0 * K0c[c]);                    // (K R K^T)[q][c]
        Pnew[c] = p1 + p2;
    }
    xorperm(Pnew, q, Px);
    return st;
}

// kSel: H = diag(1, 1, 0, 0) with R in the same block (chosen by launch_forward from the matrices): closed-form update.
template <bool kGains, bool kRobust, bool kSel>
// __launch_bounds__(64, 2): at most 256 VGPRs, so that two waves fit on a SIMD.  What no longer fits is needed only by the
// branching fallback of the propagation (its library-call constants go to scratch); the step loop itself has no scratch
// access.  Alone the kernel is 5 % faster than the 362-VGPR build (no AGPR traffic), and two forward passes on the same
// compute units take 3.5 ms together instead of 2 x 2.36: a lone wave leaves half of the fp64 pipe's issue slots unused.
__global__ __launch_bounds__(64, 2) void ukf_forward_q4(const KParams p) {
    const size_t B = (size_t)p.ld;  // row pitch of every per-track array
    // p.qpw quads per wave (16 fill it).  A wave pays for the slowest of its tracks at every step -- another Jacobi sweep, another
    // robust rescaling, the device-library fallback of the fan whenever ANY of its lanes asks -- so when there are fewer tracks than
    // the chip has SIMDs to spare, each gets a wave of its own (launch_forward): BASELINE configs[3]'s seven ships take 41 .. 89 ms
    // each on their own and 128 ms sharing one wave.
    const int quad = (int)(threadIdx.x >> 2);
    if (quad >= p.qpw) return;
    const size_t t = (size_t)blockIdx.x * (size_t)p.qpw + (size_t)quad;
    const int q = (int)(threadIdx.x & 3);
    if (t >= (size_t)p.B) return;  // whole quads leave together
    const bool cont = (p.flags & kFlagContinue) != 0;  // a later time slice (see ukf_forward_l1; full covariances only)
    const int ns = (p.nsteps ? p.nsteps[t] - p.k0 : p.Nmax);
    if (cont && ns <= 0) return;
    QuadCtx cx;
    quad_ctx_init(p.m, q, cx);
    // the sin / cos and arctangent polynomial coefficients in VGPRs for the whole kernel: a wave that has its SIMD to itself
    // pays an issue slot for every s_mov_b32 that brings half a literal into a register (92 of them in the step's main block,
    // 46 with these 23 coefficients resident).  The closed-form update (kSel) leaves the 46 registers free; the general one
    // (246-256 registers already) keeps the literals.
    typename std::conditional<kSel, TrigReg, TrigLit>::type tk;
    typename std::conditional<kSel, GeoReg, GeoLit>::type gk;
    trig_reg_init(tk);
    geo_reg_init(gk);

    double x[4], Px[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) x[c] = p.x0[c * B + t];
    STE_UNROLL
    for (int k = 0; k < 4; ++k) {
        const int e = q * 4 + (q ^ k);
        Px[k] = (p.flags & STE_FLAG_SHARED_P0) ? p.P0[e] : p.P0[(size_t)e * B + t];
    }
    auto store_row = [&](size_t row) {
        p.fwd_mean[(row * 4 + q) * B + t] = sel4(x, q);
        STE_UNROLL
        for (int k = 0; k < 4; ++k) {
            const int c = q ^ k;
            if (!(p.flags & STE_FLAG_PACKED_COV))
                p.fwd_cov[(row * 16 + q * 4 + c) * B + t] = Px[k];
            else if (c >= q)
                p.fwd_cov[(row * 10 + tri_index(q, c)) * B + t] = Px[k];
        }
    };
    store_row(0);  // slot 0 = prior (kalman_filter.py:76-77)

    int st = cont ? (p.status[t] & ~STE_STATUS_NAN) : 0;
    bool flagged = false;
    double* first_bad = (kGains && p.rts_work) ? p.first_bad + t : nullptr;
    if (first_bad) {
        if (cont)
            flagged = *first_bad != kNeverBad;
        else if (q == 0)
            *first_bad = kNeverBad;
    }
    const bool noise_mode = p.noise_pred || p.noise_upd || p.noise_rts;
    const bool initial_update = !(p.flags & STE_FLAG_NO_INITIAL_UPDATE);
    if (kGains && initial_update && ns > 0) {
        // smoother step 0 reads history row 0 = the prior, not the state the first predict starts from
        double xc[4], Pc[4];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            xc[c] = x[c];
            Pc[c] = Px[c];
        }
        QuadBasis cold;
        cold.valid = false;
        st |= quad_predict(p.m, cx, xc, Pc, cold, p.dt[t], p.sog_rate[t], p.cog_rate[t], nullptr, p.noise_rts,
                           p.rts_work, 0, B, t, true, noise_mode, flagged, first_bad, tk, gk);
    }
    if (initial_update) {
        double z0[4];
        load_vec(p.z, 0, B, t, z0);
        st |= kSel ? quad_update_sel2<kRobust>(p.m, cx, x, Px, z0, p.noise_upd, 0, B, t)
                   : quad_update<kRobust>(p.m, cx, x, Px, z0, p.noise_upd, 0, B, t);  // kalman_filter.py:81
    }
    QuadBasis basis;
    basis.valid = false;
    double dt_n = 0.0, sr_n = 0.0, cr_n = 0.0;
    int ui_n = -1;
    if (ns > 0) {
        dt_n = p.dt[t];
        sr_n = p.sog_rate[t];
        cr_n = p.cog_rate[t];
        ui_n = p.upd_idx[t];
    }
    for (int k = 0; k < p.Nmax; ++k) {
        const bool live = k < ns;
        if (!__any(live)) break;
        if (live) {
            const double dt = dt_n, sr = sr_n, cr = cr_n;
            const int ui = ui_n;
            // Unconditional loads with clamped indices: a load inside an `if` makes hipcc drain the queue with
            // s_waitcnt vmcnt(0) where the branch rejoins, i.e. right after issuing it -- a full HBM round trip per step.
            double zk[4];
            const bool ui_ok = ui < p.Tmax;  // an observation column past the padded batch: flagged, update skipped
            load_vec(p.z, (size_t)((ui >= 0 && ui_ok) ? ui : 0), B, t, zk);
            {
                const size_t o = (size_t)(k + 1 < ns ? k + 1 : k) * B + t;
                dt_n = p.dt[o];
                sr_n = p.sog_rate[o];
                cr_n = p.cog_rate[o];
                ui_n = p.upd_idx[o];
            }
            if ((k & (kColdEvery - 1)) == 0) basis.valid = false;
            double* work = (kGains && !(k == 0 && initial_update)) ? p.rts_work : nullptr;
            st |= quad_predict(p.m, cx, x, Px, basis, dt, sr, cr, p.noise_pred, p.noise_rts, work, (size_t)k, B, t,
                               (ui >= 0 && ui_ok) || noise_mode, noise_mode, flagged, first_bad, tk, gk);
            if (ui >= 0 && ui_ok)
                st |= kSel ? quad_update_sel2<kRobust>(p.m, cx, x, Px, zk, p.noise_upd, (size_t)k + 1, B, t)
                           : quad_update<kRobust>(p.m, cx, x, Px, zk, p.noise_upd, (size_t)k + 1, B, t);
            if (!ui_ok) st |= STE_STATUS_BAD_INDEX;
            store_row((size_t)k + 1);
        }
    }
    double chk = 0.0;
    STE_UNROLL
    for (int c = 0; c < 4; ++c) chk += x[c] * 0.0 + Px[c] * 0.0;
    if (!(chk == 0.0)) st |= STE_STATUS_NAN;
    st |= dpp_move_i<0xB1>(st);
    st |= dpp_move_i<0x4E>(st);
    if (q == 0) p.status[t] = st;
}

// ---------------------------------------------------------------------------------------------------------------
// URTSS backward pass, one lane per track (unscented.py:285-351)
// ---------------------------------------------------------------------------------------------------------------
// The literal smoother: recomputes the fan, its nine great-circle steps and both pseudo-inverses per step.  Used when
// there are no work rows from the forward pass (rts_work == NULL, a history that ste_ukf_forward_f64 did not write).
__device__ __forceinline__ void store_pos(const KParams& p, size_t row, size_t B, size_t t, const double (&xs)[4]) {
    if (p.sm_pos) {  // the tensor configs[2] exchanges between GPUs: smoothed lon / lat on their own (ste.h: sm_pos)
        st_stream(&p.sm_pos[(row * 2 + 0) * B + t], xs[0]);
        st_stream(&p.sm_pos[(row * 2 + 1) * B + t], xs[1]);
    }
}

__global__ __launch_bounds__(64) void urtss_backward_l1(const KParams p) {
    const size_t B = (size_t)p.ld;
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= (size_t)p.B) return;
    const int ns = p.nsteps ? p.nsteps[t] : p.Nmax;
    const double* srp = p.sog_rate_rts ? p.sog_rate_rts : p.sog_rate;
    const double* crp = p.cog_rate_rts ? p.cog_rate_rts : p.cog_rate;

    // row ns: smoothed = filtered
    double xs[4], Ps[4][4];
    load_vec(p.fwd_mean, (size_t)ns, B, t, xs);
    const bool packed = (p.flags & STE_FLAG_PACKED_COV) != 0;
    load_cov_m(p.fwd_cov, packed, (size_t)ns, B, t, Ps);
    store_vec(p.sm_mean, (size_t)ns, B, t, xs);
    store_cov_m(p.sm_cov, packed, (size_t)ns, B, t, Ps);
    store_pos(p, (size_t)ns, B, t, xs);

    // filtered row of the first step to process, prefetched
    double xn[4] = {0, 0, 0, 0}, Pn[4][4] = {};
    double dt_n = 0.0, sr_n = 0.0, cr_n = 0.0;
    if (ns > 0) {
        load_vec(p.fwd_mean, (size_t)ns - 1, B, t, xn);
        load_cov_m(p.fwd_cov, packed, (size_t)ns - 1, B, t, Pn);
        const size_t o = (size_t)(ns - 1) * B + t;
        dt_n = p.dt[o];
        sr_n = srp[o];
        cr_n = crp[o];
    }
    int st = 0;
    EigBasis fan_basis, pb_basis;
    fan_basis.valid = false;
    pb_basis.valid = false;
    for (int k = p.Nmax - 1; k >= 0; --k) {
        if (!__any(k < ns)) continue;  // ragged batch: nobody in this wave has reached its last step yet
        if (k < ns) {
            double xk[4], Pk[4][4];
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                xk[r] = xn[r];
                STE_UNROLL
                for (int c = 0; c < 4; ++c) Pk[r][c] = Pn[r][c];
            }
            const double dt = dt_n, sr = sr_n, cr = cr_n;
            if (k > 0) {
                load_vec(p.fwd_mean, (size_t)k - 1, B, t, xn);
                load_cov_m(p.fwd_cov, packed, (size_t)k - 1, B, t, Pn);
                const size_t o = (size_t)(k - 1) * B + t;
                dt_n = p.dt[o];
                sr_n = srp[o];
                cr_n = crp[o];
            }
            double sig0[9][4], sig[9][4];
            if ((k & (kColdEvery - 1)) == kColdEvery - 1) {
                fan_basis.valid = false;
                pb_basis.valid = false;
            }
            st |= propagate_fan<true>(xk, Pk, p.m.fan_scale, dt, sr, cr, sig0, sig, fan_basis);
            double xb[4];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) {
                double acc = 0.0;
                STE_UNROLL
                for (int j = 1; j < 9; ++j) acc += sig[j][c];
                xb[c] = fma(p.m.w0, sig[0][c], p.m.wi * acc);
            }
            if (p.noise_rts) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) xb[c] += p.noise_rts[((size_t)k * 4 + c) * B + t];
            }
            // P_b is centred on the filtered mean x_k, not on x_b (unscented.py:324-325)
            double dk[9][4], db[9][4];
            STE_UNROLL
            for (int j = 0; j < 9; ++j) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) {
                    dk[j][c] = sig[j][c] - xk[c];
                    db[j][c] = sig[j][c] - xb[c];
                    sig0[j][c] -= xk[c];
                }
            }
            double Pb[4][4], D[4][4], Pbi[4][4], K[4][4];
            weighted_outer<true>(dk, dk, p.m.w0, p.m.wi, Pb);
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) Pb[r][c] += p.m.Q[r * 4 + c];
            }
            weighted_outer<false>(sig0, db, p.m.w0, p.m.wi, D);  // unscented.py:328-330
            st |= sym_pinv4<true>(Pb, Pbi, pb_basis);
            pb_basis.valid = true;
            mm(D, Pbi, K);  // unscented.py:333
            double y[4];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) y[c] = xs[c] - xb[c];
            y[3] = wrap180(y[3]);
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                double acc = xk[r];
                STE_UNROLL
                for (int c = 0; c < 4; ++c) acc = fma(K[r][c], y[c], acc);
                xs[r] = acc;
            }
            xs[3] = floored_mod(xs[3], 360.0);
            double dP[4][4], KdP[4][4], U[4][4];
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) dP[r][c] = Ps[r][c] - Pb[r][c];
            }
            mm(K, dP, KdP);
            mmt_sym(KdP, K, U);
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) Ps[r][c] = Pk[r][c] + U[r][c];
            }
            store_vec(p.sm_mean, (size_t)k, B, t, xs);
            store_cov_m(p.sm_cov, packed, (size_t)k, B, t, Ps);
            store_pos(p, (size_t)k, B, t, xs);
        }
    }
    if (!all_finite(xs, Ps)) st |= STE_STATUS_NAN;
    if (st) atomicOr(&p.status[t], st);
}

// ---------------------------------------------------------------------------------------------------------------
// URTSS backward pass from the rows the forward pass left in rts_work: per step the gain K = D pinv(P_b)
// (unscented.py:333) and the recurrence of :337-349,
//     y = x^s_{k+1} - x_b (heading wrapped),  x^s_k = x_k + K y,  P^s_k = P_k + K (P^s_{k+1} - P_b) K^T,
// one lane per track.  Columns 0-1 of D come from the work row, columns 2-3 from the filtered covariance of row k (or the
// work row, at and after the track's first bad square root); x_b and P_b from the work row where they were stored (see
// kWorkD) and otherwise from the filtered rows k and k + 1 the recurrence reads anyway.  Per track-step it reads 8 + 14
// doubles (D; mean and upper triangle of the filtered covariance), plus 14 on the steps that were followed by an
// update, and writes the 20 of the smoothed row.  No LDS, no barriers: the kernel is a chain of ~300 fp64 instructions
// per step whose loads are one row ahead; it is meant to run beside the forward passes of the next batches, whose waves
// take the issue slots it leaves (batch.SmootherPipeline).  Only reads the work rows: repeatable.
// ---------------------------------------------------------------------------------------------------------------
// Is work row k "full" (x_b and P_b stored)?  Mirrors the forward kernels' choice: full when the step was followed by a
// measurement update (ui = upd_idx[k] names a valid observation), for row 0 of a run that starts with an update, and
// throughout a run with recorded noise.
__device__ __forceinline__ bool work_row_full(const KParams& p, int k, int ui, bool always_full) {
    return always_full || (ui >= 0 && ui < p.Tmax) || (k == 0 && !(p.flags & STE_FLAG_NO_INITIAL_UPDATE));
}

struct RecurRow {
    double D2[8], xk[4], Pk[10];  // columns 0-1 of D of step k (rows 2-3 only with recorded noise); filtered row k
    double shift[2];              // smoother rates that differ from the forward rates: what they add to x_b[2:4]
};
// Speed and heading pass through the process model as x + rate * dt (non_linear_process.py:74-75) and the position
// components do not see the rates at all, so a smoother step that uses other rates than the forward step did
// (unscented.py:287-292: the smoother indexes the repeated rate arrays by step) sees the same propagated fan moved by
// (0, 0, d_sog * dt, d_cog * dt): x_b moves by that much, the fan's spread and D not at all, and P_b -- taken about the
// filtered mean x_k (:324-325) -- becomes C + b' b'^T with the new b' = x_b' - x_k.
template <bool kShift>
__device__ __forceinline__ void load_recur_row(const KParams& p, size_t k, size_t B, size_t t, bool d_rows23, RecurRow& g) {
    g.shift[0] = g.shift[1] = 0.0;
    if (kShift) {
        const double dt = p.dt[k * B + t];
        if (p.sog_rate_rts) g.shift[0] = (p.sog_rate_rts[k * B + t] - p.sog_rate[k * B + t]) * dt;
        if (p.cog_rate_rts) g.shift[1] = (p.cog_rate_rts[k * B + t] - p.cog_rate[k * B + t]) * dt;
    }
    const double* w = p.rts_work + (k * kWorkElems) * B + t;
    STE_UNROLL
    for (int e = 0; e < 4; ++e) g.D2[e] = w[(kWorkD + e) * B];
    if (d_rows23) {
        STE_UNROLL
        for (int e = 4; e < 8; ++e) g.D2[e] = w[(kWorkD + e) * B];
    }
    load_vec(p.fwd_mean, k, B, t, g.xk);
    load_cov_p(p.fwd_cov, (p.flags & STE_FLAG_PACKED_COV) != 0, k, B, t, g.Pk);
}

// What the smoother's step k needs besides the recurrence itself: x_b, P_b (stored, or rebuilt from history rows k and
// k + 1) and the gain K = D pinv(P_b) (unscented.py:315-333).  `cur` = work-row / history data of step k, (xn, Pn) = filtered
// row k + 1, `full` = work row k holds x_b and P_b.  Shared by the one-kernel smoother (urtss_recur_l1) and the two-kernel
// form for small batches (urtss_gains_all + urtss_recur_lean): same code, same bits.
template <bool kShift>
__device__ __forceinline__ int smoother_step_gain(const KParams& p, int k, size_t B, size_t t, const RecurRow& cur,
                                                  const double (&xn)[4], const double (&Pn)[10], bool full, bool always_full,
                                                  bool all_eig, double kappa, double first_bad, double (&xb)[4],
                                                  double (&Pb)[10], double (&K)[4][4]) {
    // x_b, P_b: stored (a quarter of the steps of the bench batch: loaded here, not a row ahead, to keep the
    // registers of a whole row free), or the prediction itself -- the step was not followed by an update, so row
    // k + 1 of the filtered history is x^-, P^-, and P_b = P^- + b b^T with b = x^- - x_k
    STE_UNROLL
    for (int c = 0; c < 4; ++c) xb[c] = xn[c];
    if (kShift) {  // the smoother's own rates (load_recur_row)
        xb[2] += cur.shift[0];
        xb[3] += cur.shift[1];
    }
    {
        double bv[4];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) bv[c] = xb[c] - cur.xk[c];
        STE_UNROLL
        for (int r = 0; r < 4; ++r) {
            STE_UNROLL
            for (int c = r; c < 4; ++c) Pb[tix(r, c)] = fma(bv[r], bv[c], Pn[tix(r, c)]);
        }
    }
    if (full) {
        const double* w = p.rts_work + ((size_t)k * kWorkElems) * B + t;
        STE_UNROLL
        for (int c = 0; c < 4; ++c) xb[c] = w[(kWorkXb + c) * B];
        STE_UNROLL
        for (int e = 0; e < 10; ++e) Pb[e] = w[(kWorkPb + e) * B];
        if (kShift) {
            // stored with the forward rates: P_b = C + b b^T with b = (weighted mean of the fan) - x_k, i.e. x_b - x_k
            // less the recorded noise that was added to x_b (unscented.py:319-325); now b' = b + s, s = (0, 0, shift):
            // P_b' = P_b + b s^T + s b^T + s s^T
            double bv[4];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) bv[c] = xb[c] - cur.xk[c];
            if (p.noise_rts) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) bv[c] -= p.noise_rts[((size_t)k * 4 + c) * B + t];
            }
            const double s2 = cur.shift[0], s3 = cur.shift[1];
            Pb[tix(0, 2)] = fma(bv[0], s2, Pb[tix(0, 2)]);
            Pb[tix(1, 2)] = fma(bv[1], s2, Pb[tix(1, 2)]);
            Pb[tix(0, 3)] = fma(bv[0], s3, Pb[tix(0, 3)]);
            Pb[tix(1, 3)] = fma(bv[1], s3, Pb[tix(1, 3)]);
            Pb[tix(2, 2)] = fma(bv[2] + bv[2] + s2, s2, Pb[tix(2, 2)]);
            Pb[tix(3, 3)] = fma(bv[3] + bv[3] + s3, s3, Pb[tix(3, 3)]);
            Pb[tix(2, 3)] = fma(bv[2], s3, fma(s2, bv[3] + s3, Pb[tix(2, 3)]));
            xb[2] += s2;
            xb[3] += s3;
        }
    }
    // the gain K = D pinv(P_b) (unscented.py:333)
    double D[4][4];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        D[r][0] = cur.D2[r * 2 + 0];
        D[r][1] = cur.D2[r * 2 + 1];
        D[r][2] = kappa * cur.Pk[tix(r, 2)];
        D[r][3] = kappa * cur.Pk[tix(r, 3)];
    }
    if (!always_full) {
        // rows 2-3 of D's first two columns are the cross moments of (speed, heading) with (lon, lat) about the
        // predicted mean, i.e. the corresponding entries of P^- before Q was added.  Where the step was not followed
        // by an update P^- is row k + 1 of the filtered history itself; on a full row (x_b, P_b stored) it is
        // P_b - b b^T with b = x_b - x_k (no recorded noise here): D[2:4, 0:2] = (P^- - Q)[0:2, 2:4]^T.  (Forming
        // P_b = P^- + b b^T first and subtracting b b^T again on every row cost ~eps |b_c b_r| on small cross moments.)
        double bv[4];
        STE_UNROLL
        for (int c = 0; c < 4; ++c) bv[c] = xb[c] - cur.xk[c];
        STE_UNROLL
        for (int r = 2; r < 4; ++r) {
            STE_UNROLL
            for (int c = 0; c < 2; ++c) {
                const double pm = full ? fma(-bv[c], bv[r], Pb[tix(c, r)]) : Pn[tix(c, r)];
                D[r][c] = pm - p.m.Q[c * 4 + r];
            }
        }
    }
    if (__builtin_expect(__any((double)k >= first_bad), 0)) {
        if ((double)k >= first_bad) {
            const double* w = p.rts_work + ((size_t)k * kWorkElems) * B + t;
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                D[r][2] = w[(kWorkD23 + r * 2 + 0) * B];
                D[r][3] = w[(kWorkD23 + r * 2 + 1) * B];
            }
        }
    }
    return smoother_gain(Pb, D, all_eig, K);

}

// kShift: the smoother has rates of its own (sog_rate_rts / cog_rate_rts); compiled out for batches that share them.
template <bool kShift>
__device__ __forceinline__ void smooth_tile_l1(const KParams& p, const size_t t) {
    const size_t B = (size_t)p.ld;
    if (t >= (size_t)p.B) return;
    const int ns = p.nsteps ? p.nsteps[t] : p.Nmax;
    const bool always_full = p.noise_pred || p.noise_upd || p.noise_rts;
    const bool all_eig = (p.tuning & 0x100) != 0;
    const double kappa = (p.m.wi + p.m.wi) * p.m.fan_scale;  // D[:, 2:4] = kappa P_k[:, 2:4] for an exact square root
    // first step at and after which this track's columns 2-3 of D are stored (kNeverBad: never)
    const double first_bad = p.first_bad[t];
    const int last_row = p.Nmax > 0 ? p.Nmax - 1 : 0;
    auto clampk = [&](int k) -> int { return min(max(k, 0), last_row); };

    // row ns: smoothed = filtered; it is also "filtered row k + 1" of the first step
    double xs[4], Ps[10], xn[4], Pn[10];
    const bool packed = (p.flags & STE_FLAG_PACKED_COV) != 0;
    load_vec(p.fwd_mean, (size_t)ns, B, t, xs);
    load_cov_p(p.fwd_cov, packed, (size_t)ns, B, t, Ps);
    store_vec(p.sm_mean, (size_t)ns, B, t, xs);
    store_cov_p(p.sm_cov, packed, (size_t)ns, B, t, Ps);
    store_pos(p, (size_t)ns, B, t, xs);
    STE_UNROLL
    for (int c = 0; c < 4; ++c) xn[c] = xs[c];
    STE_UNROLL
    for (int e = 0; e < 10; ++e) Pn[e] = Ps[e];

    // the row of the first step this lane processes, and the update index of the one after it, in flight
    RecurRow nxt;
    int ui_n = -1;
    if (ns > 0) {
        load_recur_row<kShift>(p, (size_t)(ns - 1), B, t, always_full, nxt);
        ui_n = p.upd_idx[(size_t)(ns - 1) * B + t];
    }
    int st = 0;
    for (int k = p.Nmax - 1; k >= 0; --k) {
        if (!__any(k < ns)) continue;  // ragged batch: nobody in this wave has reached its last step yet
        if (k < ns) {
            const RecurRow cur = nxt;
            const bool full = work_row_full(p, k, ui_n, always_full);
            double xb[4], Pb[10], K[4][4];
            st |= smoother_step_gain<kShift>(p, k, B, t, cur, xn, Pn, full, always_full, all_eig, kappa, first_bad, xb, Pb, K);
            // the row of the next step: in flight during the recurrence arithmetic below (and across the loop edge)
            {
                const int kn = clampk(k - 1);
                load_recur_row<kShift>(p, (size_t)kn, B, t, always_full, nxt);
                ui_n = p.upd_idx[(size_t)kn * B + t];
            }
            double y[4];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) y[c] = xs[c] - xb[c];
            y[3] = wrap180(y[3]);
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                double acc = cur.xk[r];
                STE_UNROLL
                for (int c = 0; c < 4; ++c) acc = fma(K[r][c], y[c], acc);
                xs[r] = acc;
            }
            xs[3] = floored_mod(xs[3], 360.0);
            // P^s_k = P_k + K (P^s_{k+1} - P_b) K^T, symmetric operands packed
            double dP[10];
            STE_UNROLL
            for (int e = 0; e < 10; ++e) dP[e] = Ps[e] - Pb[e];
            double KdP[4][4];
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) {
                    double acc = K[r][0] * dP[tix(0, c)];
                    STE_UNROLL
                    for (int i = 1; i < 4; ++i) acc = fma(K[r][i], dP[tix(i, c)], acc);
                    KdP[r][c] = acc;
                }
            }
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                STE_UNROLL
                for (int c = r; c < 4; ++c) {
                    double acc = KdP[r][0] * K[c][0];
                    STE_UNROLL
                    for (int i = 1; i < 4; ++i) acc = fma(KdP[r][i], K[c][i], acc);
                    Ps[tix(r, c)] = cur.Pk[tix(r, c)] + acc;
                }
            }
            store_vec(p.sm_mean, (size_t)k, B, t, xs);
            store_cov_p(p.sm_cov, packed, (size_t)k, B, t, Ps);
            store_pos(p, (size_t)k, B, t, xs);
            // row k becomes "row k + 1" of the next step
            STE_UNROLL
            for (int c = 0; c < 4; ++c) xn[c] = cur.xk[c];
            STE_UNROLL
            for (int e = 0; e < 10; ++e) Pn[e] = cur.Pk[e];
        }
    }
    double chk = 0.0;
    STE_UNROLL
    for (int c = 0; c < 4; ++c) chk += xs[c] * 0.0;
    STE_UNROLL
    for (int e = 0; e < 10; ++e) chk += Ps[e] * 0.0;
    if (!(chk == 0.0)) st |= STE_STATUS_NAN;
    if (st) atomicOr(&p.status[t], st);
}

template <bool kShift>
__global__ __launch_bounds__(64) void urtss_recur_l1(const KParams p) {
    smooth_tile_l1<kShift>(p, (size_t)blockIdx.x * 64 + threadIdx.x);
}

// The smoothers of every window of a scheduled forward launch as ONE launch (ste_urtss_backward_sched_f64): a wave per
// (window, 64-track tile), in the order the schedule finishes the tiles, each waiting (bounded) for ITS tile's last slice
// -- a tile is smoothed as soon as it has been filtered, not when the last tile of its window has.  The body is
// urtss_recur_l1's.  Waves that wait hold a wave slot each: the launch goes behind a gate on the forward launch's started-wave
// count, so that every forward wave has its SIMD before a waiting smoother wave could be in its way (include/ste.h).
struct SmoothItem {
    int kp;    // index of the window's parameter block
    int tile;  // 64-track tile of that window
    int prog;  // index of the tile's progress counter (the forward launch's)
    int need;  // slices of a forward pass of that window: the count the tile's counter reaches
};
struct SmoothSchedParams {
    const KParams* kps;
    const SmoothItem* items;
    const int* progress;
    int* error;  // 2 = a smoother wave waited longer than its bound
    unsigned long long timeout_ticks;
};
template <bool kShift>
__global__ __launch_bounds__(64) void urtss_recur_sched(const SmoothSchedParams sp) {
    typedef const KParams __attribute__((address_space(4))) ConstKParams;
    const SmoothItem* ip = sp.items + blockIdx.x;
    const int kp = __builtin_amdgcn_readfirstlane(ip->kp), tile = __builtin_amdgcn_readfirstlane(ip->tile);
    const int prog = __builtin_amdgcn_readfirstlane(ip->prog), need = __builtin_amdgcn_readfirstlane(ip->need);
    if (__hip_atomic_load(sp.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
        !sched_wait(sp.progress + prog, need, sp.timeout_ticks)) {
        if (threadIdx.x == 0) __hip_atomic_store(sp.error, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const KParams& p = *(const KParams*)(ConstKParams*)(uintptr_t)(sp.kps + kp);
    smooth_tile_l1<kShift>(p, (size_t)tile * 64 + threadIdx.x);
}



// ---------------------------------------------------------------------------------------------------------------
// The same smoother in two kernels, for SMALL batches (a few waves: the real-data runs -- BASELINE configs[0] and [3], the
// reference's 116-ship file).  There a wave of urtss_recur_l1 has a SIMD, indeed most of the chip, to itself, and its step
// is a chain: the loads of x_b / P_b on full rows issued where they are needed, the gain solve (with the eigenvalue route out
// of line for lanes whose P_b is close to singular), then the recurrence -- 4.7 us per step on configs[3].  But only the
// recurrence  x^s_k = x_k + K_k (x^s_{k+1} - x_b),  P^s_k = P_k + K_k (P^s_{k+1} - P_b) K_k^T  is sequential; x_b, P_b and the
// gain K_k = D_k pinv(P_b) depend on the forward pass alone (unscented.py:297-333).  So:
//   urtss_gains_all   one lane per (step, track): x_b, P_b, K for every step of every track at once, written back into the
//                     work row of that step (K over the D columns it was formed from, x_b and P_b into their slots on every row);
//   urtss_recur_lean  one lane per track: loads K, x_b, P_b and the filtered row a step ahead, runs the recurrence.
// Same device functions, same arithmetic, same bits as urtss_recur_l1.  The work rows hold K afterwards: the word that keeps
// a track's first bad square root is stored negated (-(v) - 1) by the second kernel to say so, and a repeated
// ste_urtss_backward_f64 on the same forward result skips the first kernel (the call stays repeatable).  Not for large
// batches: 440 B more traffic per track-step, which a batch that fills the chip cannot afford (launch_backward).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kLeanK01 = kWorkD, kLeanK23 = kWorkD23;  // K[r][0:2] at kLeanK01 + 2 r, K[r][2:4] at kLeanK23 + 2 r

template <bool kShift>
__global__ __launch_bounds__(64) void urtss_gains_all(const KParams p) {
    const size_t B = (size_t)p.ld;
    const size_t g = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (g >= (size_t)p.B * (size_t)p.Nmax) return;
    const size_t t = g % (size_t)p.B;
    const int k = (int)(g / (size_t)p.B);
    const int ns = p.nsteps ? p.nsteps[t] : p.Nmax;
    if (k >= ns) return;
    const double fb_raw = p.first_bad[t];
    if (fb_raw < 0.0) return;  // the rows of this track already hold gains (a repeated backward call)
    const bool always_full = p.noise_pred || p.noise_upd || p.noise_rts;
    const bool all_eig = (p.tuning & 0x100) != 0;
    const double kappa = (p.m.wi + p.m.wi) * p.m.fan_scale;
    const bool packed = (p.flags & STE_FLAG_PACKED_COV) != 0;
    RecurRow cur;
    load_recur_row<kShift>(p, (size_t)k, B, t, always_full, cur);
    double xn[4], Pn[10];
    load_vec(p.fwd_mean, (size_t)k + 1, B, t, xn);
    load_cov_p(p.fwd_cov, packed, (size_t)k + 1, B, t, Pn);
    const bool full = work_row_full(p, k, p.upd_idx[(size_t)k * B + t], always_full);
    double xb[4], Pb[10], K[4][4];
    const int st = smoother_step_gain<kShift>(p, k, B, t, cur, xn, Pn, full, always_full, all_eig, kappa, fb_raw, xb, Pb, K);
    double* w = p.rts_work + ((size_t)k * kWorkElems) * B + t;
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        w[(kLeanK01 + 2 * r + 0) * B] = K[r][0];
        w[(kLeanK01 + 2 * r + 1) * B] = K[r][1];
        w[(kLeanK23 + 2 * r + 0) * B] = K[r][2];
        w[(kLeanK23 + 2 * r + 1) * B] = K[r][3];
    }
    STE_UNROLL
    for (int c = 0; c < 4; ++c) w[(kWorkXb + c) * B] = xb[c];
    STE_UNROLL
    for (int e = 0; e < 10; ++e) w[(kWorkPb + e) * B] = Pb[e];
    if (st) atomicOr(&p.status[t], st);
}

struct LeanRow {
    double K[4][4], xb[4], Pb[10], xk[4], Pk[10];
};
__device__ __forceinline__ void load_lean_row(const KParams& p, size_t k, size_t B, size_t t, bool packed, LeanRow& g) {
    const double* w = p.rts_work + (k * kWorkElems) * B + t;
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        g.K[r][0] = w[(kLeanK01 + 2 * r + 0) * B];
        g.K[r][1] = w[(kLeanK01 + 2 * r + 1) * B];
        g.K[r][2] = w[(kLeanK23 + 2 * r + 0) * B];
        g.K[r][3] = w[(kLeanK23 + 2 * r + 1) * B];
    }
    STE_UNROLL
    for (int c = 0; c < 4; ++c) g.xb[c] = w[(kWorkXb + c) * B];
    STE_UNROLL
    for (int e = 0; e < 10; ++e) g.Pb[e] = w[(kWorkPb + e) * B];
    load_vec(p.fwd_mean, k, B, t, g.xk);
    load_cov_p(p.fwd_cov, packed, k, B, t, g.Pk);
}

__global__ __launch_bounds__(64) void urtss_recur_lean(const KParams p) {
    const size_t B = (size_t)p.ld;
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= (size_t)p.B) return;
    const int ns = p.nsteps ? p.nsteps[t] : p.Nmax;
    const bool packed = (p.flags & STE_FLAG_PACKED_COV) != 0;
    const int last_row = p.Nmax > 0 ? p.Nmax - 1 : 0;
    {
        const double fb = p.first_bad[t];
        if (fb >= 0.0) p.first_bad[t] = -fb - 1.0;  // the work rows of this track hold gains from here on (see above)
    }
    double xs[4], Ps[10];
    load_vec(p.fwd_mean, (size_t)ns, B, t, xs);
    load_cov_p(p.fwd_cov, packed, (size_t)ns, B, t, Ps);
    store_vec(p.sm_mean, (size_t)ns, B, t, xs);
    store_cov_p(p.sm_cov, packed, (size_t)ns, B, t, Ps);
    store_pos(p, (size_t)ns, B, t, xs);
    LeanRow nxt;
    if (ns > 0) load_lean_row(p, (size_t)(ns - 1), B, t, packed, nxt);
    for (int k = p.Nmax - 1; k >= 0; --k) {
        if (!__any(k < ns)) continue;
        if (k < ns) {
            const LeanRow cur = nxt;
            load_lean_row(p, (size_t)min(max(k - 1, 0), last_row), B, t, packed, nxt);  // in flight during this step's arithmetic
            double y[4];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) y[c] = xs[c] - cur.xb[c];
            y[3] = wrap180(y[3]);
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                double acc = cur.xk[r];
                STE_UNROLL
                for (int c = 0; c < 4; ++c) acc = fma(cur.K[r][c], y[c], acc);
                xs[r] = acc;
            }
            xs[3] = floored_mod(xs[3], 360.0);
            double dP[10];
            STE_UNROLL
            for (int e = 0; e < 10; ++e) dP[e] = Ps[e] - cur.Pb[e];
            double KdP[4][4];
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                STE_UNROLL
                for (int c = 0; c < 4; ++c) {
                    double acc = cur.K[r][0] * dP[tix(0, c)];
                    STE_UNROLL
                    for (int i = 1; i < 4; ++i) acc = fma(cur.K[r][i], dP[tix(i, c)], acc);
                    KdP[r][c] = acc;
                }
            }
            STE_UNROLL
            for (int r = 0; r < 4; ++r) {
                STE_UNROLL
                for (int c = r; c < 4; ++c) {
                    double acc = KdP[r][0] * cur.K[c][0];
                    STE_UNROLL
                    for (int i = 1; i < 4; ++i) acc = fma(KdP[r][i], cur.K[c][i], acc);
                    Ps[tix(r, c)] = cur.Pk[tix(r, c)] + acc;
                }
            }
            store_vec(p.sm_mean, (size_t)k, B, t, xs);
            store_cov_p(p.sm_cov, packed, (size_t)k, B, t, Ps);
            store_pos(p, (size_t)k, B, t, xs);
        }
    }
    double chk = 0.0;
    STE_UNROLL
    for (int c = 0; c < 4; ++c) chk += xs[c] * 0.0;
    STE_UNROLL
    for (int e = 0; e < 10; ++e) chk += Ps[e] * 0.0;
    if (!(chk == 0.0)) atomicOr(&p.status[t], STE_STATUS_NAN);
}

// The lean recurrence with ONE DPP QUAD PER TRACK.  A block of a few tracks runs urtss_recur_lean at the pace at which one
// wave issues its ~450 instructions per step (44 loads and 16-22 stores with their 64-bit address arithmetic, ~180 fp64
// operations): 1.3 us per step on configs[3], and neither deeper prefetch nor a run-ahead wave warming L2 nor plain instead
// of nontemporal stores moved it -- it is issue, not latency.  Here lane q of a quad owns row q: K[q][:], x[q], and the part
// (q, c >= q) of the packed covariances; y and dP are broadcast through the quad, the K rows of the lanes to the right come
// by quad rotation: ~60 fp64 operations, 52 DPP moves, 14 loads and <= 9 stores per lane and step.  Every element is
// computed by exactly one lane with urtss_recur_lean's sequence of operations: same bits (tests/test_fleet.py).
struct LeanRowQ {
    double K[4], xb, xk, Pb[4], Pk[4];  // row q of K; element q of x_b and x_k; slot j = element (q, q + j) of P_b and P_k
};
__device__ __forceinline__ int pidx(int r, int c) { return r * 4 - (r * (r - 1)) / 2 + (c - r); }  // tix for r <= c at run time
__device__ __forceinline__ void load_lean_row_q(const KParams& p, size_t k, size_t B, size_t t, int q, bool packed, LeanRowQ& g) {
    const double* w = p.rts_work + (k * kWorkElems) * B + t;
    g.K[0] = w[(size_t)(kLeanK01 + 2 * q + 0) * B];
    g.K[1] = w[(size_t)(kLeanK01 + 2 * q + 1) * B];
    g.K[2] = w[(size_t)(kLeanK23 + 2 * q + 0) * B];
    g.K[3] = w[(size_t)(kLeanK23 + 2 * q + 1) * B];
    g.xb = w[(size_t)(kWorkXb + q) * B];
    g.xk = p.fwd_mean[(k * 4 + (size_t)q) * B + t];
    STE_UNROLL
    for (int j = 0; j < 4; ++j) {
        const int c = min(q + j, 3);  // slots past the row's end repeat its last element and are never used
        g.Pb[j] = w[(size_t)(kWorkPb + pidx(q, c)) * B];
        g.Pk[j] = p.fwd_cov[(packed ? k * 10 + (size_t)pidx(q, c) : k * 16 + (size_t)(q * 4 + c)) * B + t];
    }
}
// x[q], P(q, q + j) -> history row `row` (both triangles of the full layout: the lane that computed (r, c) also stores (c, r))
__device__ __forceinline__ void store_lean_row_q(const KParams& p, size_t row, size_t B, size_t t, int q, bool packed, double x,
                                                 const double (&P)[4]) {
    st_stream(&p.sm_mean[(row * 4 + (size_t)q) * B + t], x);
    if (p.sm_pos && q < 2) st_stream(&p.sm_pos[(row * 2 + (size_t)q) * B + t], x);
    STE_UNROLL
    for (int j = 0; j < 4; ++j) {
        const int c = q + j;
        if (c < 4) {
            if (packed) {
                st_stream(&p.sm_cov[(row * 10 + (size_t)pidx(q, c)) * B + t], P[j]);
            } else {
                st_stream(&p.sm_cov[(row * 16 + (size_t)(q * 4 + c)) * B + t], P[j]);
                if (j) st_stream(&p.sm_cov[(row * 16 + (size_t)(c * 4 + q)) * B + t], P[j]);
            }
        }
    }
}
// value held by lane (q + J) mod 4 of the quad
template <int J>
__device__ __forceinline__ double quad_rot(double v) {
    static_assert(J >= 1 && J <= 3, "quad rotation");
    return dpp_move<(J == 1) ? 0x39 : (J == 2) ? 0x4E : 0x93>(v);  // quad_perm [1,2,3,0] / [2,3,0,1] / [3,0,1,2]
}

__global__ __launch_bounds__(64) void urtss_recur_lean_q4(const KParams p) {
    const size_t B = (size_t)p.ld;
    const int q = threadIdx.x & 3;
    const size_t t = (size_t)blockIdx.x * 16 + (threadIdx.x >> 2);
    if (t >= (size_t)p.B) return;  // whole quads leave
    const int ns = p.nsteps ? p.nsteps[t] : p.Nmax;
    const bool packed = (p.flags & STE_FLAG_PACKED_COV) != 0;
    const int last_row = p.Nmax > 0 ? p.Nmax - 1 : 0;
    if (q == 0) {
        const double fb = p.first_bad[t];
        if (fb >= 0.0) p.first_bad[t] = -fb - 1.0;  // the work rows of this track hold gains from here on (urtss_recur_lean)
    }
    double xs, Ps[4];  // x^s[q]; slot j = P^s(q, q + j)
    xs = p.fwd_mean[((size_t)ns * 4 + (size_t)q) * B + t];
    STE_UNROLL
    for (int j = 0; j < 4; ++j) {
        const int c = min(q + j, 3);
        Ps[j] = p.fwd_cov[(packed ? (size_t)ns * 10 + (size_t)pidx(q, c) : (size_t)ns * 16 + (size_t)(q * 4 + c)) * B + t];
    }
    store_lean_row_q(p, (size_t)ns, B, t, q, packed, xs, Ps);
    LeanRowQ nxt;
    if (ns > 0) load_lean_row_q(p, (size_t)(ns - 1), B, t, q, packed, nxt);
    for (int k = p.Nmax - 1; k >= 0; --k) {
        if (!__any(k < ns)) continue;
        if (k < ns) {
            const LeanRowQ cur = nxt;
            load_lean_row_q(p, (size_t)min(max(k - 1, 0), last_row), B, t, q, packed, nxt);  // in flight during this step
            // y = x^s_{k+1} - x_b, the course difference wrapped (lane 3's element)
            double yq = xs - cur.xb;
            {
                const double yw = wrap180(yq);
                yq = (q == 3) ? yw : yq;
            }
            const double y0 = bcast<0>(yq), y1 = bcast<1>(yq), y2 = bcast<2>(yq), y3 = bcast<3>(yq);
            double acc = cur.xk;
            acc = fma(cur.K[0], y0, acc);
            acc = fma(cur.K[1], y1, acc);
            acc = fma(cur.K[2], y2, acc);
            acc = fma(cur.K[3], y3, acc);
            {
                const double am = floored_mod(acc, 360.0);
                xs = (q == 3) ? am : acc;
            }
            // dP = P^s_{k+1} - P_b on the owner of each element, then the whole symmetric matrix to every lane of the quad
            double d[4];
            STE_UNROLL
            for (int j = 0; j < 4; ++j) d[j] = Ps[j] - cur.Pb[j];
            double dP[10];
            dP[tix(0, 0)] = bcast<0>(d[0]);
            dP[tix(0, 1)] = bcast<0>(d[1]);
            dP[tix(0, 2)] = bcast<0>(d[2]);
            dP[tix(0, 3)] = bcast<0>(d[3]);
            dP[tix(1, 1)] = bcast<1>(d[0]);
            dP[tix(1, 2)] = bcast<1>(d[1]);
            dP[tix(1, 3)] = bcast<1>(d[2]);
            dP[tix(2, 2)] = bcast<2>(d[0]);
            dP[tix(2, 3)] = bcast<2>(d[1]);
            dP[tix(3, 3)] = bcast<3>(d[0]);
            // row q of K dP
            double KdP[4];
            STE_UNROLL
            for (int c = 0; c < 4; ++c) {
                double a2 = cur.K[0] * dP[tix(0, c)];
                STE_UNROLL
                for (int i = 1; i < 4; ++i) a2 = fma(cur.K[i], dP[tix(i, c)], a2);
                KdP[c] = a2;
            }
            // P^s(q, q + j) = P_k(q, q + j) + sum_i KdP[q][i] K[q + j][i]: the K row of the lane j places to the right
            {
                double a2 = KdP[0] * cur.K[0];
                STE_UNROLL
                for (int i = 1; i < 4; ++i) a2 = fma(KdP[i], cur.K[i], a2);
                Ps[0] = cur.Pk[0] + a2;
            }
            {
                const double k0 = quad_rot<1>(cur.K[0]), k1 = quad_rot<1>(cur.K[1]), k2 = quad_rot<1>(cur.K[2]), k3 = quad_rot<1>(cur.K[3]);
                double a2 = KdP[0] * k0;
                a2 = fma(KdP[1], k1, a2);
                a2 = fma(KdP[2], k2, a2);
                a2 = fma(KdP[3], k3, a2);
                Ps[1] = cur.Pk[1] + a2;
            }
            {
                const double k0 = quad_rot<2>(cur.K[0]), k1 = quad_rot<2>(cur.K[1]), k2 = quad_rot<2>(cur.K[2]), k3 = quad_rot<2>(cur.K[3]);
                double a2 = KdP[0] * k0;
                a2 = fma(KdP[1], k1, a2);
                a2 = fma(KdP[2], k2, a2);
                a2 = fma(KdP[3], k3, a2);
                Ps[2] = cur.Pk[2] + a2;
            }
            {
                const double k0 = quad_rot<3>(cur.K[0]), k1 = quad_rot<3>(cur.K[1]), k2 = quad_rot<3>(cur.K[2]), k3 = quad_rot<3>(cur.K[3]);
                double a2 = KdP[0] * k0;
                a2 = fma(KdP[1], k1, a2);
                a2 = fma(KdP[2], k2, a2);
                a2 = fma(KdP[3], k3, a2);
                Ps[3] = cur.Pk[3] + a2;
            }
            store_lean_row_q(p, (size_t)k, B, t, q, packed, xs, Ps);
        }
    }
    double chk = xs * 0.0;
    STE_UNROLL
    for (int j = 0; j < 4; ++j)
        if (q + j < 4) chk += Ps[j] * 0.0;
    if (!(chk == 0.0)) atomicOr(&p.status[t], STE_STATUS_NAN);
}

// ---------------------------------------------------------------------------------------------------------------
// single-function kernels (fine-grained API parity: geodetic_dynamics, compute_sigma_points)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void geodetic_kernel(size_t count, const double* x, const double* dt,
                                                      const double* sr, const double* cr, double* out) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    double xi[4], o[4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) xi[c] = x[c * count + i];
    geodetic_step(xi, dt[i], sr[i], cr[i], o);
    STE_UNROLL
    for (int c = 0; c < 4; ++c) out[c * count + i] = o[c];
}

__global__ __launch_bounds__(64) void sigma_points_kernel(size_t count, const double* x, const double* P,
                                                          double scale, double* out) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    double xi[4], Pi[4][4], sig[9][4];
    STE_UNROLL
    for (int c = 0; c < 4; ++c) xi[c] = x[c * count + i];
    load_mat(P, 0, count, i, Pi);
    sigma_fan(xi, Pi, scale, sig);
    STE_UNROLL
    for (int j = 0; j < 9; ++j) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) out[((size_t)j * 4 + c) * count + i] = sig[j][c];
    }
}

// Sigma fan for a general state dimension n <= kMaxGenericN (unscented.py:76-107 accepts any n; the reference's own
// unit tests use n = 2).  Not a hot path: one lane per matrix, run-time loops, arrays in scratch.
constexpr int kMaxGenericN = 16;
__global__ __launch_bounds__(64) void sigma_points_generic_kernel(int n, size_t count, const double* x,
                                                                  const double* P, double scale, double* out) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    double A[kMaxGenericN][kMaxGenericN], V[kMaxGenericN][kMaxGenericN];
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            A[r][c] = scale * 0.5 * (P[((size_t)r * n + c) * count + i] + P[((size_t)c * n + r) * count + i]);
            V[r][c] = (r == c) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p][q], app = A[p][p], aqq = A[q][q];
                if (!(apq * apq > kRotTol2 * fabs(app * aqq))) continue;
                rotated = true;
                const double delta = aqq - app;
                double t = 2.0 * apq / (fabs(delta) + sqrt(delta * delta + 4.0 * apq * apq));
                t = delta < 0.0 ? -t : t;
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int r = 0; r < n; ++r) {  // columns p, q
                    const double arp = A[r][p], arq = A[r][q];
                    A[r][p] = c * arp - s * arq;
                    A[r][q] = s * arp + c * arq;
                    const double vrp = V[r][p], vrq = V[r][q];
                    V[r][p] = c * vrp - s * vrq;
                    V[r][q] = s * vrp + c * vrq;
                }
                for (int r = 0; r < n; ++r) {  // rows p, q
                    const double apr = A[p][r], aqr = A[q][r];
                    A[p][r] = c * apr - s * aqr;
                    A[q][r] = s * apr + c * aqr;
                }
                A[p][q] = 0.0;
                A[q][p] = 0.0;
            }
        if (!rotated) break;
    }
    const size_t m = 2 * (size_t)n + 1;
    for (int c = 0; c < n; ++c) out[((size_t)0 * n + c) * count + i] = x[(size_t)c * count + i];
    for (int col = 0; col < n; ++col)
        for (int r = 0; r < n; ++r) {
            double t = 0.0;  // T[r][col] = sum_l V[r][l] sqrt(max(w_l,0)) V[col][l]
            for (int l = 0; l < n; ++l) t += V[r][l] * sqrt(fmax(A[l][l], 0.0)) * V[col][l];
            const double xr = x[(size_t)r * count + i];
            out[((size_t)(1 + col) * n + r) * count + i] = xr + t;
            out[((size_t)(1 + n + col) * n + r) * count + i] = xr - t;
        }
    (void)m;
}

// One predict (unscented.py:144-207) / one update (:209-265) on `count` independent (x, P) pairs.
__global__ __launch_bounds__(64) void predict_kernel(size_t count, const Mats m, const double* x, const double* P,
                                                     const double* dt, const double* sr, const double* cr,
                                                     const double* noise, double* x_out, double* P_out,
                                                     int32_t* status) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    double xi[4], Pi[4][4];
    load_vec(x, 0, count, i, xi);
    load_mat(P, 0, count, i, Pi);
    int st = ukf_predict(m, xi, Pi, dt[i], sr[i], cr[i], noise, 0, count, i);
    if (!all_finite(xi, Pi)) st |= STE_STATUS_NAN;
    store_vec(x_out, 0, count, i, xi);
    store_mat(P_out, 0, count, i, Pi);
    if (status) status[i] = st;
}

__global__ __launch_bounds__(64) void update_kernel(size_t count, const Mats m, const double* x, const double* P,
                                                    const double* z, const double* noise, double* x_out,
                                                    double* P_out, int32_t* status) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    double xi[4], Pi[4][4], zi[4];
    load_vec(x, 0, count, i, xi);
    load_mat(P, 0, count, i, Pi);
    load_vec(z, 0, count, i, zi);
    int st = ukf_update(m, xi, Pi, zi, noise, 0, count, i);
    if (!all_finite(xi, Pi)) st |= STE_STATUS_NAN;
    store_vec(x_out, 0, count, i, xi);
    store_mat(P_out, 0, count, i, Pi);
    if (status) status[i] = st;
}

// Robustification terms for `count` independent (x, P, z) triples (criterion_index / update_lambda_factor).
__global__ __launch_bounds__(64) void robust_terms_kernel(size_t count, const Mats m, const double* x, const double* P,
                                                          const double* z, double* gamma, double* denom) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    double H[4][4], R[4][4], xi[4], Pi[4][4], zi[4];
    STE_UNROLL
    for (int r = 0; r < 4; ++r) {
        STE_UNROLL
        for (int c = 0; c < 4; ++c) {
            H[r][c] = m.H[r * 4 + c];
            R[r][c] = m.R[r * 4 + c];
        }
    }
    load_vec(x, 0, count, i, xi);
    load_mat(P, 0, count, i, Pi);
    load_vec(z, 0, count, i, zi);
    double g, d;
    robust_terms(H, R, xi, Pi, zi, g, d);
    gamma[i] = g;
    denom[i] = d;
}

}  // namespace ste

// ===============================================================================================================
// C ABI
// ===============================================================================================================
namespace {

thread_local char g_err[512] = "";
// Lane mapping of the forward kernel.  A quad per track shortens the per-wave instruction stream ~1.7x and puts 4x as
// many waves on the chip, but replicates work across its lanes.  The quad forward kernel fits two waves on a SIMD
// (<= 256 VGPRs), so it stays a single round up to 32 768 tracks (2 048 waves), after which one lane per track wins
// (measured on MI355X).  STE_FLAG_LANES_1 / STE_FLAG_LANES_4 in the batch's flags override the choice for
// that call.
constexpr int kQuadMaxTracks = 32768;
constexpr size_t kQuadSpreadWaves = 1024;  // SIMDs of an MI355X: up to this many tracks, one quad (track) per wave
int choose_lanes(int B, unsigned flags) {
    if (flags & STE_FLAG_LANES_1) return 1;
    if (flags & STE_FLAG_LANES_4) return 4;
    return B <= kQuadMaxTracks ? 4 : 1;
}

int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return STE_OK;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return STE_ELAUNCH;
}
}  // namespace

// ste_err.h: the same error string for the other translation units of this ABI (ste_prep.hip)
int ste::abi_fail(int code, const char* msg) { return fail(code, "%s", msg); }
int ste::abi_check_hip(hipError_t e, const char* what) { return check_hip(e, what); }

namespace {

int make_params(const ste_ukf_batch_f64* b, bool need_fwd_in, bool need_sm_out, ste::KParams* kp) {
    if (!b) return fail(STE_EINVAL, "batch pointer is NULL");
    if (b->n != 4) return fail(STE_EINVAL, "state dimension n must be 4 (heading index 3 is hard-wired, unscented.py:250)");
    if (b->B <= 0) return fail(STE_EINVAL, "B must be > 0");
    if (b->Nmax < 0) return fail(STE_EINVAL, "Nmax must be >= 0");
    if (b->Tmax < 1) return fail(STE_EINVAL, "Tmax must be >= 1");
    if (!b->H || !b->Q || !b->R) return fail(STE_EINVAL, "H, Q and R (host 4x4) are required");
    if (!b->x0 || !b->P0) return fail(STE_EINVAL, "x0 and P0 are required");
    if (b->Nmax > 0 && (!b->dt || !b->sog_rate || !b->cog_rate || !b->upd_idx))
        return fail(STE_EINVAL, "dt, sog_rate, cog_rate and upd_idx are required when Nmax > 0");
    if (!b->z) return fail(STE_EINVAL, "z is required");
    if (!b->fwd_mean || !b->fwd_cov || !b->status) return fail(STE_EINVAL, "fwd_mean, fwd_cov and status are required");
    if (need_sm_out && (!b->sm_mean || !b->sm_cov)) return fail(STE_EINVAL, "sm_mean and sm_cov are required");
    (void)need_fwd_in;
    kp->B = b->B;
    kp->Nmax = b->Nmax;
    kp->Tmax = b->Tmax;
    if ((b->flags & STE_FLAG_LANES_1) && (b->flags & STE_FLAG_LANES_4))
        return fail(STE_EINVAL, "STE_FLAG_LANES_1 and STE_FLAG_LANES_4 exclude each other");
    kp->flags = b->flags;
    kp->tuning = b->tuning;
    kp->m.fan_scale = b->fan_scale;
    kp->m.w0 = b->w0;
    kp->m.wi = b->wi;
    memcpy(kp->m.H, b->H, sizeof(double) * 16);
    memcpy(kp->m.Q, b->Q, sizeof(double) * 16);
    memcpy(kp->m.R, b->R, sizeof(double) * 16);
    kp->m.chi_alpha = b->chi_alpha;
    kp->m.robust_iters = (b->flags & STE_FLAG_ROBUST) ? (b->robust_max_iter > 0 ? b->robust_max_iter : 50) : 0;
    // The sigma weights sum to one for every weight0 the reference accepts (unscented.py:125-132: wi = (1 - w0) / 2n);
    // the streamed moments of the forward kernels rely on it.
    if (!(fabs(b->w0 + 8.0 * b->wi - 1.0) <= 1e-14))
        return fail(STE_EINVAL, "sigma weights must sum to one: w0 + 2 n wi = 1 (unscented.py:125-132)");
    {
        bool sel = true;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                sel = sel && b->H[r * 4 + c] == ((r == c && r < 2) ? 1.0 : 0.0);
                if (r >= 2 || c >= 2) sel = sel && b->R[r * 4 + c] == 0.0;
            }
        kp->fast_upd = sel && b->R[1] == b->R[4];
    }
    kp->nsteps = b->nsteps;
    kp->x0 = b->x0;
    kp->P0 = b->P0;
    kp->dt = b->dt;
    kp->sog_rate = b->sog_rate;
    kp->cog_rate = b->cog_rate;
    kp->sog_rate_rts = b->sog_rate_rts;
    kp->cog_rate_rts = b->cog_rate_rts;
    kp->upd_idx = b->upd_idx;
    kp->z = b->z;
    kp->noise_pred = b->noise_pred;
    kp->noise_upd = b->noise_upd;
    kp->noise_rts = b->noise_rts;
    kp->fwd_mean = b->fwd_mean;
    kp->fwd_cov = b->fwd_cov;
    kp->sm_mean = b->sm_mean;
    kp->sm_cov = b->sm_cov;
    kp->status = b->status;
    // the forward pass leaves the smoother's rows (see kWorkD)
    // (smoother rates of its own only move x_b and P_b by a known amount: load_recur_row)
    kp->rts_work = b->rts_work;
    if (b->track_stride != 0 && (b->track_stride < b->B || b->track_stride > 0x7fffffff))
        return fail(STE_EINVAL, "track_stride must be 0 (= B) or in [B, 2^31): a window lies inside the rows of its fleet");
    kp->ld = b->track_stride ? (int)b->track_stride : b->B;
    kp->k0 = 0;
    kp->qpw = 16;
    kp->first_bad = b->rts_work ? b->rts_work + (size_t)b->Nmax * STE_RTS_WORK_ROWS * (size_t)kp->ld : nullptr;
    kp->sm_pos = b->sm_pos;
    return STE_OK;
}

// The forward pass in time slices (ste.h: step_begin / step_end).  A slice that starts at step k0 > 0 is the same kernel
// started from history row k0: x0 / P0 point at that row, every per-step array at its row k0, the histories and work
// rows likewise, the initial update is off and Nmax is the slice's length.  Slice boundaries are multiples of
// STE_SLICE_ALIGN = kColdEvery, the steps at which the eigenvector basis of the fan's square root restarts from the
// identity in a whole pass too -- so the history row is the complete state and the slices reproduce the whole pass bit
// for bit.  (The quad mapping keeps both triangles of P in its lanes and a packed history holds one: slices of that
// combination are refused.)
static_assert(STE_SLICE_ALIGN == ste::kColdEvery, "include/ste.h: STE_SLICE_ALIGN");
int slice_params(const ste_ukf_batch_f64* b, ste::KParams* kp) {
    const int k0 = b->step_begin, k1 = b->step_end ? b->step_end : b->Nmax;
    if (k0 < 0 || k1 > b->Nmax || k0 > k1 || (k0 == k1 && b->Nmax > 0))
        return fail(STE_EINVAL, "time slice: need 0 <= step_begin < step_end <= Nmax (step_end 0 = Nmax)");
    if (k0 % STE_SLICE_ALIGN != 0 || (k1 != b->Nmax && k1 % STE_SLICE_ALIGN != 0))
        return fail(STE_EINVAL, "time slice: step_begin and step_end must be multiples of STE_SLICE_ALIGN (64) or the ends of the pass");
    if (k0 == 0 && k1 == b->Nmax) return STE_OK;
    const bool packed = (b->flags & STE_FLAG_PACKED_COV) != 0;
    if (packed && choose_lanes(kp->B, kp->flags) == 4)
        return fail(STE_EINVAL, "time slices with packed covariances need the lane-per-track mapping (STE_FLAG_LANES_1)");
    const size_t ld = (size_t)kp->ld, r0 = (size_t)k0;
    kp->Nmax = k1 - k0;
    kp->k0 = k0;
    if (k0 > 0) {
        kp->flags = (kp->flags | ste::kFlagContinue | STE_FLAG_NO_INITIAL_UPDATE) & ~STE_FLAG_SHARED_P0;
        kp->x0 = b->fwd_mean + r0 * 4 * ld;
        kp->P0 = b->fwd_cov + r0 * (packed ? 10 : 16) * ld;
        kp->dt += r0 * ld;
        kp->sog_rate += r0 * ld;
        kp->cog_rate += r0 * ld;
        kp->upd_idx += r0 * ld;
        if (kp->noise_pred) kp->noise_pred += r0 * 4 * ld;
        if (kp->noise_upd) kp->noise_upd += r0 * 4 * ld;  // row k + 1 belongs to the update after step k
        if (kp->noise_rts) kp->noise_rts += r0 * 4 * ld;
        kp->fwd_mean += r0 * 4 * ld;
        kp->fwd_cov += r0 * (packed ? 10 : 16) * ld;
        if (kp->rts_work) kp->rts_work += r0 * STE_RTS_WORK_ROWS * ld;
    }
    return STE_OK;
}

int launch_forward(const ste::KParams& kp, hipStream_t s) {
    const bool robust = kp.m.robust_iters > 0;
    if (choose_lanes(kp.B, kp.flags) == 4) {
        // quads per wave: 16 fill a wave; with fewer tracks than SIMDs to spare every track gets a wave (and a SIMD) of its own,
        // so that no track waits for another's extra sweep, rescaling or slow-path fan (see ukf_forward_q4)
        ste::KParams kq = kp;
        kq.qpw = (int)std::min<size_t>(16, std::max<size_t>(1, ((size_t)kp.B + kQuadSpreadWaves - 1) / kQuadSpreadWaves));
        const unsigned gridq = (unsigned)(((size_t)kp.B + kq.qpw - 1) / kq.qpw);
        const int which = (robust ? 4 : 0) | (kp.rts_work ? 2 : 0) | (kp.fast_upd ? 1 : 0);
        switch (which) {
#define STE_Q4(n, g, r, f) \
    case n: hipLaunchKernelGGL((ste::ukf_forward_q4<g, r, f>), dim3(gridq), dim3(64), 0, s, kq); break;
            STE_Q4(0, false, false, false) STE_Q4(1, false, false, true) STE_Q4(2, true, false, false) STE_Q4(3, true, false, true)
            STE_Q4(4, false, true, false) STE_Q4(5, false, true, true) STE_Q4(6, true, true, false) STE_Q4(7, true, true, true)
#undef STE_Q4
        }
        return check_hip(hipGetLastError(), "ukf_forward_q4 launch");
    }
    const unsigned grid = (unsigned)((kp.B + 63) / 64);
    if (kp.fast_upd && robust) {
        if (kp.rts_work)
            hipLaunchKernelGGL((ste::ukf_forward_l1<true, true, true>), dim3(grid), dim3(64), 0, s, kp);
        else
            hipLaunchKernelGGL((ste::ukf_forward_l1<false, true, true>), dim3(grid), dim3(64), 0, s, kp);
    } else if (kp.fast_upd) {
        if (kp.rts_work)
            hipLaunchKernelGGL((ste::ukf_forward_l1<true, true>), dim3(grid), dim3(64), 0, s, kp);
        else
            hipLaunchKernelGGL((ste::ukf_forward_l1<false, true>), dim3(grid), dim3(64), 0, s, kp);
    } else {
        if (kp.rts_work)
            hipLaunchKernelGGL((ste::ukf_forward_l1<true, false>), dim3(grid), dim3(64), 0, s, kp);
        else
            hipLaunchKernelGGL((ste::ukf_forward_l1<false, false>), dim3(grid), dim3(64), 0, s, kp);
    }
    return check_hip(hipGetLastError(), "ukf_forward launch");
}

// Batches of at most this many tracks smooth with the two-kernel form (urtss_gains_all + urtss_recur_lean): up to there the
// one-kernel smoother is a few waves running a latency chain of ~1.5-4.7 us per step, and the extra 440 B per track-step of
// the gains pass cost less than the chain they remove; a batch that fills the chip is bound by bytes and issue instead.
// tuning bit 9 (0x200) forces the two-kernel form, bit 10 (0x400) the one-kernel form (tests, measurements); the choice must
// not change between the backward calls made on one forward result (the first two-kernel call turns the work rows into gains).
constexpr int kLeanSmootherMaxTracks = 4096;

int launch_backward(const ste::KParams& kp, hipStream_t s) {
    const unsigned grid = (unsigned)((kp.B + 63) / 64);
    const bool shift = kp.sog_rate_rts || kp.cog_rate_rts;
    if (kp.rts_work && kp.Nmax > 0 && ((kp.tuning & 0x200) || (!(kp.tuning & 0x400) && kp.B <= kLeanSmootherMaxTracks))) {
        const size_t lanes = (size_t)kp.B * (size_t)kp.Nmax;
        const unsigned ggrid = (unsigned)((lanes + 63) / 64);
        if (shift)
            hipLaunchKernelGGL(ste::urtss_gains_all<true>, dim3(ggrid), dim3(64), 0, s, kp);
        else
            hipLaunchKernelGGL(ste::urtss_gains_all<false>, dim3(ggrid), dim3(64), 0, s, kp);
        int rc = check_hip(hipGetLastError(), "urtss_gains_all launch");
        if (rc) return rc;
        // the recurrence: a quad per track (16 tracks per wave); tuning bit 11 (0x800) keeps the lane-per-track form
        if (kp.tuning & 0x800)
            hipLaunchKernelGGL(ste::urtss_recur_lean, dim3(grid), dim3(64), 0, s, kp);
        else
            hipLaunchKernelGGL(ste::urtss_recur_lean_q4, dim3((unsigned)((kp.B + 15) / 16)), dim3(64), 0, s, kp);
        return check_hip(hipGetLastError(), "urtss_recur_lean launch");
    }
    if (kp.rts_work) {
        if (kp.sog_rate_rts || kp.cog_rate_rts)
            hipLaunchKernelGGL(ste::urtss_recur_l1<true>, dim3(grid), dim3(64), 0, s, kp);
        else
            hipLaunchKernelGGL(ste::urtss_recur_l1<false>, dim3(grid), dim3(64), 0, s, kp);
        return check_hip(hipGetLastError(), "urtss_recur launch");
    }
    hipLaunchKernelGGL(ste::urtss_backward_l1, dim3(grid), dim3(64), 0, s, kp);
    return check_hip(hipGetLastError(), "urtss_backward launch");
}

// which instantiation of the lane-per-track forward kernel a batch takes (launch_forward's own choice, as a number)
int forward_variant(const ste::KParams& kp) {
    const bool robust = kp.m.robust_iters > 0;
    return (kp.fast_upd ? (robust ? 4 : 2) : 0) | (kp.rts_work ? 1 : 0);
}

size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

struct SchedLayout {
    size_t kps, items, progress, total;  // byte offsets into the workspace (kps and items are uploaded, progress is zeroed)
};
SchedLayout sched_layout(size_t nkp, size_t nitems, size_t ntiles) {
    SchedLayout l;
    l.kps = 0;
    l.items = align16(nkp * sizeof(ste::KParams));
    l.progress = l.items + align16(nitems * sizeof(ste::SchedItem));
    l.total = l.progress + align16(ntiles * sizeof(int));
    return l;
}

// `ncopy` bytes of a schedule's table from the caller's host workspace into the device workspace and `nzero` zero bytes behind
// them (multiples of 16).  Page-locked host memory: a kernel on `s` reads it in place (no copy engine between two launches);
// anything else: a staged copy.
int upload_table(char* dw, const char* hw, size_t ncopy, size_t nzero, hipStream_t s, const char* what) {
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    const bool locked = hipPointerGetAttributes(&attr, hw) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer &&
                        ((uintptr_t)attr.devicePointer & 15) == 0 && ((uintptr_t)dw & 15) == 0;
    if (locked) {
        const size_t nwords = (ncopy + nzero) / 16;
        const unsigned blocks = (unsigned)std::min<size_t>(256, (nwords + 255) / 256);
        hipLaunchKernelGGL(ste::sched_upload, dim3(blocks), dim3(256), 0, s, (uint4*)dw, (const uint4*)attr.devicePointer, ncopy / 16,
                           nzero / 16);
        return check_hip(hipGetLastError(), what);
    }
    (void)hipGetLastError();  // (the attribute query of pageable memory reports an error: not this call's)
    int rc = check_hip(hipMemcpyAsync(dw, hw, ncopy, hipMemcpyHostToDevice, s), what);
    if (rc || !nzero) return rc;
    return check_hip(hipMemsetAsync(dw + ncopy, 0, nzero, s), what);
}

}  // namespace

extern "C" {

int ste_version(void) { return STE_VERSION; }

#ifdef STE_DEBUG_SWEEPS
int ste_dbg_counters(unsigned long long* out, int reset) {  // developer instrumentation, not part of the ABI
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(ste::g_dbg), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ste::g_dbg), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

const char* ste_last_error(void) { return g_err; }

int ste_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ste_stream_create_cu_range(int32_t first_cu, int32_t num_cus, void** stream) {
    if (!stream) return fail(STE_EINVAL, "stream out-pointer is NULL");
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return fail(STE_ENOGPU, "no HIP device");
    const int ncu = prop.multiProcessorCount;
    if (first_cu < 0 || num_cus < 1 || first_cu + num_cus > ncu)
        return fail(STE_EINVAL, "CU range must lie inside [0, multiProcessorCount)");
    uint32_t mask[32];
    memset(mask, 0, sizeof(mask));
    const int words = (ncu + 31) / 32;
    if (words > 32) return fail(STE_EINVAL, "device has more than 1024 CUs");
    for (int i = first_cu; i < first_cu + num_cus; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t s = nullptr;
    int rc = check_hip(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask), "hipExtStreamCreateWithCUMask");
    if (rc) return rc;
    *stream = (void*)s;
    return STE_OK;
}

int ste_stream_destroy(void* stream) {
    if (!stream) return STE_OK;
    return check_hip(hipStreamDestroy((hipStream_t)stream), "hipStreamDestroy");
}

int ste_ukf_forward_f64(const ste_ukf_batch_f64* b, void* stream) {
    ste::KParams kp;
    int rc = make_params(b, false, false, &kp);
    if (rc) return rc;
    rc = slice_params(b, &kp);
    if (rc) return rc;
    return launch_forward(kp, (hipStream_t)stream);
}

size_t ste_ukf_forward_sched_workspace(int32_t nwindows, int32_t max_slices, int64_t ntiles_total, int32_t nrounds,
                                       int32_t nwaves) {
    if (nwindows < 0 || max_slices < 0 || ntiles_total < 0 || nrounds < 0 || nwaves < 0) return 0;
    // a parameter block per (window, run of consecutive slices): max_slices (max_slices + 1) / 2 of them per window
    return sched_layout((size_t)nwindows * ((size_t)max_slices * ((size_t)max_slices + 1) / 2), (size_t)nrounds * (size_t)nwaves,
                        (size_t)ntiles_total).total;
}

int ste_ukf_forward_sched_f64(const ste_fwd_sched_f64* sc, void* stream) {
    if (!sc) return fail(STE_EINVAL, "schedule pointer is NULL");
    if (sc->nwindows < 1 || !sc->windows) return fail(STE_EINVAL, "scheduled forward pass: nwindows >= 1 and windows are required");
    if (sc->nwaves < 1 || sc->nrounds < 1 || !sc->items) return fail(STE_EINVAL, "scheduled forward pass: nwaves, nrounds >= 1 and items are required");
    if (!sc->host_ws || !sc->dev_ws || !sc->window_done || !sc->error)
        return fail(STE_EINVAL, "scheduled forward pass: host_ws, dev_ws, window_done and error are required");
    const int step = sc->slice_steps ? sc->slice_steps : STE_SLICE_ALIGN;
    if (step < STE_SLICE_ALIGN || step % STE_SLICE_ALIGN != 0)
        return fail(STE_EINVAL, "scheduled forward pass: slice_steps must be a positive multiple of STE_SLICE_ALIGN (64)");
    if (sc->nwindows >= (1 << 21)) return fail(STE_EINVAL, "scheduled forward pass: too many windows");
    // per window: slices, tiles, and where its KParams and progress counters start
    std::vector<int> nslices((size_t)sc->nwindows), kp0((size_t)sc->nwindows), tile0((size_t)sc->nwindows + 1);
    size_t nkp = 0, ntiles = 0;
    int max_slices = 0;
    for (int w = 0; w < sc->nwindows; ++w) {
        const ste_ukf_batch_f64& b = sc->windows[w];
        if (b.B <= 0 || b.Nmax < 0) return fail(STE_EINVAL, "scheduled forward pass: a window has B <= 0 or Nmax < 0");
        if (b.step_begin != 0 || b.step_end != 0) return fail(STE_EINVAL, "scheduled forward pass: windows must not carry a step range of their own");
        if (b.flags & STE_FLAG_LANES_4) return fail(STE_EINVAL, "scheduled forward pass: lane-per-track only (STE_FLAG_LANES_4 is set)");
        nslices[w] = std::max(1, (b.Nmax + step - 1) / step);
        if (nslices[w] > 255) return fail(STE_EINVAL, "scheduled forward pass: more than 255 time slices per window; raise slice_steps");
        max_slices = std::max(max_slices, nslices[w]);
        kp0[w] = (int)nkp;
        tile0[w] = (int)ntiles;
        nkp += (size_t)nslices[w] * ((size_t)nslices[w] + 1) / 2;  // one per run of slices [q0, q1], q0 <= q1
        ntiles += ((size_t)b.B + 63) / 64;
        if (ntiles > 0x7fffffff) return fail(STE_EINVAL, "scheduled forward pass: too many tiles");
    }
    tile0[sc->nwindows] = (int)ntiles;
    const size_t nitems = (size_t)sc->nrounds * (size_t)sc->nwaves;
    const SchedLayout lay = sched_layout(nkp, nitems, ntiles);
    if (sc->ws_bytes < lay.total) return fail(STE_EINVAL, "scheduled forward pass: workspace too small (ste_ukf_forward_sched_workspace)");
    char* hw = (char*)sc->host_ws;
    ste::KParams* kps = (ste::KParams*)(hw + lay.kps);
    // parameter block of window w for the run of slices [q0, q1]: built when a (merged) item first needs it
    int variant = -1;
    std::vector<char> have(nkp, 0);
    auto run_index = [&](int w, int q0, int q1) { return kp0[w] + q0 * nslices[w] - q0 * (q0 - 1) / 2 + (q1 - q0); };
    auto run_params = [&](int w, int q0, int q1, int* index) -> int {
        const int k = run_index(w, q0, q1);
        *index = k;
        if (have[(size_t)k]) return STE_OK;
        ste_ukf_batch_f64 b = sc->windows[w];
        b.flags |= STE_FLAG_LANES_1;
        b.step_begin = q0 * step;
        b.step_end = std::min(b.Nmax, (q1 + 1) * step);
        ste::KParams* kp = kps + k;
        int rc = make_params(&b, false, false, kp);
        if (rc) return rc;
        rc = slice_params(&b, kp);
        if (rc) return rc;
        const int v = forward_variant(*kp);
        if (variant >= 0 && v != variant)
            return fail(STE_EINVAL, "scheduled forward pass: the windows must agree on H / R structure, robust flag and rts_work (one kernel runs them all)");
        variant = v;
        have[(size_t)k] = 1;
        return STE_OK;
    };
    // items: (window, tile) per (round, wave); the slice is the tile's number of earlier appearances.  Every tile must run
    // all its slices, in rounds that strictly increase.
    ste::SchedItem* items = (ste::SchedItem*)(hw + lay.items);
    std::vector<int> next_slice(ntiles, 0), last_round(ntiles, -1), last_wave(ntiles, -1);
    std::vector<size_t> last_item(ntiles, 0);
    for (int r = 0; r < sc->nrounds; ++r)
        for (int v = 0; v < sc->nwaves; ++v) {
            const size_t i = (size_t)r * sc->nwaves + v;
            const int w = sc->items[2 * i], tile = sc->items[2 * i + 1];
            ste::SchedItem it = {-1, 0, 0, 0};
            if (w >= 0) {
                if (w >= sc->nwindows) return fail(STE_EINVAL, "scheduled forward pass: an item names a window that does not exist");
                if (tile < 0 || tile >= tile0[w + 1] - tile0[w]) return fail(STE_EINVAL, "scheduled forward pass: an item names a tile outside its window");
                const int g = tile0[w] + tile, q = next_slice[g];
                if (q >= nslices[w]) return fail(STE_EINVAL, "scheduled forward pass: a tile is scheduled for more slices than it has");
                if (last_round[g] >= r) return fail(STE_EINVAL, "scheduled forward pass: two slices of one tile in the same round");
                it.kp = 0;  // (set below, once runs of slices are known)
                it.tile = tile;
                it.prog = g;
                it.meta = q | (q + 1 == nslices[w] ? 0x100 : 0) | (w << 9);
                if (q + 1 == nslices[w]) it.meta |= 0x40000000;  // the window's smoother reads what the last slice leaves
                if (q > 0 && last_wave[g] != v) {                 // the tile changes waves: hand over through memory
                    it.meta |= (int)0x80000000u;
                    items[last_item[g]].meta |= 0x40000000;
                }
                next_slice[g] = q + 1;
                last_round[g] = r;
                last_wave[g] = v;
                last_item[g] = i;
            }
            items[i] = it;
        }
    for (int w = 0; w < sc->nwindows; ++w)
        for (int g = tile0[w]; g < tile0[w + 1]; ++g)
            if (next_slice[g] != nslices[w]) return fail(STE_EINVAL, "scheduled forward pass: a tile is missing slices (every tile of every window must run all of them)");
    // Runs: consecutive slices of a tile that follow one another on the SAME wave, nothing in between, become one item over
    // the whole step range -- the step loop of a plain forward pass, with no state written out and read back at the slice
    // boundaries inside it (same bits: that is what a slice boundary is).  The run publishes the count of its last slice;
    // the rounds it swallows are idle entries of that wave's column.
    for (int v = 0; v < sc->nwaves; ++v)
        for (int r = 0; r < sc->nrounds;) {
            ste::SchedItem& head = items[(size_t)r * sc->nwaves + v];
            if (head.kp < 0) {
                ++r;
                continue;
            }
            const int w = (head.meta >> 9) & 0x1fffff, q0 = head.meta & 0xff;
            int q1 = q0, r2 = r + 1;
            while (r2 < sc->nrounds) {
                ste::SchedItem& nx = items[(size_t)r2 * sc->nwaves + v];
                if (nx.kp < 0 || nx.prog != head.prog || (nx.meta & 0xff) != q1 + 1 || nx.meta < 0) break;
                // the run keeps its first slice's number, window and acquire flag; last-slice and release flags are its last slice's
                head.meta = (head.meta & (int)(0xffu | (0x1fffffu << 9) | 0x80000000u)) | (nx.meta & 0x40000100);
                ++q1;
                nx.kp = -1;
                ++r2;
            }
            int k = 0;
            const int rc = run_params(w, q0, q1, &k);
            if (rc) return rc;
            head.kp = k;
            head.tile |= (q1 + 1) << 24;  // the slice count the run publishes
            r = r2;
        }
    if (ntiles >= (1u << 24)) return fail(STE_EINVAL, "scheduled forward pass: a window has too many tiles");
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return fail(STE_ENOGPU, "no HIP device");
    // every wave of the launch must be resident at once (a wait may look at any other wave's earlier rounds): one per SIMD
    if (sc->nwaves > 4 * ncu) return fail(STE_EINVAL, "scheduled forward pass: nwaves exceeds the device's SIMD count (4 per compute unit)");
    hipStream_t s = (hipStream_t)stream;
    char* dw = (char*)sc->dev_ws;
    int rc = upload_table(dw, hw, lay.progress, lay.total - lay.progress, s, "scheduled forward pass: table upload");
    if (rc) return rc;
    ste::SchedParams sp;
    sp.kps = (const ste::KParams*)(dw + lay.kps);
    sp.items = (const ste::SchedItem*)(dw + lay.items);
    sp.nrounds = sc->nrounds;
    sp.nwaves = sc->nwaves;
    sp.progress = (int*)(dw + lay.progress);
    sp.window_done = sc->window_done;
    sp.error = sc->error;
    sp.timeout_ticks = (unsigned long long)((sc->timeout_s > 0 ? sc->timeout_s : 2.0) * 1e8);
    sp.started = sc->started;
    const dim3 grid((unsigned)sc->nwaves), block(64);
    switch (variant) {
#define STE_SCHED(n, g, f, r) \
    case n: hipLaunchKernelGGL((ste::ukf_forward_sched<g, f, r>), grid, block, 0, s, sp); break;
        STE_SCHED(0, false, false, false) STE_SCHED(1, true, false, false) STE_SCHED(2, false, true, false)
        STE_SCHED(3, true, true, false) STE_SCHED(4, false, true, true) STE_SCHED(5, true, true, true)
#undef STE_SCHED
        default: return fail(STE_EINVAL, "scheduled forward pass: no kernel variant");
    }
    return check_hip(hipGetLastError(), "ukf_forward_sched launch");
}

int ste_stream_wait_counter(const int32_t* counter, int32_t need, int32_t* error, double timeout_s, void* stream) {
    if (!counter) return fail(STE_EINVAL, "counter is NULL");
    hipLaunchKernelGGL(ste::sched_gate, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)counter, (int)need, (int*)error,
                       (unsigned long long)((timeout_s > 0 ? timeout_s : 2.0) * 1e8));
    return check_hip(hipGetLastError(), "sched_gate launch");
}

size_t ste_ukf_forward_sched_progress_offset(int32_t nwindows, int32_t max_slices, int64_t ntiles_total, int32_t nrounds,
                                             int32_t nwaves) {
    if (nwindows < 0 || max_slices < 0 || ntiles_total < 0 || nrounds < 0 || nwaves < 0) return 0;
    return sched_layout((size_t)nwindows * ((size_t)max_slices * ((size_t)max_slices + 1) / 2), (size_t)nrounds * (size_t)nwaves,
                        (size_t)ntiles_total).progress;
}

size_t ste_urtss_backward_sched_workspace(int32_t nwindows, int64_t ntiles_total) {
    if (nwindows < 0 || ntiles_total < 0) return 0;
    return align16((size_t)nwindows * sizeof(ste::KParams)) + align16((size_t)ntiles_total * sizeof(ste::SmoothItem));
}

int ste_urtss_backward_sched_f64(const ste_bwd_sched_f64* sc, void* stream) {
    if (!sc) return fail(STE_EINVAL, "schedule pointer is NULL");
    if (sc->nwindows < 1 || !sc->windows) return fail(STE_EINVAL, "scheduled smoother: nwindows >= 1 and windows are required");
    if (sc->nitems < 1 || !sc->items) return fail(STE_EINVAL, "scheduled smoother: nitems >= 1 and items are required");
    if (!sc->host_ws || !sc->dev_ws || !sc->progress || !sc->error)
        return fail(STE_EINVAL, "scheduled smoother: host_ws, dev_ws, progress and error are required");
    const int step = sc->slice_steps ? sc->slice_steps : STE_SLICE_ALIGN;
    if (step < STE_SLICE_ALIGN || step % STE_SLICE_ALIGN != 0)
        return fail(STE_EINVAL, "scheduled smoother: slice_steps must be a positive multiple of STE_SLICE_ALIGN (64)");
    const size_t kp_bytes = align16((size_t)sc->nwindows * sizeof(ste::KParams));
    const size_t total = kp_bytes + align16((size_t)sc->nitems * sizeof(ste::SmoothItem));
    if (sc->ws_bytes < total) return fail(STE_EINVAL, "scheduled smoother: workspace too small (ste_urtss_backward_sched_workspace)");
    char* hw = (char*)sc->host_ws;
    ste::KParams* kps = (ste::KParams*)hw;
    std::vector<int> nslices((size_t)sc->nwindows), tile0((size_t)sc->nwindows + 1);
    size_t ntiles = 0;
    int shift = -1;
    for (int w = 0; w < sc->nwindows; ++w) {
        const ste_ukf_batch_f64& b = sc->windows[w];
        if (b.B <= 0 || b.Nmax <= 0) return fail(STE_EINVAL, "scheduled smoother: a window has B <= 0 or Nmax <= 0");
        if (b.step_begin != 0 || b.step_end != 0) return fail(STE_EINVAL, "scheduled smoother: windows must not carry a step range of their own");
        int rc = make_params(&b, true, true, kps + w);
        if (rc) return rc;
        const ste::KParams& kp = kps[w];
        // one kernel runs every window: the one-kernel smoother from the forward pass's work rows (what launch_backward picks
        // for a batch of more than kLeanSmootherMaxTracks tracks), with or without rates of its own
        if (!kp.rts_work || (kp.tuning & 0x200) || (!(kp.tuning & 0x400) && kp.B <= kLeanSmootherMaxTracks))
            return fail(STE_EINVAL, "scheduled smoother: every window must take the one-kernel smoother (rts_work, more than 4096 tracks or tuning bit 10)");
        const int sh = (kp.sog_rate_rts || kp.cog_rate_rts) ? 1 : 0;
        if (shift >= 0 && sh != shift) return fail(STE_EINVAL, "scheduled smoother: the windows must agree on smoother rates (one kernel runs them all)");
        shift = sh;
        nslices[w] = std::max(1, (b.Nmax + step - 1) / step);
        tile0[w] = (int)ntiles;
        ntiles += ((size_t)b.B + 63) / 64;
        if (ntiles > 0x7fffffff) return fail(STE_EINVAL, "scheduled smoother: too many tiles");
    }
    tile0[sc->nwindows] = (int)ntiles;
    if ((size_t)sc->nitems != ntiles) return fail(STE_EINVAL, "scheduled smoother: items must name every tile of every window once");
    ste::SmoothItem* items = (ste::SmoothItem*)(hw + kp_bytes);
    std::vector<char> seen(ntiles, 0);
    for (int i = 0; i < sc->nitems; ++i) {
        const int w = sc->items[2 * i], t = sc->items[2 * i + 1];
        if (w < 0 || w >= sc->nwindows || t < 0 || t >= tile0[w + 1] - tile0[w])
            return fail(STE_EINVAL, "scheduled smoother: an item names a window or tile that does not exist");
        if (seen[(size_t)(tile0[w] + t)]) return fail(STE_EINVAL, "scheduled smoother: a tile appears twice");
        seen[(size_t)(tile0[w] + t)] = 1;
        items[i].kp = w;
        items[i].tile = t;
        items[i].prog = tile0[w] + t;
        items[i].need = nslices[w];
    }
    hipStream_t s = (hipStream_t)stream;
    char* dw = (char*)sc->dev_ws;
    int rc = upload_table(dw, hw, total, 0, s, "scheduled smoother: table upload");
    if (rc) return rc;
    ste::SmoothSchedParams sp;
    sp.kps = (const ste::KParams*)dw;
    sp.items = (const ste::SmoothItem*)(dw + kp_bytes);
    sp.progress = sc->progress;
    sp.error = sc->error;
    sp.timeout_ticks = (unsigned long long)((sc->timeout_s > 0 ? sc->timeout_s : 2.0) * 1e8);
    if (shift)
        hipLaunchKernelGGL(ste::urtss_recur_sched<true>, dim3((unsigned)sc->nitems), dim3(64), 0, s, sp);
    else
        hipLaunchKernelGGL(ste::urtss_recur_sched<false>, dim3((unsigned)sc->nitems), dim3(64), 0, s, sp);
    return check_hip(hipGetLastError(), "urtss_recur_sched launch");
}

int ste_urtss_backward_f64(const ste_ukf_batch_f64* b, void* stream) {
    ste::KParams kp;
    int rc = make_params(b, true, true, &kp);
    if (rc) return rc;
    return launch_backward(kp, (hipStream_t)stream);
}

int ste_ukf_urtss_f64(const ste_ukf_batch_f64* b, void* stream) {
    ste::KParams kp;
    int rc = make_params(b, false, true, &kp);
    if (rc) return rc;
    if (b->step_begin != 0 || (b->step_end != 0 && b->step_end != b->Nmax))
        return fail(STE_EINVAL, "ste_ukf_urtss_f64 runs whole passes: time slices go through ste_ukf_forward_f64");
    rc = launch_forward(kp, (hipStream_t)stream);
    if (rc) return rc;
    return launch_backward(kp, (hipStream_t)stream);
}

int ste_geodetic_dynamics_f64(int64_t count, const double* x, const double* dt, const double* sog_rate,
                              const double* cog_rate, double* out, void* stream) {
    if (count < 0) return fail(STE_EINVAL, "count must be >= 0");
    if (count == 0) return STE_OK;
    if (!x || !dt || !sog_rate || !cog_rate || !out) return fail(STE_EINVAL, "NULL pointer argument");
    const unsigned grid = (unsigned)((count + 63) / 64);
    hipLaunchKernelGGL(ste::geodetic_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, (size_t)count, x, dt,
                       sog_rate, cog_rate, out);
    return check_hip(hipGetLastError(), "geodetic_dynamics launch");
}

int ste_ukf_predict_f64(int64_t count, const double* x, const double* P, const double* dt, const double* sog_rate,
                        const double* cog_rate, const double* noise, const double* Q, double fan_scale, double w0,
                        double wi, double* x_out, double* P_out, int32_t* status, void* stream) {
    if (count < 0) return fail(STE_EINVAL, "count must be >= 0");
    if (count == 0) return STE_OK;
    if (!x || !P || !dt || !sog_rate || !cog_rate || !Q || !x_out || !P_out) return fail(STE_EINVAL, "NULL pointer argument");
    ste::Mats m;
    memset(&m, 0, sizeof(m));
    m.fan_scale = fan_scale;
    m.w0 = w0;
    m.wi = wi;
    memcpy(m.Q, Q, sizeof(double) * 16);
    const unsigned grid = (unsigned)((count + 63) / 64);
    hipLaunchKernelGGL(ste::predict_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, (size_t)count, m, x, P, dt,
                       sog_rate, cog_rate, noise, x_out, P_out, status);
    return check_hip(hipGetLastError(), "ukf_predict launch");
}

int ste_ukf_update_f64(int64_t count, const double* x, const double* P, const double* z, const double* noise,
                       const double* H, const double* R, double* x_out, double* P_out, int32_t* status, void* stream) {
    if (count < 0) return fail(STE_EINVAL, "count must be >= 0");
    if (count == 0) return STE_OK;
    if (!x || !P || !z || !H || !R || !x_out || !P_out) return fail(STE_EINVAL, "NULL pointer argument");
    ste::Mats m;
    memset(&m, 0, sizeof(m));
    memcpy(m.H, H, sizeof(double) * 16);
    memcpy(m.R, R, sizeof(double) * 16);
    const unsigned grid = (unsigned)((count + 63) / 64);
    hipLaunchKernelGGL(ste::update_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, (size_t)count, m, x, P, z,
                       noise, x_out, P_out, status);
    return check_hip(hipGetLastError(), "ukf_update launch");
}

int ste_ukf_robust_terms_f64(int64_t count, const double* x, const double* P, const double* z, const double* H,
                             const double* R, double* gamma, double* denom, void* stream) {
    if (count < 0) return fail(STE_EINVAL, "count must be >= 0");
    if (count == 0) return STE_OK;
    if (!x || !P || !z || !H || !R || !gamma || !denom) return fail(STE_EINVAL, "NULL pointer argument");
    ste::Mats m;
    memset(&m, 0, sizeof(m));
    memcpy(m.H, H, sizeof(double) * 16);
    memcpy(m.R, R, sizeof(double) * 16);
    const unsigned grid = (unsigned)((count + 63) / 64);
    hipLaunchKernelGGL(ste::robust_terms_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, (size_t)count, m, x, P, z,
                       gamma, denom);
    return check_hip(hipGetLastError(), "robust_terms launch");
}

int ste_sigma_points_f64(int64_t count, const double* x, const double* P, double scale, double* out, void* stream) {
    if (count < 0) return fail(STE_EINVAL, "count must be >= 0");
    if (count == 0) return STE_OK;
    if (!x || !P || !out) return fail(STE_EINVAL, "NULL pointer argument");
    const unsigned grid = (unsigned)((count + 63) / 64);
    hipLaunchKernelGGL(ste::sigma_points_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, (size_t)count, x, P,
                       scale, out);
    return check_hip(hipGetLastError(), "sigma_points launch");
}

int ste_sigma_points_generic_f64(int32_t n, int64_t count, const double* x, const double* P, double scale, double* out,
                                 void* stream) {
    if (n < 1 || n > ste::kMaxGenericN) return fail(STE_EINVAL, "n must be in 1..16");
    if (count < 0) return fail(STE_EINVAL, "count must be >= 0");
    if (count == 0) return STE_OK;
    if (!x || !P || !out) return fail(STE_EINVAL, "NULL pointer argument");
    const unsigned grid = (unsigned)((count + 63) / 64);
    hipLaunchKernelGGL(ste::sigma_points_generic_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, (int)n,
                       (size_t)count, x, P, scale, out);
    return check_hip(hipGetLastError(), "sigma_points_generic launch");
}

}  // extern "C"

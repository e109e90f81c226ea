// ste_gp.hip — second kernel set: Gaussian-process regression on many ship tracks at once (gfx950, fp64).
//
// The reference's GPRegression (src/track_estimators/gaussian_processes/gaussian_process.py:28-89) hands X = cumulative
// time (n x 1) and y = (lon, lat) (n x 2) to scikit-learn's GaussianProcessRegressor with the kernel
// ConstantKernel * RBF + WhiteKernel (examples/example_gaussian_process_batch.py:41).  Every objective evaluation of its
// L-BFGS-B fit is: build K(X,X), Cholesky, alpha = K^-1 y, log-marginal likelihood, K^-1, gradient (GPML Alg. 2.1,
// eq. 5.9).  This file evaluates that objective for a batch of tracks with hand-written kernels:
//
//   gp_kbuild      K = c exp(-(xi-xj)^2 / 2 l^2) + (s + jitter) I, lower 64x64 tiles, identity padding
//   gp_potrf_cols  blocked left-looking Cholesky in column order, one workgroup per matrix, four tiles per pass
//   gp_trtri_cols  U = L^-T (upper, row-major) in column order (large batches); gp_trtri_rows: one workgroup per
//                  (matrix, block row) for small batches
//   gp_kinv_trace  K^-1 = U U^T tile by tile, reduced on the fly against dK/dtheta (never materialised), and
//                  alpha^T (dK/dtheta) alpha with the same Krbf values -> gradient
//   alpha = U (U^T y): in the column-ordered kernels the right-hand sides ride along (w = L^-1 y by forward substitution
//                  in gp_potrf_cols, alpha = U w tile by tile in gp_trtri_cols); gp_w / gp_alpha for the row-ordered inverse
//   gp_alpha / gp_finish   per-block shares of y.alpha, log det, alpha.alpha -> log-marginal likelihood, gradient assembly
//   gp_kstar / gp_predict   posterior mean and variance at new times (variance as a tile GEMM against K^-1)
//
// All dense work is 64x64-tile "NT" products C += A_rows * B_rows^T with both operands row-major and contiguous along
// the contraction index, issued as v_mfma_f64_16x16x4_f64: lane l feeds A[row l&15][k = l>>4] and B[col l&15][k = l>>4]
// and each lane loads four consecutive k (32 B), so a wave-level load covers 16 rows x one full 128-B line.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "../../include/ste.h"

namespace stegp {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int T = 64;  // tile edge
constexpr int kMaxOut = 4;  // ste_gp_batch_f64.nout <= 4
constexpr double kLog2Pi = 1.8378770664093453;

struct GpParams {
    int B, nmax, nb_max, nout;
    int ld;  // leading dimension of every n x n buffer = nb_max * 64
    const int32_t* n;
    const double* x;      // [B][nmax]
    const double* y;      // [B][nout][nmax]
    const double* theta;  // [B][3]
    double jitter;
    double* K;      // [B][ld][ld]  K, then L (lower)
    double* U;      // [B][ld][ld]  L^-T (upper); lower triangle reused for K^-1 when requested
    double* Dinv;   // [B][nb_max][64][64] inverses of the diagonal blocks of L
    double* alpha;  // [B][nout][nmax]
    double* lml;    // [B]
    double* grad;   // [B][3] or null
    double* tr;     // [B][3][ntiles_max] per-tile partial sums of tr(K^-1 Krbf), tr(K^-1 (Krbf o D2)), tr(K^-1)
    int32_t* status;
    int store_kinv;
    int inverse_cols;  // U = L^-T by gp_trtri_cols (else gp_trtri_rows)
    // Subset launches (ste_gp_lml_subset_f64): the grid covers `nslots` matrices and slot s works on matrix active[s];
    // without a list, slot s is matrix s and nslots == B.
    int nslots;
    const int32_t* active;
};

__device__ __forceinline__ int matrix_of(const GpParams& p, int slot) { return p.active ? p.active[slot] : slot; }

__device__ __forceinline__ int nblocks(int n) { return (n + T - 1) / T; }

// One wave accumulates a 32x32 block (2x2 MFMA tiles) of C += A_rows[ra..ra+32) * B_rows[rb..rb+32)^T over k in [k0, k1).
// acc[mi][ni] holds rows ra + 16 mi + (lane>>4) + 4 reg, col rb + 16 ni + (lane & 15).
__device__ __forceinline__ void wave_gemm_nt(v4d (&acc)[2][2], const double* __restrict__ A, size_t lda,
                                             const double* __restrict__ Bm, size_t ldb, int k0, int k1, int lane) {
    const int r = lane & 15, g = lane >> 4;
    const double* pa0 = A + (size_t)r * lda + 4 * g;
    const double* pa1 = pa0 + 16 * lda;
    const double* pb0 = Bm + (size_t)r * ldb + 4 * g;
    const double* pb1 = pb0 + 16 * ldb;
    if (k0 >= k1) return;
    // Software pipeline without register shuffling: two named fragment sets ping-pong, each reloaded (for the chunk 32
    // further on) right after its 16 MFMAs were issued, so one chunk of loads is always in flight behind 1024 cycles of
    // matrix work.  Every k-range in this file is a multiple of 64 long.
    auto mma = [&](const v4d& a0, const v4d& a1, const v4d& b0, const v4d& b1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], b0[e], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], b1[e], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[e], b0[e], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[e], b1[e], acc[1][1], 0, 0, 0);
        }
    };
    v4d pa_0 = *reinterpret_cast<const v4d*>(pa0 + k0), pa_1 = *reinterpret_cast<const v4d*>(pa1 + k0);
    v4d pb_0 = *reinterpret_cast<const v4d*>(pb0 + k0), pb_1 = *reinterpret_cast<const v4d*>(pb1 + k0);
    for (int k = k0; k < k1; k += 32) {
        const v4d qa_0 = *reinterpret_cast<const v4d*>(pa0 + k + 16), qa_1 = *reinterpret_cast<const v4d*>(pa1 + k + 16);
        const v4d qb_0 = *reinterpret_cast<const v4d*>(pb0 + k + 16), qb_1 = *reinterpret_cast<const v4d*>(pb1 + k + 16);
        mma(pa_0, pa_1, pb_0, pb_1);
        if (k + 32 < k1) {
            pa_0 = *reinterpret_cast<const v4d*>(pa0 + k + 32);
            pa_1 = *reinterpret_cast<const v4d*>(pa1 + k + 32);
            pb_0 = *reinterpret_cast<const v4d*>(pb0 + k + 32);
            pb_1 = *reinterpret_cast<const v4d*>(pb1 + k + 32);
        }
        mma(qa_0, qa_1, qb_0, qb_1);
    }
}

__device__ __forceinline__ void zero_acc(v4d (&acc)[2][2]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
}

// Visit the 16 elements a lane owns in its wave's 32x32 block: f(row, col, value&), rows/cols relative to the 64x64 tile.
template <typename F>
__device__ __forceinline__ void for_each_acc(v4d (&acc)[2][2], int wave, int lane, F f) {
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) f(wr + 16 * m + (lane >> 4) + 4 * e, wc + 16 * n + (lane & 15), acc[m][n][e]);
}

// ---------------------------------------------------------------------------------------------------------------
// K build: lower tiles of K (including the diagonal tiles in full), identity on the padding.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_kbuild(const GpParams p) {
    const int b = matrix_of(p, blockIdx.y);
    const int n = p.n[b];
    const int nb = nblocks(n);
    // tile index -> (ti >= tj)
    int tile = blockIdx.x;
    int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    while (ti * (ti + 1) / 2 > tile) --ti;
    const int tj = tile - ti * (ti + 1) / 2;
    if (ti >= nb) return;
    const double c = exp(p.theta[b * 3 + 0]), inv_l = exp(-p.theta[b * 3 + 1]), s = exp(p.theta[b * 3 + 2]);
    const double* x = p.x + (size_t)b * p.nmax;
    double* K = p.K + (size_t)b * p.ld * p.ld;
    for (int e = threadIdx.x; e < T * T; e += 256) {
        const int r = ti * T + e / T, cidx = tj * T + e % T;
        double v;
        if (r < n && cidx < n) {
            const double d = (x[r] - x[cidx]) * inv_l;
            v = c * exp(-0.5 * d * d);
            if (r == cidx) v += s + p.jitter;
        } else {
            v = (r == cidx) ? 1.0 : 0.0;
        }
        K[(size_t)r * p.ld + cidx] = v;
    }
}

constexpr int LD = T + 1;  // leading dimension of the 64 x 64 diagonal-block buffers in LDS (spreads banks)

// ---------------------------------------------------------------------------------------------------------------
// U = L^-T (upper triangular, row-major).  Row block a of U depends only on L and Dinv: one workgroup per (a, matrix).
//   U[a][a] = Dinv_a^T ;  U[a][b] = -( sum_{k in [a, b)} U[a][k] L[b][k]^T ) Dinv_b^T   for b > a
// ---------------------------------------------------------------------------------------------------------------
// Row-ordered variant for small batches: one workgroup per (block row, matrix), grid = nb x B, so that a handful of
// matrices still fills the chip; large batches use gp_trtri_cols below.
__global__ __launch_bounds__(256) void gp_trtri_rows(const GpParams p) {
    __shared__ double S[T * LD];
    const int b = matrix_of(p, blockIdx.y);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = nblocks(p.n[b]);
    const int a_begin = blockIdx.x, a_end = blockIdx.x + 1;
    if (a_begin >= nb) return;
    const size_t ld = p.ld;
    for (int a = a_begin; a < a_end; ++a) {
    const double* L = p.K + (size_t)b * ld * ld;
    double* U = p.U + (size_t)b * ld * ld;
    const double* Dinv = p.Dinv + (size_t)b * p.nb_max * T * T;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    for (int e = tid; e < T * T; e += 256) {
        const int r = e / T, c = e % T;
        U[(size_t)(a * T + r) * ld + a * T + c] = Dinv[(size_t)a * T * T + (size_t)c * T + r];
        if (a & 1) U[(size_t)(a * T + r) * ld + (a - 1) * T + c] = 0.0;  // gp_kinv_trace starts at the 128-aligned column
    }
    __threadfence_block();
    __syncthreads();
    for (int bb = a + 1; bb < nb; ++bb) {
        v4d acc[2][2];
        zero_acc(acc);
        wave_gemm_nt(acc, U + (size_t)(a * T + wr) * ld, ld, L + (size_t)(bb * T + wc) * ld, ld, a * T, bb * T, lane);
        for_each_acc(acc, wave, lane, [&](int r, int c, double v) { S[r * LD + c] = -v; });
        __syncthreads();
        v4d a2[2][2];
        zero_acc(a2);
        const int r = lane & 15, g = lane >> 4;
        const double* Db = Dinv + (size_t)bb * T * T;
        for (int k = 0; k < T; k += 16) {
            double a0[4], a1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] = S[(wr + r) * LD + k + 4 * g + e];
                a1[e] = S[(wr + 16 + r) * LD + k + 4 * g + e];
            }
            const v4d b0 = *reinterpret_cast<const v4d*>(Db + (size_t)(wc + r) * T + k + 4 * g);
            const v4d b1 = *reinterpret_cast<const v4d*>(Db + (size_t)(wc + 16 + r) * T + k + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a2[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], b0[e], a2[0][0], 0, 0, 0);
                a2[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], b1[e], a2[0][1], 0, 0, 0);
                a2[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[e], b0[e], a2[1][0], 0, 0, 0);
                a2[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[e], b1[e], a2[1][1], 0, 0, 0);
            }
        }
        for_each_acc(a2, wave, lane, [&](int rr, int cc, double v) {
            U[(size_t)(a * T + rr) * ld + bb * T + cc] = v;
        });
        __threadfence_block();
        __syncthreads();
    }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Column-ordered panel kernels (large batches).  Measured at 1000 x 2000: the row-ordered kernels above re-read both
// 64 x 64 operands of every tile product from HBM -- 394 GB per gp_potrf launch, 4.2 TB/s, HBM-bound
// (profiles/r01_gp_pmc_hbm_traffic_1000x2000.csv).  In column order the tiles of one block column are independent and
// share one operand (row panel j of L), so a workgroup takes four of them at a time, one 64 x 64 tile per wave:
//   * the shared panel goes through LDS once per workgroup (double-buffered 64 x 64 blocks, one barrier per block),
//   * each wave streams its own rows from global memory into four rotating fragment sets (refilled right after use:
//     three sub-blocks = ~6000 MFMA cycles of prefetch distance at one wave per SIMD),
//   * the product is accumulated TRANSPOSED (shared rows index the accumulator rows), because a 16x16x4 accumulator --
//     lane l holds rows (l>>4) + 4 reg of column l&15 -- is exactly the B operand of the next MFMA: the multiplication
//     by the inverted diagonal block, Dinv * S^T, consumes the accumulators in place with no LDS or memory round trip.
// Bytes per tile product: 32 KB own rows + 32/4 KB shared = 40 KB instead of 64 KB.
// ---------------------------------------------------------------------------------------------------------------
constexpr int LDB = T + 2;  // LDS row stride of a staged 64 x 64 block: 528 B keeps 16-byte alignment and spreads banks

struct RowFrag {
    v4d v[4];  // rows r + 16 n (n = 0..3) of a 64-row operand, 4 consecutive k per lane
};
__device__ __forceinline__ void load_rows(RowFrag& f, const double* const (&pr)[4], int k) {
#pragma unroll
    for (int n = 0; n < 4; ++n) f.v[n] = *reinterpret_cast<const v4d*>(pr[n] + k);
}
// only the first `nl` strips (wave-uniform): the strips of a pass's triangle that have not joined yet are structurally zero
// or not yet written -- gp_trtri_cols moves 4.97 TB/s, and a sixth of its own-row fetches were of such strips
__device__ __forceinline__ void load_rows_live(RowFrag& f, const double* const (&pr)[4], int k, int nl) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
        if (n < nl) f.v[n] = *reinterpret_cast<const v4d*>(pr[n] + k);
}
__device__ __forceinline__ void stage_load(v4d (&st)[4], const double* src, size_t ld, int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        st[q] = *reinterpret_cast<const v4d*>(src + (size_t)(e >> 4) * ld + (e & 15) * 4);
    }
}
__device__ __forceinline__ void stage_store(double* dst, const v4d (&st)[4], int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        *reinterpret_cast<v4d*>(dst + (e >> 4) * LDB + (e & 15) * 4) = -st[q];  // the panel is staged NEGATED
    }
}
// ---- the shared panel by LDS-DMA (gfx950 global_load_lds_dwordx4), -DSTE_GP_GLDS=1 -------------------------------------
// One wave-instruction moves 64 x 16 B from per-lane global addresses to ONE contiguous KB of LDS (wave-uniform base in M0 +
// 16 B x lane), so the staged block is kept in FRAGMENT order: chunk (sub, m, h) holds, for lane (r, g), the two doubles
// k = 16 sub + 4 g + 2 h + {0, 1} of panel row 16 m + r -- exactly what that lane feeds MFMAs e = 2 h, 2 h + 1 of block row m.
// Fragment reads are then two conflict-free ds_read_b128 of consecutive lanes.  No staging registers (32 per lane), no
// ds_write, no negation pass: the MFMAs negate their A operand (neg:[1,0,0]).  Wave w fills the chunks of block row m = w
// (its own 16 panel rows, whole 128-byte lines between its two halves).
#ifndef STE_GP_GLDS
#define STE_GP_GLDS 0
#endif
constexpr bool kGlds = STE_GP_GLDS != 0;
constexpr int kChunk = 128;  // doubles per chunk (1 KB)
__device__ __forceinline__ int chunk_of(int sub, int m, int h) { return ((sub * 4 + m) * 2 + h) * kChunk; }
__device__ __forceinline__ void glds_panel_block(double* dst, const double* src_lane, int wave) {
    // src_lane = panel + (16 wave + r) * ld + 4 g + k0 (this lane's row of its wave's block row, at the block's first k)
    typedef __attribute__((address_space(3))) void* lds_ptr;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            __builtin_amdgcn_global_load_lds(src_lane + 16 * sub + 2 * h, (lds_ptr)(dst + chunk_of(sub, wave, h)), 16, 0, 0);
}
__device__ __forceinline__ v4d frag_glds(const double* blk, int sub, int m, int lane) {
    const v2d lo = *reinterpret_cast<const v2d*>(blk + chunk_of(sub, m, 0) + 2 * lane);
    const v2d hi = *reinterpret_cast<const v2d*>(blk + chunk_of(sub, m, 1) + 2 * lane);
    return v4d{lo[0], lo[1], hi[0], hi[1]};
}
constexpr int kNegA = kGlds ? 1 : 0;  // blgp of the f64 MFMA = neg modifiers: bit 0 negates A

// acc[m][n] += shared[16 m + ..][k] * own_n[..][k] for one 16-k sub-block; shared fragments come from LDS.  Only the first
// `nlive` of the wave's four own strips take part (4 in the steady state: one basic block of 64 MFMAs).
template <bool kFull>
__device__ __forceinline__ void mma_sub(v4d (&acc)[4][4], const double* blk, int sub, const RowFrag& own, int r, int g,
                                        int nlive, int mcap) {
    v4d a[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
        a[m] = kGlds ? frag_glds(blk, sub, m, r + 16 * g) : *reinterpret_cast<const v4d*>(blk + (r + 16 * m) * LDB + 16 * sub + 4 * g);
    if (kFull) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][e], own.v[n][e], acc[m][n], 0, 0, kNegA);
    } else {
        // own strips n >= nlive and shared 16-row groups m >= mcap (the zero padding of the last panel) stay out
#pragma unroll
        for (int n = 0; n < 4; ++n)
            if (n < nlive) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (m < mcap) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][e], own.v[n][e], acc[m][n], 0, 0, kNegA);
                    }
            }
    }
}
// One 16-k sub-block of the steady state with the shared fragments ROLLING: a[m] holds block row m of sub-block `sub` on
// entry; after its sixteen MFMAs it is refilled with block row m of sub-block sub + 1 (kLast: the block's last sub-block,
// nothing to fetch -- the next block's fragments lie in the other staging buffer, behind the barrier).
// The wave's own rows ride along the same way: strip m of the fragment set the PREVIOUS sub-block used (`refill`) is fetched
// for its next use (16-k offset `refill_k`) behind block row m's MFMAs -- two 16-byte loads per sixteen MFMAs instead of
// eight loads in a row between two sub-blocks, where each load's issue slot beyond the first MFMA's shadow is a matrix-pipe
// bubble.
template <bool kLast, typename Extra>
__device__ __forceinline__ void mma_sub_rolling(v4d (&acc)[4][4], const double* blk, int sub, const RowFrag& own, int r, int g,
                                                v4d (&a)[4], RowFrag& refill, const double* const (&rows)[4], int refill_k,
                                                Extra extra) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][e], own.v[n][e], acc[m][n], 0, 0, kNegA);
        // (fenced on both sides: unfenced, hipcc hoists the reads to the top of the block or sinks them to their use)
        __builtin_amdgcn_sched_barrier(0);
        if (!kLast)
            a[m] = kGlds ? frag_glds(blk, sub + 1, m, r + 16 * g)
                         : *reinterpret_cast<const v4d*>(blk + (r + 16 * m) * LDB + 16 * (sub + 1) + 4 * g);
        refill.v[m] = *reinterpret_cast<const v4d*>(rows[m] + refill_k);
        extra(m);  // (the caller's share of the same slot: a quarter of the next panel block's staging loads)
        __builtin_amdgcn_sched_barrier(0);
    }
}
// accT -= sum over 64-k blocks [kb0, kb1) of shared_rows[.][k] * own_rows[.][k]^T (both callers subtract the product, so
// the shared panel is negated once on its way into LDS).  `shared` points at row 0 / column 0
// of the shared 64-row panel.  A wave's four own strips are 16 rows of FOUR DIFFERENT tiles (own[n] = strip `wave` of
// the pass's n-th tile), so that the four waves of a pass always carry the same load: strip n joins at block wb0 + n
// (before that its operand is structurally zero -- or not yet written) and strips n >= cap belong to tiles the pass
// does not have (the last pass of a column).  A pass with t < 4 tiles costs t/4 of a full one instead of a full one.
// mcap < 4: only the first mcap 16-row groups of the shared panel are real (the last block row of a padded matrix).
// Every thread of the workgroup must call this with the same kb0, kb1 (it contains barriers).
__device__ __forceinline__ void panel_gemm_t(v4d (&acc)[4][4], const double* shared, size_t lds_ld,
                                             const double* const (&own)[4], int kb0, int kb1, int wb0, int cap, int mcap,
                                             double* stage, int tid, int lane) {
    if (kb0 >= kb1) return;
    const int r = lane & 15, g = lane >> 4;
    v4d st[4];
    const double* panel_lane = shared + (size_t)(16 * (tid >> 6) + r) * lds_ld + 4 * g;  // kGlds: this lane's panel row
    if (kGlds)
        glds_panel_block(stage, panel_lane + (size_t)kb0 * T, tid >> 6);
    else
        stage_load(st, shared + (size_t)kb0 * T, lds_ld, tid);
    // fragment sets f0, f1, f2 hold sub-blocks 0-2 of the block about to run; f3 (sub-block 3) is fetched during sub-block 0
    RowFrag f0, f1, f2, f3;
    {
        const int nl0 = min(cap, kb0 - wb0 + 1);
        load_rows_live(f0, own, kb0 * T, nl0);
        load_rows_live(f1, own, kb0 * T + 16, nl0);
        load_rows_live(f2, own, kb0 * T + 32, nl0);
    }
    if (!kGlds) stage_store(stage, st, tid);
    __syncthreads();
    auto block = [&](auto full, int kb) {
        constexpr bool kFull = decltype(full)::value;
        double* cur = stage + ((kb - kb0) & 1) * (T * LDB);
        double* nxt = stage + (((kb - kb0) & 1) ^ 1) * (T * LDB);
        const bool more = kb + 1 < kb1;
        const int nlive = min(cap, kb - wb0 + 1);
        const int kn = (more ? kb + 1 : kb0) * T;  // the refills past the end re-read the first block and are dropped
        if (kGlds)
            glds_panel_block(nxt, panel_lane + (size_t)kn, tid >> 6);  // nxt was last read in the previous block: free since its barrier
        else if (!kFull)
            stage_load(st, shared + (size_t)kn, lds_ld, tid);
        if (kFull) {
            // steady state (round 5): block row m of the shared fragment is refilled for the NEXT sub-block right behind its own
            // sixteen MFMAs, so its LDS read has the other three block rows' 48 MFMAs to land in; before, all four rows were
            // read and waited for in front of every sub-block's 64 MFMAs
            v4d a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
                a[m] = kGlds ? frag_glds(cur, 0, m, lane) : *reinterpret_cast<const v4d*>(cur + (r + 16 * m) * LDB + 4 * g);
            // the next panel block's staging loads (register path) go a quarter each behind sub-block 0's block rows
            const double* stage_src = shared + (size_t)kn;
            auto stage_piece = [&](int q) {
                if (!kGlds) {
                    const int e = tid + 256 * q;
                    st[q] = *reinterpret_cast<const v4d*>(stage_src + (size_t)(e >> 4) * lds_ld + (e & 15) * 4);
                }
            };
            // ... and are parked (negated) in the other staging buffer a quarter each behind sub-block 3's block rows: that
            // buffer was last read in the previous block, i.e. before the barrier every wave has passed since
            auto park_piece = [&](int q) {
                if (!kGlds) {
                    const int e = tid + 256 * q;
                    *reinterpret_cast<v4d*>(nxt + (e >> 4) * LDB + (e & 15) * 4) = -st[q];
                }
            };
            auto nothing = [](int) {};
            mma_sub_rolling<false>(acc, cur, 0, f0, r, g, a, f3, own, kb * T + 48, stage_piece);
            mma_sub_rolling<false>(acc, cur, 1, f1, r, g, a, f0, own, kn, nothing);
            mma_sub_rolling<false>(acc, cur, 2, f2, r, g, a, f1, own, kn + 16, nothing);
            mma_sub_rolling<true>(acc, cur, 3, f3, r, g, a, f2, own, kn + 32, park_piece);
        } else {
            const int nlive_next = min(cap, kb + 1 - wb0 + 1);  // strips live in the block these refills are for
            load_rows_live(f3, own, kb * T + 48, nlive);
            mma_sub<kFull>(acc, cur, 0, f0, r, g, nlive, mcap);
            load_rows_live(f0, own, kn, nlive_next);
            mma_sub<kFull>(acc, cur, 1, f1, r, g, nlive, mcap);
            load_rows_live(f1, own, kn + 16, nlive_next);
            mma_sub<kFull>(acc, cur, 2, f2, r, g, nlive, mcap);
            load_rows_live(f2, own, kn + 32, nlive_next);
            mma_sub<kFull>(acc, cur, 3, f3, r, g, nlive, mcap);
        }
        if (!kGlds && !kFull) stage_store(nxt, st, tid);
        __syncthreads();
    };
    // Two loops, not one with both bodies: with the partial and the full MFMA sequences in one loop the accumulators of the
    // two paths get different registers and 768 v_accvgpr_mov per block to reconcile them.  The first covers the blocks in
    // which fewer than four strips are live (the pass's triangle, or all of a short pass), the second is the steady state.
    int kb = kb0;
    for (; kb < kb1 && (mcap < 4 || min(cap, kb - wb0 + 1) < 4); ++kb) block(std::false_type{}, kb);
    for (; kb < kb1; ++kb) block(std::true_type{}, kb);
}
// The same contraction in at most 256 VGPRs, for kernels that run two workgroups per CU (two waves per SIMD): two
// rotating fragment sets instead of four, the shared fragments read one block row at a time, the staged panel held in
// two halves.  Each wave prefetches less far ahead, but the second wave on the SIMD keeps the MFMA pipe busy meanwhile.
template <bool kFull>
__device__ __forceinline__ void mma_sub_lean(v4d (&acc)[4][4], const double* blk, int sub, const RowFrag& own, int r, int g,
                                             int nlive) {
    if (kFull) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const v4d a = *reinterpret_cast<const v4d*>(blk + (r + 16 * m) * LDB + 16 * sub + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], own.v[n][e], acc[m][n], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int n = 0; n < 3; ++n)
            if (n < nlive) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const v4d a = *reinterpret_cast<const v4d*>(blk + (r + 16 * m) * LDB + 16 * sub + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], own.v[n][e], acc[m][n], 0, 0, 0);
                }
            }
    }
}
__device__ __forceinline__ void stage_half_load(v4d (&st)[2], const double* src, size_t ld, int tid, int half) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + 256 * (2 * half + q);
        st[q] = *reinterpret_cast<const v4d*>(src + (size_t)(e >> 4) * ld + (e & 15) * 4);
    }
}
__device__ __forceinline__ void stage_half_store(double* dst, const v4d (&st)[2], int tid, int half) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + 256 * (2 * half + q);
        *reinterpret_cast<v4d*>(dst + (e >> 4) * LDB + (e & 15) * 4) = -st[q];
    }
}
__device__ __forceinline__ void stage_half(double* dst, const double* src, size_t ld, int tid, int half) {
    v4d st[2];
    stage_half_load(st, src, ld, tid, half);
    stage_half_store(dst, st, tid, half);
}
// (nlive is fixed for the pass here; kFull = all four strips live: the two cases are separate loops, because one loop with
//  both bodies does not fit the 256 registers of two workgroups per CU)
// Round 5: the shared fragment of block row m is read in two 16-byte halves (k = 4g, 4g+1 | 4g+2, 4g+3) that are refilled
// in turn -- each half for the NEXT block row as soon as its eight MFMAs have been issued, while the other half's eight
// run -- so that an LDS read always has 512 matrix-pipe cycles to land in.  Before, both halves were read and waited for
// (s_waitcnt lgkmcnt(0)) in front of every group of sixteen MFMAs: ~130 exposed cycles per 1 024.  Same registers.
struct HalfFrag {
    v2d lo, hi;
};
__device__ __forceinline__ v2d frag_half(const double* blk, int sub, int m, int h, int r, int g) {
    return *reinterpret_cast<const v2d*>(blk + (r + 16 * m) * LDB + 16 * sub + 4 * g + 2 * h);
}
// one 16-k sub-block of a full pass: 64 MFMAs, the fragment halves of (sub, m + 1) -- or (nsub_next, 0) after m = 3 --
// fetched behind the halves in use
// slot(j), j = 0 .. 7: the caller's loads and stores for this sub-block, one behind every group of eight MFMAs (own-row
// refills in the first four, a quarter of the next panel block's staging in the last four) instead of in a run between two
// sub-blocks
template <bool kLast, typename Slot>
__device__ __forceinline__ void mma_sub_lean_rolling(v4d (&acc)[4][4], const double* blk, int sub, const RowFrag& own, int r,
                                                     int g, HalfFrag& a, Slot slot) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const bool more = m < 3 || !kLast;
        const int ns = m < 3 ? sub : sub + 1, nm = m < 3 ? m + 1 : 0;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.lo[e], own.v[n][e], acc[m][n], 0, 0, 0);
        // (scheduling barriers: left alone, hipcc hoists every LDS read of the block to its top -- ten fragments live at once,
        //  800 spilled registers -- and a read that is merely fenced off from above sinks to the end of its region, right in
        //  front of its use)
        __builtin_amdgcn_sched_barrier(0);
        if (more) a.lo = frag_half(blk, ns, nm, 0, r, g);
        slot(2 * m);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.hi[e], own.v[n][2 + e], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) a.hi = frag_half(blk, ns, nm, 1, r, g);
        slot(2 * m + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <bool kFull>
__device__ __forceinline__ void panel_gemm_t_lean(v4d (&acc)[4][4], const double* shared, size_t lds_ld,
                                                  const double* const (&own)[4], int kb0, int kb1, int nlive, int nsub,
                                                  double* stage, int tid, int lane) {
    if (kb0 >= kb1) return;
    const int r = lane & 15, g = lane >> 4;
    stage_half(stage, shared + (size_t)kb0 * T, lds_ld, tid, 0);
    stage_half(stage, shared + (size_t)kb0 * T, lds_ld, tid, 1);
    RowFrag f0, f1;
    load_rows(f0, own, kb0 * T);
    if (!kFull) load_rows(f1, own, kb0 * T + 16);  // (the full loop fetches sub-block 1's rows during sub-block 0)
    __syncthreads();
    for (int kb = kb0; kb < kb1; ++kb) {
        double* cur = stage + ((kb - kb0) & 1) * (T * LDB);
        double* nxt = stage + (((kb - kb0) & 1) ^ 1) * (T * LDB);
        const int kn = (kb + 1 < kb1 ? kb + 1 : kb0) * T;  // refills past the end re-read the first block and are dropped
        // (sub-blocks at or beyond nsub are the zero padding of the last 64-block: nothing to accumulate)
        // each half of the next panel block is fetched one sub-block before it is parked in LDS: fetched and parked in one go,
        // the wave sat through a whole memory latency twice per block (s_waitcnt vmcnt(0) right behind the loads)
        v4d st[2];
        if (kFull) {
            HalfFrag a;
            a.lo = frag_half(cur, 0, 0, 0, r, g);
            a.hi = frag_half(cur, 0, 0, 1, r, g);
            // Round 5: every load and store of the block sits behind a group of eight MFMAs.  During sub-block s the fragment
            // set the sub-block before it used is refilled for sub-block s + 1 (slots 0-3: two whole sub-block halves of cover);
            // the next panel block is staged in two halves, fetched in slots 4-5 of sub-blocks 0 / 2 and parked in slots 4-5 of
            // sub-blocks 1 / 3.
            const double* stage_src = shared + (size_t)kn;
            auto refill = [&](RowFrag& f, int k, int j) {
                if (j < 4) f.v[j] = *reinterpret_cast<const v4d*>(own[j] + k);
            };
            auto fetch = [&](int half, int j) {
                if (j == 4 || j == 5) {
                    const int e = tid + 256 * (2 * half + (j - 4));
                    st[j - 4] = *reinterpret_cast<const v4d*>(stage_src + (size_t)(e >> 4) * lds_ld + (e & 15) * 4);
                }
            };
            auto park = [&](int half, int j) {
                if (j == 4 || j == 5) {
                    const int e = tid + 256 * (2 * half + (j - 4));
                    *reinterpret_cast<v4d*>(nxt + (e >> 4) * LDB + (e & 15) * 4) = -st[j - 4];
                }
            };
            mma_sub_lean_rolling<false>(acc, cur, 0, f0, r, g, a, [&](int j) { refill(f1, kb * T + 16, j); fetch(0, j); });
            if (4 * kb + 1 < nsub)
                mma_sub_lean_rolling<false>(acc, cur, 1, f1, r, g, a, [&](int j) { refill(f0, kb * T + 32, j); park(0, j); });
            if (4 * kb + 2 < nsub)
                mma_sub_lean_rolling<false>(acc, cur, 2, f0, r, g, a, [&](int j) { refill(f1, kb * T + 48, j); fetch(1, j); });
            if (4 * kb + 3 < nsub)
                mma_sub_lean_rolling<true>(acc, cur, 3, f1, r, g, a, [&](int j) { refill(f0, kn, j); park(1, j); });
        } else {
            stage_half_load(st, shared + (size_t)kn, lds_ld, tid, 0);
            mma_sub_lean<kFull>(acc, cur, 0, f0, r, g, nlive);
            load_rows(f0, own, kb * T + 32);
            stage_half_store(nxt, st, tid, 0);
            stage_half_load(st, shared + (size_t)kn, lds_ld, tid, 1);
            if (4 * kb + 1 < nsub) mma_sub_lean<kFull>(acc, cur, 1, f1, r, g, nlive);
            load_rows(f1, own, kb * T + 48);
            if (4 * kb + 2 < nsub) mma_sub_lean<kFull>(acc, cur, 2, f0, r, g, nlive);
            load_rows(f0, own, kn);
            stage_half_store(nxt, st, tid, 1);
            if (4 * kb + 3 < nsub) mma_sub_lean<kFull>(acc, cur, 3, f1, r, g, nlive);
            load_rows(f1, own, kn + 16);
        }
        __syncthreads();
    }
}

// For m2 = 0..3: row[n] = sum_m sum_e Dl[16 m2 + r][16 m + 4 e + g] * in[m][n][e], handed to f(m2, row) one block row at
// a time so that only four result tiles are live (Dl: 64 x 64 in LDS, leading dimension LD).
template <typename F>
__device__ __forceinline__ void left_mul_lds(const double* Dl, const v4d (&in)[4][4], int r, int g, F f) {
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        v4d row[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) row[n] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double a = Dl[(16 * m2 + r) * LD + 16 * m + 4 * e + g];
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    row[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, in[m][n][e], row[n], 0, 0, 0);
            }
        f(m2, row);
    }
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// The right-hand sides ride along with the factorisation and the inversion (gp_potrf_cols: w = L^-1 y by forward
// substitution, gp_trtri_cols: alpha = U w), in place in the alpha buffer.  Its 64-row blocks are read, modified and
// written by different waves of the one workgroup that owns the matrix, in different columns: the accesses go to L2
// (agent scope) so that no wave sees a line its CU cached before another wave's update.
__device__ __forceinline__ double ld_l2(const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_l2(double* q, double v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// A 64-vector parked in LDS in the order the transposed accumulators meet it: element 16 m + 4 e + g at (4 m + g) * 4 + e,
// so that lane group g reads the four values of block row m as one 32-byte word.
__device__ __forceinline__ int rhs_slot(int idx) { return (((idx >> 4) * 4 + (idx & 3)) * 4 + ((idx >> 2) & 3)); }
// part[o][n] += sum_e row[n][e] * vec_o[16 m + 4 e + g]: one block row of a (tile x vector) product, tile in accumulator layout
__device__ __forceinline__ void rhs_accumulate(double (&part)[4][4], const v4d (&row)[4], const double* vec, int m, int g,
                                               int nout) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (o < nout) {
            const v4d wv = *reinterpret_cast<const v4d*>(vec + o * T + (m * 4 + g) * 4);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) part[o][n] = fma(row[n][e], wv[e], part[o][n]);
        }
}
// rhs[o][base + 64 n + r] += sign * (part[o][n] summed over the four lane groups) for the strips n0 <= n < n1 (strip n of a
// wave lies in the pass's n-th tile); rows >= nrows do not exist
__device__ __forceinline__ void rhs_apply(double* rhs, size_t pitch, const double (&part)[4][4], int base, int r, int g,
                                          int nrows, int nout, double sign, int n0, int n1) {
    // three phases -- reduce, load everything, store everything -- so that the L2 round trips of the sixteen
    // read-modify-writes overlap instead of queueing behind one s_waitcnt vmcnt(0) each
    double t[4][4], old[4][4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            double v = part[o][n];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            t[o][n] = v;
            old[o][n] = 0.0;
        }
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (o < nout) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int idx = base + T * n + r;
                if (g == 0 && n >= n0 && n < n1 && idx < nrows) old[o][n] = ld_l2(rhs + (size_t)o * pitch + idx);
            }
        }
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (o < nout) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int idx = base + T * n + r;
                if (g == 0 && n >= n0 && n < n1 && idx < nrows) st_l2(rhs + (size_t)o * pitch + idx, fma(sign, t[o][n], old[o][n]));
            }
        }
}

// Cholesky factor and inverse of one 64 x 64 diagonal block by ONE wave, rows in registers: lane r holds row r of the
// block (64 doubles).  A column step parks the scaled column in LDS and every lane reads it back with uniform addresses
// (a broadcast, two values per ds_read_b128); the substitution reads L the same way.  No barrier inside either loop
// nest -- LDS operations of one wave complete in order -- and ~2 x 2016 FMAs in all, where the 256-thread LDS version
// (column-by-column with three barriers per column, then a 64-thread substitution) took ~250 us per block.
// In:  S = the symmetric block (row-major, leading dimension LD).  Out: S = L (lower triangle, upper part zeroed),
// X = L^-1 (lower triangular, zeros above).  `work` is a 64 x LDB scratch area.  *ok is cleared on a non-positive pivot.
// (kCaller: one copy per calling kernel -- with a single call site the out-of-line function gets registers the caller is
//  not using; shared between two kernels it follows the general convention and the caller spills 400 bytes around it)
template <int kCaller>
__device__ __noinline__ void chol_trinv_wave(double* S, double* X, double* work, int lane, int* ok) {
    double a[T];
#pragma unroll
    for (int c = 0; c < T; ++c) a[c] = S[lane * LD + c];
    bool good = true;
    double* col = work + T * LDB - T;  // last 64 doubles of the scratch area, 16-byte aligned
#pragma unroll
    for (int c = 0; c < T; ++c) {
        const double piv = readlane_f64(a[c], c);
        good = good && (piv > 0.0);
        // 1 / sqrt(piv): v_rsq_f64 and one third-order correction (~2^-70 before rounding) instead of the library's sqrt and
        // a division, ~60 dependent instructions per column on the one wave everything else waits for
        const double pv = piv > 0.0 ? piv : 1.0;
        const double y0 = __builtin_amdgcn_rsq(pv);
        const double e0 = fma(-(pv * y0), y0, 1.0);
        const double rd = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
        const double l = a[c] * rd;  // lane c: sqrt(piv); lanes below: the column of L; lanes above: unused
        a[c] = l;
        col[lane] = l;
        if (!(c & 1)) a[c + 1] = fma(-l, col[c + 1], a[c + 1]);  // odd head, then 16-byte aligned pairs
#pragma unroll
        for (int cc = (c + 2) & ~1; cc < T; cc += 2) {
            const v2d s2 = *reinterpret_cast<const v2d*>(col + cc);
            a[cc] = fma(-l, s2[0], a[cc]);
            a[cc + 1] = fma(-l, s2[1], a[cc + 1]);
        }
    }
    if (!good && lane == 0) *ok = 0;
    // L row-major with 16-byte aligned rows for the broadcast reads of the substitution; S gets its copy now, so that the 64
    // registers of a[] are free for x[] below (both live at once is 256 registers of arrays: accumulator-register shuffling)
#pragma unroll
    for (int c = 0; c < T; c += 2) *reinterpret_cast<v2d*>(work + lane * LDB + c) = v2d{a[c], a[c + 1]};
#pragma unroll
    for (int c = 0; c < T; ++c) S[lane * LD + c] = (c <= lane) ? a[c] : 0.0;
    // lane c solves L x = e_c by forward substitution
    double x[T];
#pragma unroll
    for (int r = 0; r < T; ++r) {
        double acc0 = (lane == r) ? 1.0 : 0.0, acc1 = 0.0;
#pragma unroll
        for (int k = 0; k + 1 < r; k += 2) {
            const v2d l2 = *reinterpret_cast<const v2d*>(work + r * LDB + k);
            acc0 = fma(-l2[0], x[k], acc0);
            acc1 = fma(-l2[1], x[k + 1], acc1);
        }
        if (r & 1) acc0 = fma(-work[r * LDB + r - 1], x[r - 1], acc0);
        x[r] = (acc0 + acc1) / work[r * LDB + r];  // (times 1 / L[r][r] kept from the factor loop: 380 instructions fewer and
                                                   //  1 500 register moves more -- the rows' loads then crowd the registers)
    }
#pragma unroll
    for (int c = 0; c < T; ++c) X[c * LD + lane] = x[c];  // x[c] of lane `lane` = X[row c][column lane]
}

// kBuild: K is not read but evaluated where gp_kbuild would have put it (same expression, same bits), so that an
// objective evaluation neither writes nor re-reads the 16 MB of K; the stand-alone ste_gp_potrf_f64 factors what is there.
template <bool kBuild>
__global__ __launch_bounds__(256) void gp_potrf_cols(const GpParams p) {
    __shared__ double S[T * LD];
    __shared__ double X[T * LD];
    __shared__ __attribute__((aligned(16))) double stage[2 * T * LDB];
    __shared__ __attribute__((aligned(32))) double wj[kMaxOut * T];
    __shared__ int ok;
    const int b = matrix_of(p, blockIdx.x), tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nrows = p.n[b], nb = nblocks(nrows), nout = p.nout;
    const size_t ld = p.ld;
    double* K = p.K + (size_t)b * ld * ld;
    double* Dinv = p.Dinv + (size_t)b * p.nb_max * T * T;
    // forward substitution of the right-hand sides, in place in the alpha buffer: it starts as y, block j becomes
    // w_j = Dinv_j s_j when column j's diagonal block is factored, and every tile L[i][j] then takes L[i][j] w_j off block i
    double* rhs = p.alpha + (size_t)b * nout * p.nmax;
    {
        const double* yb = p.y + (size_t)b * nout * p.nmax;
        for (int o = 0; o < nout; ++o)
            for (int e = tid; e < nrows; e += 256) st_l2(rhs + (size_t)o * p.nmax + e, yb[(size_t)o * p.nmax + e]);
    }
    if (tid == 0) ok = 1;
    __threadfence_block();
    __syncthreads();
    const int r = lane & 15, g = lane >> 4;
    for (int j = 0; j < nb; ++j) {
        // Tiles of block column j, four per pass, starting at the diagonal tile; every wave takes strip `wave` (16 rows) of
        // each of them:
        //   T(i) = K[i][j] - sum_{k<j} L[i][k] L[j][k]^T;   L[j][j] = chol(T(j));   L[i][j]^T = Dinv_j * T(i)^T
        const int mcap = min(4, (nrows - j * T + 15) / 16);  // real 16-row groups of row panel j
        for (int i0 = j; i0 < nb; i0 += 4) {
            const int ntile = min(4, nb - i0);
            // the identity padding of the last tile has nothing to accumulate (its rows of L are zero left of the diagonal)
            const int cap = ntile - ((i0 + ntile == nb && (nb - 1) * T + 16 * wave >= nrows) ? 1 : 0);
            const double* own[4];
            size_t row0[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                row0[n] = (size_t)(min(i0 + n, nb - 1) * T + 16 * wave + r) * ld;  // tiles the pass lacks: clamped, never stored
                own[n] = K + row0[n] + 4 * g;
            }
            // acc[m][n][e] is element (jj = 16 m + 4 e + g, row r of strip n) of the transposed tiles; it starts as K[i][j]^T
            // (loads in flight behind the first panel blocks) and the negated panel products are accumulated onto it
            v4d acc[4][4];
            if (kBuild) {
                // (theta and x are re-read per pass: nothing of this is live across the panel loop)
                const double* xb = p.x + (size_t)b * p.nmax;
                const double kc = exp(p.theta[b * 3 + 0]), kinv_l = exp(-p.theta[b * 3 + 1]), ks = exp(p.theta[b * 3 + 2]);
                double xc[4][4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = j * T + 16 * m + 4 * e + g;
                        xc[m][e] = col < nrows ? xb[col] : 0.0;
                    }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int row = min(i0 + n, nb - 1) * T + 16 * wave + r;
                    const double xr = row < nrows ? xb[row] : 0.0;
                    if (n < ntile) {  // (uniform: the strips of tiles the pass lacks are never stored)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int col = j * T + 16 * m + 4 * e + g;
                                double v;
                                if (row < nrows && col < nrows) {
                                    const double d = (xr - xc[m][e]) * kinv_l;
                                    v = kc * exp(-0.5 * d * d);
                                    if (row == col) v += ks + p.jitter;
                                } else {
                                    v = (row == col) ? 1.0 : 0.0;
                                }
                                acc[m][n][e] = v;
                            }
                    } else {
#pragma unroll
                        for (int m = 0; m < 4; ++m) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[m][n][e] = K[row0[n] + j * T + 16 * m + 4 * e + g];
            }
            // (first live block passed as a run-time value: with a literal hipcc merges the four sub-blocks of the panel loop
            //  into one basic block and then shuffles 160 accumulator registers between AGPRs and VGPRs per iteration)
            panel_gemm_t(acc, K + (size_t)(j * T) * ld, ld, own, 0, j, (p.nb_max < 0) - 3, cap, mcap, stage, tid, lane);
            const bool first = i0 == j;
            if (first) {
                // strip 0 of every wave is its quarter of the diagonal tile
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) S[(16 * wave + r) * LD + 16 * m + 4 * e + g] = acc[m][0][e];
                __syncthreads();
                if (wave == 0) {
                    chol_trinv_wave<kBuild ? 1 : 0>(S, X, stage, lane, &ok);
                    for (int o = 0; o < nout; ++o) {
                        const int row = j * T + lane;
                        const double sj = row < nrows ? ld_l2(rhs + (size_t)o * p.nmax + row) : 0.0;
                        double w = 0.0;
#pragma unroll
                        for (int k = 0; k < T; ++k) w = fma(X[lane * LD + k], readlane_f64(sj, k), w);
                        wj[o * T + rhs_slot(lane)] = w;
                        if (row < nrows) st_l2(rhs + (size_t)o * p.nmax + row, w);
                    }
                }
                __syncthreads();
                for (int e = tid; e < T * T; e += 256) {
                    const int rr = e / T, cc = e % T;
                    if (cc <= rr) K[(size_t)(j * T + rr) * ld + j * T + cc] = S[rr * LD + cc];
                    Dinv[(size_t)j * T * T + e] = X[rr * LD + cc];
                }
            }
            const int n0 = first ? 1 : 0;
            if (n0 < ntile) {
                double part[4][4] = {};
                left_mul_lds(X, acc, r, g, [&](int m, const v4d (&row)[4]) {
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        if (n >= n0 && n < ntile) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) K[row0[n] + j * T + 16 * m + 4 * e + g] = row[n][e];
                        }
                    rhs_accumulate(part, row, wj, m, g, nout);
                });
                rhs_apply(rhs, p.nmax, part, i0 * T + 16 * wave, r, g, nrows, nout, -1.0, n0, ntile);
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) p.status[b] = ok ? 0 : 1;
}

// U = L^-T in column order, one workgroup per matrix:  U[a][c]^T = Dinv_c * (-sum_{k in [a, c)} L[c][k] U[a][k]^T)
__global__ __launch_bounds__(256) void gp_trtri_cols(const GpParams p) {
    __shared__ double X[T * LD];
    __shared__ __attribute__((aligned(16))) double stage[2 * T * LDB];
    const int b = matrix_of(p, blockIdx.x), tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = nblocks(p.n[b]);
    const size_t ld = p.ld;
    const double* L = p.K + (size_t)b * ld * ld;
    double* U = p.U + (size_t)b * ld * ld;
    const double* Dinv = p.Dinv + (size_t)b * p.nb_max * T * T;
    const int r = lane & 15, g = lane >> 4;
    // alpha = U w in place in the alpha buffer, which gp_potrf_cols left holding w = L^-1 y: in column c block c becomes
    // Dinv_c^T w_c and every tile U[a][c] adds U[a][c] w_c to block a
    __shared__ __attribute__((aligned(32))) double wc[kMaxOut * T];
    const int nrows = p.n[b], nout = p.nout;
    double* rhs = p.alpha + (size_t)b * nout * p.nmax;
    for (int c = 0; c < nb; ++c) {
        for (int e = tid; e < T * T; e += 256) {
            const int rr = e / T, cc = e % T;
            const double d = Dinv[(size_t)c * T * T + e];
            X[rr * LD + cc] = d;
            U[(size_t)(c * T + cc) * ld + c * T + rr] = d;                     // U[c][c] = Dinv_c^T
            if (c & 1) U[(size_t)(c * T + rr) * ld + (c - 1) * T + cc] = 0.0;  // gp_kinv_trace starts 128-aligned
        }
        if (tid < T)
            for (int o = 0; o < nout; ++o)
                wc[o * T + rhs_slot(tid)] = c * T + tid < nrows ? ld_l2(rhs + (size_t)o * p.nmax + c * T + tid) : 0.0;
        __syncthreads();
        if (tid < T && c * T + tid < nrows)
            for (int o = 0; o < nout; ++o) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < T; ++k) t = fma(X[k * LD + tid], wc[o * T + rhs_slot(k)], t);
                st_l2(rhs + (size_t)o * p.nmax + c * T + tid, t);
            }
        const int mcap = min(4, (nrows - c * T + 15) / 16);  // real 16-row groups of row panel c of L (the rest: identity padding)
        for (int a0 = 0; a0 < c; a0 += 4) {
            // tiles a0 .. a0+3 of column c, strip `wave` of each: tile a0+n joins at block a0+n (U[a][k] = 0 for k < a, and
            // that part of U is not even written), so the four waves carry the same load through the pass's triangle
            const int ntile = min(4, c - a0);
            const double* own[4];
            size_t row0[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                row0[n] = (size_t)(min(a0 + n, c - 1) * T + 16 * wave + r) * ld;
                own[n] = U + row0[n] + 4 * g;
            }
            v4d acc[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = v4d{0.0, 0.0, 0.0, 0.0};
            panel_gemm_t(acc, L + (size_t)(c * T) * ld, ld, own, a0, c, a0, ntile, mcap, stage, tid, lane);
            double part[4][4] = {};
            left_mul_lds(X, acc, r, g, [&](int m, const v4d (&row)[4]) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (n < ntile) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) U[row0[n] + c * T + 16 * m + 4 * e + g] = row[n][e];
                    }
                rhs_accumulate(part, row, wc, m, g, nout);
            });
            rhs_apply(rhs, p.nmax, part, a0 * T + 16 * wave, r, g, nrows, nout, 1.0, 0, ntile);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K^-1 tile (ta >= tb) = sum_{k >= ta} U[ta][k] U[tb][k]^T, reduced against the kernel derivatives:
//   tr[0] += sum Kinv * Krbf,  tr[1] += sum Kinv * Krbf * d^2,  tr[2] += trace(Kinv)      (off-diagonal tiles count twice)
// and optionally stored (both triangles) into `kinv_out` for the predictive variance.
// One workgroup per strip = block row ta x four block columns tb (one 64 x 64 tile per wave): all four contractions run
// over the same k-range [ta, nb) and share the rows of U[ta], which is exactly the shape of panel_gemm_t -- shared rows
// through LDS, own rows through four rotating fragment sets.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int kinv_strips(int nb) {  // strips of a matrix with nb block rows, in (ta, group) order
    int s = 0;
    for (int ta = 0; ta < nb; ++ta) s += (ta + 4) / 4;
    return s;
}

__device__ __forceinline__ double* gp_wbuf(const GpParams& p, int b) { return p.Dinv + (size_t)b * p.nb_max * T * T; }
__device__ __forceinline__ double* gp_share(const GpParams& p, int b) { return gp_wbuf(p, b) + (size_t)p.nout * p.ld; }
// per-strip partial sums of alpha^T Krbf alpha and alpha^T (Krbf o d^2) alpha, behind the per-block shares
__device__ __forceinline__ double* gp_quad_share(const GpParams& p, int b) { return gp_share(p, b) + 5 * (size_t)p.nb_max; }

__global__ __launch_bounds__(256, 2) void gp_kinv_trace(const GpParams p, double* kinv_out) {
    __shared__ double red[5][4];
    __shared__ __attribute__((aligned(16))) double stage[2 * T * LDB];
    // XCD-aware mapping.  Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MB L2), so with a
    // (strip, matrix) grid one matrix's strips would land on all eight.  Here ids that are congruent mod 8 -- the ones
    // that share an XCD -- walk the strips of the SAME matrix in order, so the U rows a strip re-reads come from that
    // XCD's L2 instead of HBM (measured: 42 GB per launch at 1000 x 2000 instead of ~400).  Placement only affects
    // speed: every tile is still computed exactly once.
    const int nstrip_grid = kinv_strips(p.nb_max);
    const int id = blockIdx.x, slot = id >> 3;
    const int bslot = (slot / nstrip_grid) * 8 + (id & 7);
    if (bslot >= p.nslots) return;
    const int b = matrix_of(p, bslot);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = p.n[b], nb = nblocks(n);
    const int strip = slot % nstrip_grid;
    int ta = 0, first = 0;
    while (ta < p.nb_max && first + (ta + 4) / 4 <= strip) {
        first += (ta + 4) / 4;
        ++ta;
    }
    if (ta >= nb) return;
    // tiles tb0 .. tb0+3 (those with tb <= ta), strip `wave` of each per wave: a strip with t < 4 tiles costs t/4 of a full one
    const int tb0 = 4 * (strip - first), ntile = min(4, ta - tb0 + 1);
    const size_t ld = p.ld;
    const double* U = p.U + (size_t)b * ld * ld;
    const int r = lane & 15, g = lane >> 4, nrows = nb * T;
    const double* own[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) own[nn] = U + (size_t)(min(tb0 + nn, ta) * T + 16 * wave + r) * ld + 4 * g;
    v4d acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) acc[m][nn] = v4d{0.0, 0.0, 0.0, 0.0};
    // contraction columns >= n are padding: rows < n of U are zero there (the padded rows are only wanted when K^-1 is stored)
    const int nsub = kinv_out ? 4 * nb : (n + 15) / 16;
    if (ntile == 4)
        panel_gemm_t_lean<true>(acc, U + (size_t)(ta * T) * ld, ld, own, ta, nb, 4, nsub, stage, tid, lane);
    else
        panel_gemm_t_lean<false>(acc, U + (size_t)(ta * T) * ld, ld, own, ta, nb, ntile, nsub, stage, tid, lane);
    // acc[m][nn][e] = -Kinv[gr][gc], gr = ta*64 + 16 m + 4 e + g (shared rows), gc = (tb0+nn)*64 + 16 wave + r (own rows)
    const double c = exp(p.theta[b * 3 + 0]), inv_l = exp(-p.theta[b * 3 + 1]);
    const double* x = p.x + (size_t)b * p.nmax;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, q0 = 0.0, q1 = 0.0;
    {
        // The quadratic forms alpha^T (dK/dtheta) alpha ride along: every pair (gr, gc) meets its Krbf and d^2 here anyway
        // (alpha is complete before this kernel is launched; summed over the outputs, like the traces).
        const double* al = p.alpha + (size_t)b * p.nout * p.nmax;
        const bool quad = p.grad != nullptr;
        double ac[kMaxOut][4];
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o)
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) {
                const int gc = (tb0 + nn) * T + 16 * wave + r;
                ac[o][nn] = (quad && o < p.nout && nn < ntile && gc < n) ? al[(size_t)o * p.nmax + gc] : 0.0;
            }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gr = ta * T + 16 * m + 4 * e + g;
                double ar[kMaxOut];
#pragma unroll
                for (int o = 0; o < kMaxOut; ++o) ar[o] = (quad && o < p.nout && gr < n) ? al[(size_t)o * p.nmax + gr] : 0.0;
                const double xr = gr < n ? x[gr] : 0.0;
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    if (nn >= ntile) continue;
                    const int gc = (tb0 + nn) * T + 16 * wave + r;
                    const double wgt = (tb0 + nn == ta) ? 1.0 : 2.0;
                    const double v = -acc[m][nn][e];
                    if (gr < n && gc < n) {
                        const double d = (xr - x[gc]) * inv_l;
                        const double d2 = d * d;
                        const double kr = c * exp(-0.5 * d2);
                        t0 += wgt * v * kr;
                        t1 += wgt * v * kr * d2;
                        if (gr == gc) t2 += v;
                        double aa = ar[0] * ac[0][nn];
#pragma unroll
                        for (int o = 1; o < kMaxOut; ++o) aa = fma(ar[o], ac[o][nn], aa);
                        const double qk = wgt * kr * aa;
                        q0 += qk;
                        q1 += qk * d2;
                    }
                    if (kinv_out && gr < nrows && gc < nrows) {
                        kinv_out[(size_t)b * ld * ld + (size_t)gr * ld + gc] = v;
                        kinv_out[(size_t)b * ld * ld + (size_t)gc * ld + gr] = v;
                    }
                }
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        t0 += __shfl_xor(t0, off);
        t1 += __shfl_xor(t1, off);
        t2 += __shfl_xor(t2, off);
        q0 += __shfl_xor(q0, off);
        q1 += __shfl_xor(q1, off);
    }
    if (lane == 0) {
        red[0][wave] = t0;
        red[1][wave] = t1;
        red[2][wave] = t2;
        red[3][wave] = q0;
        red[4][wave] = q1;
    }
    __syncthreads();
    if (tid < 5) {
        // one slot per strip, summed in strip order by gp_finish: bitwise reproducible, unlike an atomic accumulation
        const int ntiles = p.nb_max * (p.nb_max + 1) / 2;
        const double v = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
        if (tid < 3)
            p.tr[((size_t)b * 3 + tid) * ntiles + strip] = v;
        else
            gp_quad_share(p, b)[(size_t)(tid - 3) * nstrip_grid + strip] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// alpha = U (U^T y) for the row-ordered inverse, lml, gradient assembly: small kernels, nb workgroups per matrix.
//   gp_w      w = U^T y for one 64-column block        (thread per column, 8 rows in flight per thread; row order only)
//   gp_alpha  alpha for one 64-row block (row order; the column-ordered kernels have left it in place) + that block's
//             share of y.alpha, log-det, alpha.alpha
//   gp_finish sums the per-block and per-strip shares in order (deterministic) -> lml, gradient
// w lives in the Dinv buffer (free once gp_trtri has run); per-block shares go to the tail of the same buffer.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gp_w(const GpParams p) {
    const int b = matrix_of(p, blockIdx.y), kb = blockIdx.x, lane = threadIdx.x;
    const int n = p.n[b], nb = nblocks(n);
    if (kb >= nb) return;
    const size_t ld = p.ld;
    const double* U = p.U + (size_t)b * ld * ld;
    const int k = kb * T + lane;
    const int amax = (k < n ? k : n - 1);
    const double* y = p.y + (size_t)b * p.nout * p.nmax;
    // one pass over the column of U for all outputs (U is the 16 MB operand, y a few KB)
    double acc[kMaxOut][4];
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[o][u] = 0.0;
    int a = 0;
    for (; a + 4 <= amax + 1; a += 4) {
        double uv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) uv[u] = U[(size_t)(a + u) * ld + k];
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o)
            if (o < p.nout) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[o][u] = fma(uv[u], y[(size_t)o * p.nmax + a + u], acc[o][u]);
            }
    }
    for (; a <= amax; ++a) {
        const double uv = U[(size_t)a * ld + k];
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o)
            if (o < p.nout) acc[o][0] = fma(uv, y[(size_t)o * p.nmax + a], acc[o][0]);
    }
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o)
        if (o < p.nout) gp_wbuf(p, b)[(size_t)o * ld + k] = (acc[o][0] + acc[o][1]) + (acc[o][2] + acc[o][3]);
}

__global__ __launch_bounds__(256) void gp_alpha(const GpParams p) {
    __shared__ double red[5][4];
    const int b = matrix_of(p, blockIdx.y), ab = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = p.n[b], nb = nblocks(n), npad = nb * T;
    if (ab >= nb) return;
    const size_t ld = p.ld;
    const double* U = p.U + (size_t)b * ld * ld;
    const double* L = p.K + (size_t)b * ld * ld;
    const double* x = p.x + (size_t)b * p.nmax;
    const double c = exp(p.theta[b * 3 + 0]), inv_l = exp(-p.theta[b * 3 + 1]);
    double yta = 0.0, sl = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0;
    const double* yb = p.y + (size_t)b * p.nout * p.nmax;
    const double* wb = gp_wbuf(p, b);
    double* alb = p.alpha + (size_t)b * p.nout * p.nmax;
    if (p.inverse_cols) {
        // gp_trtri_cols left alpha in place: only the shares remain
        const int a = ab * T + tid;
        if (tid < T && a < n) {
            for (int o = 0; o < p.nout; ++o) {
                const double t = alb[(size_t)o * p.nmax + a];
                yta += yb[(size_t)o * p.nmax + a] * t;
                q2 += t * t;
            }
            sl += log(L[(size_t)a * ld + a]);
        }
    }
    for (int r = wave; r < T && !p.inverse_cols; r += 4) {
        const int a = ab * T + r;
        if (a >= n) continue;
        double acc[kMaxOut] = {0.0, 0.0, 0.0, 0.0};
        const double* urow = U + (size_t)a * ld;  // the row of U is read once for all outputs, four loads in flight
        int k = a + lane;
        for (; k + 192 < npad; k += 256) {
            double u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = urow[k + 64 * q];
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o)
                if (o < p.nout) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[o] = fma(u[q], wb[(size_t)o * ld + k + 64 * q], acc[o]);
                }
        }
        for (; k < npad; k += 64) {
            const double u = urow[k];
#pragma unroll
            for (int o = 0; o < kMaxOut; ++o)
                if (o < p.nout) acc[o] = fma(u, wb[(size_t)o * ld + k], acc[o]);
        }
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o >= p.nout) continue;
            double t = acc[o];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if (lane == 0) {
                alb[(size_t)o * p.nmax + a] = t;
                yta += yb[(size_t)o * p.nmax + a] * t;
                q2 += t * t;
            }
        }
        if (lane == 0) sl += log(L[(size_t)a * ld + a]);
    }
    // (alpha^T (dK/dtheta) alpha needs the whole alpha vector: gp_kinv_trace, launched next, reduces it with the traces)
    double vals[5] = {yta, sl, q0, q1, q2};
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        double t = vals[v];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
        if (lane == 0) red[v][wave] = t;
    }
    __syncthreads();
    if (tid < 5) gp_share(p, b)[(size_t)tid * p.nb_max + ab] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    (void)x; (void)c; (void)inv_l;
}

__global__ __launch_bounds__(64) void gp_finish(const GpParams p) {
    const int bslot = blockIdx.x * 64 + threadIdx.x;
    if (bslot >= p.nslots) return;
    const int b = matrix_of(p, bslot);
    const int n = p.n[b], nb = nblocks(n), nout = p.nout;
    const double* sh = gp_share(p, b);
    double r[5] = {0, 0, 0, 0, 0};
    for (int v = 0; v < 5; ++v)
        for (int i = 0; i < nb; ++i) r[v] += sh[(size_t)v * p.nb_max + i];
    p.lml[b] = -0.5 * r[0] - nout * r[1] - nout * (0.5 * n) * kLog2Pi;
    if (p.grad) {
        const int ntiles = p.nb_max * (p.nb_max + 1) / 2, mine = kinv_strips(nb), nstrip_grid = kinv_strips(p.nb_max);
        double tr[3];
        for (int v = 0; v < 3; ++v) {
            double acc = 0.0;
            for (int i = 0; i < mine; ++i) acc += p.tr[((size_t)b * 3 + v) * ntiles + i];
            tr[v] = acc;
        }
        for (int v = 0; v < 2; ++v) {  // the quadratic forms, one share per strip of gp_kinv_trace
            double acc = 0.0;
            for (int i = 0; i < mine; ++i) acc += gp_quad_share(p, b)[(size_t)v * nstrip_grid + i];
            r[2 + v] = acc;
        }
        const double s = exp(p.theta[b * 3 + 2]);
        p.grad[b * 3 + 0] = 0.5 * (r[2] - nout * tr[0]);
        p.grad[b * 3 + 1] = 0.5 * (r[3] - nout * tr[1]);
        p.grad[b * 3 + 2] = 0.5 * s * (r[4] - nout * tr[2]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Prediction: Kstar[m][i] = c exp(-(xs_m - x_i)^2 / 2 l^2) (padded to tiles with zeros), mean = Kstar alpha,
// var_m = c + s - kstar_m^T K^-1 kstar_m as tile products W = Kstar K^-1 reduced against Kstar on the fly.
// ---------------------------------------------------------------------------------------------------------------
struct GpPredict {
    int B, mmax, mb_max, ldk;  // ldk = leading dim of Kstar rows (= ld of the n x n buffers)
    const int32_t* m;
    const double* xs;  // [B][mmax]
    double* Kstar;     // [B][mb_max*64][ldk]
    const double* Kinv;
    double* mean;  // [B][nout][mmax]
    double* var;   // [B][mmax]
};

__global__ __launch_bounds__(256) void gp_kstar(const GpParams p, const GpPredict q) {
    const int b = blockIdx.y, mt = blockIdx.x;
    const int n = p.n[b], m = q.m[b];
    if (mt * T >= ((m + T - 1) / T) * T) return;
    const double c = exp(p.theta[b * 3 + 0]), inv_l = exp(-p.theta[b * 3 + 1]);
    const double* x = p.x + (size_t)b * p.nmax;
    const double* xs = q.xs + (size_t)b * q.mmax;
    double* Ks = q.Kstar + (size_t)b * q.mb_max * T * q.ldk;
    const int npad = nblocks(n) * T;
    for (int r = 0; r < T; ++r) {
        const int gm = mt * T + r;
        for (int i = threadIdx.x; i < npad; i += 256) {
            double v = 0.0;
            if (gm < m && i < n) {
                const double d = (xs[gm] - x[i]) * inv_l;
                v = c * exp(-0.5 * d * d);
            }
            Ks[(size_t)gm * q.ldk + i] = v;
        }
    }
}

__global__ __launch_bounds__(256) void gp_predict(const GpParams p, const GpPredict q) {
    __shared__ double rowsum[T][2];
    const int b = blockIdx.y, mt = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = p.n[b], m = q.m[b], nb = nblocks(n);
    if (mt * T >= m) return;
    const size_t ld = p.ld;
    const double* Ks = q.Kstar + (size_t)b * q.mb_max * T * q.ldk;
    const double* Kinv = q.Kinv + (size_t)b * ld * ld;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    double part[8];  // per-lane partial of sum_i W[m][i] Kstar[m][i] for the 8 rows this lane touches (2 m-blocks x 4 regs)
#pragma unroll
    for (int e = 0; e < 8; ++e) part[e] = 0.0;
    for (int it = 0; it < nb; ++it) {
        v4d acc[2][2];
        zero_acc(acc);
        wave_gemm_nt(acc, Ks + (size_t)(mt * T + wr) * q.ldk, q.ldk, Kinv + (size_t)(it * T + wc) * ld, ld, 0, nb * T, lane);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = wr + 16 * mi + (lane >> 4) + 4 * e, c = wc + 16 * ni + (lane & 15);
                    part[mi * 4 + e] += acc[mi][ni][e] * Ks[(size_t)(mt * T + r) * q.ldk + it * T + c];
                }
    }
    // reduce over the 16 lanes that share a row (lane & 15), then over the two waves that share the row block
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        double t = part[e];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) t += __shfl_xor(t, off);
        part[e] = t;
    }
    if ((lane & 15) == 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 4; ++e) rowsum[wr + 16 * mi + (lane >> 4) + 4 * e][wave & 1] = part[mi * 4 + e];
    }
    __syncthreads();
    const double c = exp(p.theta[b * 3 + 0]), s = exp(p.theta[b * 3 + 2]);
    if (tid < T) {
        const int gm = mt * T + tid;
        if (gm < m) q.var[(size_t)b * q.mmax + gm] = (c + s) - (rowsum[tid][0] + rowsum[tid][1]);
    }
    // mean: wave per row, lanes over i
    for (int o = 0; o < p.nout; ++o) {
        const double* al = p.alpha + ((size_t)b * p.nout + o) * p.nmax;
        for (int r = wave; r < T; r += 4) {
            const int gm = mt * T + r;
            if (gm >= m) continue;
            double acc = 0.0;
            for (int i = lane; i < n; i += 64) acc = fma(Ks[(size_t)gm * q.ldk + i], al[i], acc);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (lane == 0) q.mean[((size_t)b * p.nout + o) * q.mmax + gm] = acc;
        }
    }
}

}  // namespace stegp

// ===============================================================================================================
// C ABI
// ===============================================================================================================
namespace {
thread_local char g_gp_err[256] = "";
int gp_fail(const char* m) {
    snprintf(g_gp_err, sizeof(g_gp_err), "%s", m);
    return STE_EINVAL;
}
int gp_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return STE_OK;
    snprintf(g_gp_err, sizeof(g_gp_err), "%s: %s", what, hipGetErrorString(e));
    return STE_ELAUNCH;
}
int gp_params(const ste_gp_batch_f64* b, stegp::GpParams* p) {
    if (!b) return gp_fail("gp batch pointer is NULL");
    if (b->B <= 0 || b->nmax <= 0) return gp_fail("B and nmax must be > 0");
    if (b->nout < 1 || b->nout > 4) return gp_fail("nout must be in 1..4");
    if (!b->n || !b->x || !b->y || !b->theta || !b->K || !b->U || !b->Dinv || !b->alpha || !b->lml || !b->tr || !b->status)
        return gp_fail("n, x, y, theta, K, U, Dinv, alpha, lml, tr and status are required");
    p->B = b->B;
    p->nmax = b->nmax;
    p->nb_max = (b->nmax + 63) / 64;
    p->nout = b->nout;
    p->ld = p->nb_max * 64 + STE_GP_LD_PAD;
    p->n = b->n;
    p->x = b->x;
    p->y = b->y;
    p->theta = b->theta;
    p->jitter = b->jitter;
    p->K = b->K;
    p->U = b->U;
    p->Dinv = b->Dinv;
    p->alpha = b->alpha;
    p->lml = b->lml;
    p->grad = b->grad;
    p->tr = b->tr;
    p->status = b->status;
    p->store_kinv = 0;
    if (b->inverse_order < STE_GP_INVERSE_AUTO || b->inverse_order > STE_GP_INVERSE_COLS)
        return gp_fail("inverse_order must be STE_GP_INVERSE_AUTO, _ROWS or _COLS");
    // AUTO: chosen by the size of the BATCH, not of a launch over a subset of it
    p->inverse_cols = b->inverse_order == STE_GP_INVERSE_AUTO ? (b->B >= 128) : (b->inverse_order == STE_GP_INVERSE_COLS);
    p->nslots = b->B;
    p->active = nullptr;
    return STE_OK;
}
}  // namespace

extern "C" {

const char* ste_gp_last_error(void) { return g_gp_err; }

int ste_gp_rbf_kmatrix_f64(const ste_gp_batch_f64* b, void* stream) {
    stegp::GpParams p;
    int rc = gp_params(b, &p);
    if (rc) return rc;
    const int tiles = p.nb_max * (p.nb_max + 1) / 2;
    hipLaunchKernelGGL(stegp::gp_kbuild, dim3(tiles, p.B), dim3(256), 0, (hipStream_t)stream, p);
    return gp_hip(hipGetLastError(), "gp_kbuild launch");
}

int ste_gp_potrf_f64(const ste_gp_batch_f64* b, void* stream) {
    stegp::GpParams p;
    int rc = gp_params(b, &p);
    if (rc) return rc;
    hipLaunchKernelGGL(stegp::gp_potrf_cols<false>, dim3(p.B), dim3(256), 0, (hipStream_t)stream, p);
    return gp_hip(hipGetLastError(), "gp_potrf launch");
}

static int gp_lml_launch(const ste_gp_batch_f64* b, int32_t count, const int32_t* active, void* stream) {
    stegp::GpParams p;
    int rc = gp_params(b, &p);
    if (rc) return rc;
    if (active) {
        if (count < 0 || count > p.B) return gp_fail("count must be in 0..B");
        if (count == 0) return STE_OK;
        p.nslots = count;
        p.active = active;
    }
    const unsigned ns = (unsigned)p.nslots;
    hipStream_t s = (hipStream_t)stream;
    // (no gp_kbuild: the factorisation evaluates the kernel function where it would read K)
    hipLaunchKernelGGL(stegp::gp_potrf_cols<true>, dim3(ns), dim3(256), 0, s, p);
    // which of the two inverse kernels runs is a property of the batch (gp_params), never of this launch: a subset launch
    // must leave the bits a full launch leaves (include/ste.h: "per-matrix results do not depend on which other matrices are
    // listed"), and the two kernels sum in different orders
    if (p.inverse_cols)
        hipLaunchKernelGGL(stegp::gp_trtri_cols, dim3(ns), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(stegp::gp_trtri_rows, dim3(p.nb_max, ns), dim3(256), 0, s, p);
    // alpha first: gp_kinv_trace reduces alpha^T (dK/dtheta) alpha along with the traces.  The column-ordered kernels
    // leave alpha in place (gp_potrf_cols: w = L^-1 y, gp_trtri_cols: alpha = U w); the row-ordered inverse needs gp_w.
    if (!p.inverse_cols) hipLaunchKernelGGL(stegp::gp_w, dim3(p.nb_max, ns), dim3(64), 0, s, p);
    hipLaunchKernelGGL(stegp::gp_alpha, dim3(p.nb_max, ns), dim3(256), 0, s, p);
    if (p.grad || b->Kinv) {
        const unsigned groups = (ns + 7) / 8, strips = (unsigned)stegp::kinv_strips(p.nb_max);
        hipLaunchKernelGGL(stegp::gp_kinv_trace, dim3(groups * 8u * strips), dim3(256), 0, s, p, b->Kinv);
    }
    hipLaunchKernelGGL(stegp::gp_finish, dim3((ns + 63) / 64), dim3(64), 0, s, p);
    return gp_hip(hipGetLastError(), "gp_lml launch");
}

int ste_gp_lml_f64(const ste_gp_batch_f64* b, void* stream) { return gp_lml_launch(b, 0, nullptr, stream); }

int ste_gp_lml_subset_f64(const ste_gp_batch_f64* b, int32_t count, const int32_t* active, void* stream) {
    if (!active) return gp_fail("active (device int32[count]) is required; use ste_gp_lml_f64 for the whole batch");
    return gp_lml_launch(b, count, active, stream);
}

int ste_gp_predict_f64(const ste_gp_batch_f64* b, int32_t mmax, const int32_t* m, const double* xs, double* Kstar,
                       double* mean, double* var, void* stream) {
    stegp::GpParams p;
    int rc = gp_params(b, &p);
    if (rc) return rc;
    if (mmax <= 0 || !m || !xs || !Kstar || !mean || !var || !b->Kinv)
        return gp_fail("mmax > 0, m, xs, Kstar, mean, var and batch.Kinv are required (run ste_gp_lml_f64 with Kinv set first)");
    stegp::GpPredict q;
    q.B = p.B;
    q.mmax = mmax;
    q.mb_max = (mmax + 63) / 64;
    q.ldk = p.ld;
    q.m = m;
    q.xs = xs;
    q.Kstar = Kstar;
    q.Kinv = b->Kinv;
    q.mean = mean;
    q.var = var;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stegp::gp_kstar, dim3(q.mb_max, p.B), dim3(256), 0, s, p, q);
    hipLaunchKernelGGL(stegp::gp_predict, dim3(q.mb_max, p.B), dim3(256), 0, s, p, q);
    return gp_hip(hipGetLastError(), "gp_predict launch");
}

}  // extern "C"

// micro-benchmark 4: the panel_gemm_t loop in isolation, pieces switchable at compile time
// -DTILEMAJOR (round 4): the operands laid out as contiguous 64 x 64 tiles (rows of a tile 512 B apart) instead of rows of
// the whole matrix (16.5 KB apart): does the memory system deliver a wave-level load of 16 rows x 128 B faster then?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#ifndef SHAREMASK
#define SHAREMASK 255  // 255: every workgroup streams its own rows; 0: all workgroups read the same, cache-resident rows
#endif
constexpr int T = 64, LDB = 66;
#if defined(TILEMAJOR) || defined(FRAGMAJOR)
constexpr int TSTEP = 4096;  // doubles from one k-block of a panel to the next
#else
constexpr int TSTEP = 64;
#endif
struct RowFrag { v4d v[4]; };
typedef double v2d __attribute__((ext_vector_type(2)));
// -DFRAGMAJOR (round 4): operands stored in MFMA fragment order -- tile (64 x 64) major, inside a tile 16 chunks
// [strip][sub-block] of 256 doubles, inside a chunk [h][lane][2]: the two 16-byte loads of a lane's four k-values are
// each 1 KB contiguous over the wave (8 full cache lines instead of 16 half-used ones)
__device__ __forceinline__ void load_rows(RowFrag& f, const double* const (&pr)[4], int k) {
#pragma unroll
#ifdef FRAGMAJOR  // pr[n] = chunk (strip n, sub 0) of tile 0 of the panel + 2 * lane
    for (int n = 0; n < 4; ++n) {
        const double* q = pr[n] + (size_t)(k >> 6) * 4096 + ((k >> 4) & 3) * 256;
        const v2d lo = *reinterpret_cast<const v2d*>(q), hi = *reinterpret_cast<const v2d*>(q + 128);
        f.v[n] = v4d{lo[0], lo[1], hi[0], hi[1]};
    }
#elif defined(TILEMAJOR)  // k = 64 kb + 16 sub: tile kb of the panel, column 16 sub of it
    for (int n = 0; n < 4; ++n) f.v[n] = *reinterpret_cast<const v4d*>(pr[n] + (size_t)(k >> 6) * 4096 + (k & 63));
#else
    for (int n = 0; n < 4; ++n) f.v[n] = *reinterpret_cast<const v4d*>(pr[n] + k);
#endif
}
__device__ __forceinline__ void stage_load(v4d (&st)[4], const double* src, size_t ld, int tid) {
#pragma unroll
#ifdef FRAGMAJOR  // the tile is 32 KB contiguous and goes to LDS as it is
    for (int q = 0; q < 4; ++q) st[q] = *reinterpret_cast<const v4d*>(src + 4 * tid + 1024 * q);
#elif defined(TILEMAJOR)  // src names the tile: 64 rows of 64 doubles, contiguous
    for (int q = 0; q < 4; ++q) { const int e = tid + 256 * q; st[q] = *reinterpret_cast<const v4d*>(src + (size_t)(e >> 4) * 64 + (e & 15) * 4); }
#else
    for (int q = 0; q < 4; ++q) { const int e = tid + 256 * q; st[q] = *reinterpret_cast<const v4d*>(src + (size_t)(e >> 4) * ld + (e & 15) * 4); }
#endif
}
__device__ __forceinline__ void stage_store(double* dst, const v4d (&st)[4], int tid) {
#pragma unroll
#ifdef FRAGMAJOR
    for (int q = 0; q < 4; ++q) *reinterpret_cast<v4d*>(dst + 4 * tid + 1024 * q) = -st[q];
    return;
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int e = tid + 256 * q; *reinterpret_cast<v4d*>(dst + (e >> 4) * LDB + (e & 15) * 4) = -st[q]; }
}
__device__ __forceinline__ void mma_sub(v4d (&acc)[4][4], const double* blk, int sub, const RowFrag& own, int r, int g, bool live) {
    v4d a[4];
#pragma unroll
#ifdef FRAGMAJOR
    for (int m = 0; m < 4; ++m) {
        const double* q = blk + (m * 4 + sub) * 256 + 2 * (r + 16 * g);
        const v2d lo = *reinterpret_cast<const v2d*>(q), hi = *reinterpret_cast<const v2d*>(q + 128);
        a[m] = v4d{lo[0], lo[1], hi[0], hi[1]};
    }
#else
    for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const v4d*>(blk + (r + 16 * m) * LDB + 16 * sub + 4 * g);
#endif
    if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][e], own.v[n][e], acc[m][n], 0, 0, 0);
    }
}
__global__ __launch_bounds__(256) void k(const double* A, double* out, long long* ticks, int nblk, int wb0, size_t ld) {
    __shared__ __attribute__((aligned(16))) double stage[2 * T * LDB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const double* base = A + (size_t)(blockIdx.x & SHAREMASK) * 5 * 64 * ld;   // 5 row panels of 64 rows per workgroup
    const double* shared = base;
    const double* own[4];
#ifdef FRAGMAJOR
    for (int n = 0; n < 4; ++n) own[n] = base + (size_t)(1 + wave) * 64 * ld + (size_t)n * 1024 + 2 * lane;
#elif defined(TILEMAJOR)  // panel p of the workgroup = ld / 64 tiles of 4096 doubles
    for (int n = 0; n < 4; ++n) own[n] = base + (size_t)(1 + wave) * 64 * ld + (size_t)(r + 16 * n) * 64 + 4 * g;
#else
    for (int n = 0; n < 4; ++n) own[n] = base + (size_t)(64 * (1 + wave) + r + 16 * n) * ld + 4 * g;
#endif
    v4d acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = v4d{0, 0, 0, 0};
    const int kb0 = 0, kb1 = nblk;
#ifdef DIRECT
    // -DDIRECT (round 4): no LDS and no barrier -- every wave loads the shared panel's fragments itself, like its own rows
    // (the four waves' copies of a block come from L1 / L2), DIRECT = number of rotating fragment sets per operand (3 or 4)
    const double* shr[4];
    for (int m = 0; m < 4; ++m) shr[m] = shared + (size_t)(r + 16 * m) * ld + 4 * g;
    RowFrag fo[DIRECT], fs[DIRECT];
#pragma unroll
    for (int q = 0; q < DIRECT; ++q) { load_rows(fo[q], own, kb0 * T + 16 * q); load_rows(fs[q], shr, kb0 * T + 16 * q); }
    long long t0 = __builtin_amdgcn_s_memtime();
    const int nsub = 4 * (kb1 - kb0);
    for (int sb = 0; sb < nsub; sb += DIRECT) {
#pragma unroll
        for (int q = 0; q < DIRECT; ++q) {
            if (wb0 == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(fs[q].v[m][e], fo[q].v[n][e], acc[m][n], 0, 0, 0);
            }
            const int kn = ((sb + q + DIRECT) % nsub) * 16;
            load_rows(fo[q], own, kn);
            load_rows(fs[q], shr, kn);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
#pragma unroll
    for (int q = 0; q < DIRECT; ++q) s += fo[q].v[0][0] + fs[q].v[0][0];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    (void)stage; (void)wave;
#else
    v4d st[4];
    stage_load(st, shared + (size_t)kb0 * TSTEP, ld, tid);
    RowFrag f0, f1, f2, f3;
    load_rows(f0, own, kb0 * T); load_rows(f1, own, kb0 * T + 16); load_rows(f2, own, kb0 * T + 32); load_rows(f3, own, kb0 * T + 48);
    stage_store(stage, st, tid);
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int kb = kb0; kb < kb1; ++kb) {
        double* cur = stage + ((kb - kb0) & 1) * (T * LDB);
        double* nxt = stage + (((kb - kb0) & 1) ^ 1) * (T * LDB);
        const bool more = kb + 1 < kb1, live = kb >= wb0;
        const int kn = (more ? kb + 1 : kb0) * T;
        stage_load(st, shared + (size_t)(kn / T) * TSTEP, ld, tid);
        mma_sub(acc, cur, 0, f0, r, g, live); load_rows(f0, own, kn);
        mma_sub(acc, cur, 1, f1, r, g, live); load_rows(f1, own, kn + 16);
        mma_sub(acc, cur, 2, f2, r, g, live); load_rows(f2, own, kn + 32);
        mma_sub(acc, cur, 3, f3, r, g, live); load_rows(f3, own, kn + 48);
        stage_store(nxt, st, tid);
        __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
#endif
}
int main() {
#ifndef NBLK
#define NBLK 32
#endif
    const int nblk = NBLK; const size_t ld = nblk * 64 + 16;
    const size_t elems = (size_t)256 * 5 * 64 * ld;
    double* A; double* out; long long* ticks;
    (void)hipMalloc(&A, elems * 8); (void)hipMemset(A, 0, elems * 8); (void)hipMalloc(&out, 8 * 256 * 256); (void)hipMalloc(&ticks, 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, A, out, ticks, nblk, rep == 2 ? 1000 : 0, ld);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("rep %d: %.0f ticks per block (wave 0 of WG 0), %.3f ms, %.1f TF/s %s\n", rep, (double)t / nblk, ms,
               2048.0 * 256 * nblk * 4 * 256 / ms / 1e9, rep == 2 ? "(MFMAs skipped)" : "");
    }
    return 0;
}

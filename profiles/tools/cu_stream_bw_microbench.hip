// cu_stream_bw_microbench.hip — how fast can a SUBSET of the CUs stream from HBM, as a function of waves per CU and of
// independent 8-byte-per-lane loads in flight per wave?  (The smoother's cost on its CU partition is set by this.)
//   hipcc -O3 --offload-arch=gfx950 cu_stream_bw_microbench.hip -o bw && ./bw
// Access pattern = the UKF/URTSS layout: rows of B doubles ([k][c][t], t fastest), lane t of a wave reads 8 B, a wave
// instruction covers 512 contiguous bytes; each wave walks k and reads U rows per step (all independent), sums them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, int WIDE>
__global__ void stream_read(const double* __restrict__ src, double* __restrict__ out, int B, int K) {
    // WIDE = 1: 8 B per lane; WIDE = 2: 16 B per lane (a wave then covers 128 tracks)
    const size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * WIDE;
    if (t >= (size_t)B) return;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
        const double* row = src + ((size_t)k * U) * B + t;
        double v[U][WIDE];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (WIDE == 2) {
                const double2 d = *reinterpret_cast<const double2*>(row + (size_t)u * B);
                v[u][0] = d.x;
                v[u][WIDE - 1] = d.y;
            } else {
                v[u][0] = row[(size_t)u * B];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int w = 0; w < WIDE; ++w) acc += v[u][w];
    }
    out[t] = acc;
}

template <int U, int WIDE>
double run(hipStream_t s, const double* src, double* out, int B, int K, int block) {
    const int threads = B / WIDE;
    dim3 grid((threads + block - 1) / block), blk(block);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((stream_read<U, WIDE>), grid, blk, 0, s, src, out, B, K);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((stream_read<U, WIDE>), grid, blk, 0, s, src, out, B, K);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)B * K * U * 8.0 * 3 / (ms * 1e-3) / 1e9;  // GB/s
}

int main() {
    const size_t total = (size_t)3 << 30;  // 3 GiB source (beyond the 256 MiB Infinity Cache)
    double *src, *out;
    CK(hipMalloc(&src, total)); CK(hipMalloc(&out, 64 << 20));
    CK(hipMemset(src, 0, total));
    const int cus_list[] = {256, 96, 64, 32};
    printf("%5s %7s %6s %3s %5s %10s %12s\n", "CUs", "tracks", "waves", "U", "wide", "GB/s", "GB/s per CU");
    for (int cus : cus_list) {
        uint32_t mask[8] = {0};
        for (int i = 256 - cus; i < 256; ++i) mask[i / 32] |= 1u << (i % 32);
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
        for (int waves_per_cu : {2, 4, 8, 16}) {
            const int B = cus * waves_per_cu * 64;  // one wave per 64 tracks, exactly waves_per_cu resident waves per CU
            auto line = [&](int U, int wide, double gbs) {
                printf("%5d %7d %6d %3d %5d %10.1f %12.2f\n", cus, B * (wide == 2 ? 2 : 1), waves_per_cu, U, wide, gbs, gbs / cus);
            };
            { const int U = 8;  const int K = (int)(total / 8 / U / B); line(U, 1, run<8, 1>(s, src, out, B, K, 64)); }
            { const int U = 20; const int K = (int)(total / 8 / U / B); line(U, 1, run<20, 1>(s, src, out, B, K, 64)); }
            { const int U = 44; const int K = (int)(total / 8 / U / B); line(U, 1, run<44, 1>(s, src, out, B, K, 64)); }
            { const int U = 20; const int B2 = B * 2; const int K = (int)(total / 8 / U / B2); line(U, 2, run<20, 2>(s, src, out, B2, K, 64)); }
        }
        CK(hipStreamDestroy(s));
    }
    return 0;
}

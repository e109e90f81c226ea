// micro-benchmark 2: f64 MFMA rate with accumulators in AGPRs vs VGPRs, 16 accumulators, varying A/B
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <bool AGPR>
__global__ __launch_bounds__(256) void k(double* out, long long* ticks, int iters) {
    v4d acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 1.0 + threadIdx.x * 1e-6 * i; }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (AGPR) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[4 * m + n]) : "v"(a[m]), "v"(b[n]));
                    else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[4 * m + n]) : "v"(a[m]), "v"(b[n]));
                }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    double* out; long long* ticks;
    (void)hipMalloc(&out, 8 * 1024 * 1024); (void)hipMalloc(&ticks, 8);
    int iters = 2000;
    for (int variant = 0; variant < 2; ++variant) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        if (variant == 0) hipLaunchKernelGGL(k<false>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        else hipLaunchKernelGGL(k<true>, dim3(256), dim3(256), 0, 0, out, ticks, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        long long n = (long long)iters * 256;
        printf("%s accumulators: %.1f ticks/mfma, %.2f ms, %.1f TF/s\n", variant ? "AGPR" : "VGPR", (double)t / n, ms, 2048.0 * n * 4 * 256 / ms / 1e9);
    }
    return 0;
}

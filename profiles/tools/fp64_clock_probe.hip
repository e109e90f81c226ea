// micro-benchmark: what one SIMD issues per second in fp64 FMAs under sustained load, as a function of waves per SIMD and
// independent chains per wave -- the roof the filter kernels' instruction counts are priced against.
// Wall time comes from HIP events around the launch; the effective shader clock follows from the densest case
// (one wave64 fp64 FMA occupies the SIMD's 16-lane pipe for 4 clocks).
//   hipcc -O3 --offload-arch=gfx950 fp64_clock_probe.hip -o fp64_clock_probe && ./fp64_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int kChains>
__global__ __launch_bounds__(64) void fma_chains(double* out, int iters, double c0) {
    double acc[kChains];
    const double x = 1.0 - threadIdx.x * 1e-12;
#pragma unroll
    for (int c = 0; c < kChains; ++c) acc[c] = 0.25 + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < kChains; ++c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(x), "v"(c0));
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < kChains; ++c) s += acc[c];
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
}

template <int kChains>
static void run(double* out, int waves, int iters_total) {
    const int iters = iters_total / kChains;  // the same number of FMAs per wave whatever the chain count
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fma_chains<kChains>, dim3(waves), dim3(64), 0, 0, out, iters / 8, 1e-9);  // warm-up
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fma_chains<kChains>, dim3(waves), dim3(64), 0, 0, out, iters, 1e-9);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fmas_per_wave = (double)iters * 16 * kChains;
    const double per_simd = fmas_per_wave * (waves < 1024 ? 1024 : waves) / 1024.0;  // 256 CUs x 4 SIMDs; fewer waves: per wave
    printf("waves %5d  chains %d  %8.3f ms  %7.1f M wave-FMA/s per SIMD  (x4 clocks = %6.0f MHz if the pipe were full)  %6.1f TFLOP/s\n",
           waves, kChains, ms, per_simd / ms * 1e-3, per_simd / ms * 1e-3 * 4, fmas_per_wave * waves * 128 / ms * 1e-9);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

int main() {
    double* out;
    (void)hipMalloc(&out, sizeof(double) * 64 * 8192);
    const int total = 1 << 17;  // x16 FMAs per wave: ~2 M wave-instructions, a few ms
    // one wave per SIMD on a part of the chip (the dispatcher spreads them): does a wave run faster when the chip is emptier?
    for (int waves : {157, 314, 628}) run<8>(out, waves, total);
    for (int waves : {1024, 2048, 4096}) {
        run<1>(out, waves, total);
        run<2>(out, waves, total);
        run<4>(out, waves, total);
        run<8>(out, waves, total);
    }
    // sustained: ten back-to-back launches of the densest case (does the clock sag under continued fp64 load?)
    for (int rep = 0; rep < 3; ++rep) run<8>(out, 4096, total * 8);
    (void)hipFree(out);
    return 0;
}

// Samples the shader clock while something else (another process) loads the GPU: one wave spins for ~0.5 ms and reports
// s_memtime ticks (shader clock) per s_memrealtime tick (100 MHz reference).  Prints one line per sample.
//   hipcc -O3 --offload-arch=gfx950 shader_clock_sampler.hip -o shader_clock_sampler && ./shader_clock_sampler 400 10
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

__global__ __launch_bounds__(64) void spin(unsigned long long* out, unsigned long long ref_ticks) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ref_ticks) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

int main(int argc, char** argv) {
    const int samples = argc > 1 ? atoi(argv[1]) : 200;
    const int gap_ms = argc > 2 ? atoi(argv[2]) : 10;
    unsigned long long* out;
    (void)hipHostMalloc(&out, 16);
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < samples; ++i) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, out, 50000ull);  // 0.5 ms of the 100 MHz reference
        (void)hipStreamSynchronize(s);
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%8.3f s  %7.1f MHz\n", t, 100.0 * (double)out[0] / (double)out[1]);
        fflush(stdout);
        std::this_thread::sleep_for(std::chrono::milliseconds(gap_ms));
    }
    return 0;
}

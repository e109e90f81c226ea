// micro-benchmark: cost of feeding fp64 FMA coefficients from VGPRs, AGPRs (v_accvgpr_read x2) or SALU literals (s_mov_b32 x2)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
__global__ __launch_bounds__(64) void k(double* out, long long* ticks, int iters, int mode, double c0) {
    double x = threadIdx.x * 1e-9 + 0.5, acc = 0.25;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {          // coefficient already in a VGPR
        for (int it = 0; it < iters; ++it) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc) : "v"(x), "v"(c0));) }
    } else if (mode == 1) {   // coefficient parked in an AGPR pair, read back before each use
        asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1" :: "v"(__double2loint(c0)), "v"(__double2hiint(c0)) : "a0", "a1");
        for (int it = 0; it < iters; ++it) {
            REP16(asm volatile("v_accvgpr_read_b32 v40, a0\n v_accvgpr_read_b32 v41, a1\n v_fma_f64 %0, %0, %1, v[40:41]" : "+v"(acc) : "v"(x) : "v40", "v41");)
        }
    } else {                  // coefficient materialised by two SALU literal moves before each use
        for (int it = 0; it < iters; ++it) {
            REP16(asm volatile("s_mov_b32 s20, 0xb5e68a13\n s_mov_b32 s21, 0x3eeba404\n v_fma_f64 %0, %0, %1, s[20:21]" : "+v"(acc) : "v"(x) : "s20", "s21");)
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    double* out; long long* ticks;
    (void)hipMalloc(&out, 8 * 64 * 1024); (void)hipMalloc(&ticks, 8);
    const char* names[3] = {"VGPR coefficient", "AGPR coefficient (2 x v_accvgpr_read)", "SALU literal (2 x s_mov_b32)"};
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, out, ticks, 4000, mode, 1.25e-5);
        (void)hipDeviceSynchronize();
        long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("%-40s %.2f ticks per dependent FMA\n", names[mode], (double)t / (4000.0 * 16));
    }
    return 0;
}

// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS code base's access pattern:
// 8 bytes per lane (global_load_dwordx2 / global_store_dwordx2), 512-byte contiguous runs per wave, streaming.
// MI355X_MICROARCH.md says FETCH_SIZE reads exactly 1/2 for 16-B-per-lane streams and that other widths must be
// calibrated on a known byte count.  This kernel moves a known number of bytes with the same instructions the
// UKF/URTSS kernels use; profiles/README.md records the ratio measured with it.
//   hipcc -O3 --offload-arch=gfx950 pmc_calibrate.hip -o pmc_calibrate
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./pmc_calibrate
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- ./pmc_calibrate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void copy8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i] + 1.0;
}

int main() {
    const size_t n = (size_t)1 << 28;  // 2 GiB in, 2 GiB out: far beyond the 256 MiB Infinity Cache
    double *a, *b;
    if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) return 1;
    hipMemset(a, 0, n * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    copy8<<<2048, 256>>>(a, b, n);
    hipEventRecord(e0);
    copy8<<<2048, 256>>>(a, b, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("copy8: %zu bytes read + %zu bytes written per launch, %.3f ms, %.1f GB/s\n", n * 8, n * 8, ms,
           2.0 * n * 8 / ms / 1e6);
    return 0;
}

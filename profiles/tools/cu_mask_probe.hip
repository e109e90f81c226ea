// which physical CUs does a CU-masked stream use?  (xcc, se, cu) of every workgroup under a few mask patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <map>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
static void run(const char* name, const std::vector<unsigned>& mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int nwg = 4096;
    unsigned* d; (void)hipMalloc(&d, nwg * 8);
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(64), 0, s, d, 200000);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> h(nwg * 2); (void)hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per;
    for (int i = 0; i < nwg; ++i) { unsigned hw = h[2 * i + 1]; unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; per[h[2 * i]].insert((se << 8) | (sh << 4) | cu); }
    int total = 0; printf("%-28s", name);
    for (auto& kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  total %d\n", total);
    (void)hipFree(d); (void)hipStreamDestroy(s);
}
int main() {
    std::vector<unsigned> all(8, 0xffffffffu);
    run("all 256 bits", all);
    std::vector<unsigned> low160(8, 0); for (int i = 0; i < 160; ++i) low160[i / 32] |= 1u << (i % 32);
    run("bits 0..159", low160);
    std::vector<unsigned> low32(8, 0); low32[0] = 0xffffffffu;
    run("bits 0..31", low32);
    std::vector<unsigned> low8(8, 0); low8[0] = 0xffu;
    run("bits 0..7", low8);
    std::vector<unsigned> stride8(8, 0); for (int i = 0; i < 256; i += 8) stride8[i / 32] |= 1u << (i % 32);
    run("every 8th bit", stride8);
    std::vector<unsigned> hi96(8, 0); for (int i = 160; i < 256; ++i) hi96[i / 32] |= 1u << (i % 32);
    run("bits 160..255", hi96);
    return 0;
}

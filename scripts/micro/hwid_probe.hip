// Diagnostic: where do the four wavefronts of 256-thread workgroups (57 KB LDS, 2 per CU) land?  SIMD of each wave and
// whether the wave 0s of the two co-resident workgroups share a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k_probe(unsigned* out, int spin) {
    extern __shared__ double lds[];
    const int wave = threadIdx.x >> 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(10);   // keep every block resident for a while
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + wave) * 2] = hw; out[(blockIdx.x * 4 + wave) * 2 + 1] = xcc; }
    if (lds[threadIdx.x] < 0) out[0] = 0;
}
int main() {
    const int nblk = 400;
    unsigned* d; hipMalloc(&d, nblk * 8 * 4);
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 60000);
    hipLaunchKernelGGL(k_probe, dim3(nblk), dim3(256), 57000, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nblk * 8);
    hipMemcpy(h.data(), d, nblk * 8 * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu_blocks;   // (xcc, se, sh, cu) -> blocks
    int simd_hist[4][4] = {};
    for (int b = 0; b < nblk; ++b) {
        for (int w = 0; w < 4; ++w) simd_hist[w][(h[(b * 4 + w) * 2] >> 4) & 3]++;
        const unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
        const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 15);
        cu_blocks[key].push_back(b);
    }
    for (int w = 0; w < 4; ++w) printf("wave %d SIMD histogram: %d %d %d %d\n", w, simd_hist[w][0], simd_hist[w][1], simd_hist[w][2], simd_hist[w][3]);
    int shared = 0, pairs = 0, singles = 0;
    for (auto& kv : cu_blocks) {
        if (kv.second.size() == 1) { ++singles; continue; }
        ++pairs;
        const int b0 = kv.second[0], b1 = kv.second[1];
        if (((h[b0 * 8] >> 4) & 3) == ((h[b1 * 8] >> 4) & 3)) ++shared;
        if (pairs <= 6) printf("CU key %05x: blocks %d and %d, wave-0 SIMDs %u %u, slots %u %u\n", kv.first, b0, b1, (h[b0 * 8] >> 4) & 3, (h[b1 * 8] >> 4) & 3, h[b0 * 8] & 15, h[b1 * 8] & 15);
    }
    printf("%zu CUs used, %d with one block, %d with two; wave 0s share a SIMD in %d of those\n", cu_blocks.size(), singles, pairs, shared);
    return 0;
}

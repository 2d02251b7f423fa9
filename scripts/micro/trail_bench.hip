// Micro-benchmark (diagnostic, not product): the trailing-update pattern of potrf_reg_body in isolation.
// Six worker wavefronts (waves 1,2,3,5,6,7 of a 512-thread workgroup) each own NT register tiles and run
//   tile -= panel_a^T panel_b   (8 ds_read_b64, 4 dependent FP64 MFMAs)
// Variants isolate what the loop costs beyond the 4 x 64-cycle MFMA issue.
//   hipcc -O3 --offload-arch=gfx950 trail_bench.hip -o trail_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define NT 12

template <int VAR>
__global__ __launch_bounds__(512) void k_trail(double* out, int reps, unsigned mask, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) double sPan[14][4][64];
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    for (int i = tid; i < 14 * 256; i += 512) (&sPan[0][0][0])[i] = 1e-3 * (i % 97);
    __syncthreads();
    d4 acc[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) acc[s] = d4{1.0 * s, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((mask >> wave) & 1) {
        for (int rep = 0; rep < reps; ++rep) {
            int z = 0;
            asm volatile("" : "+v"(z));
            const double* pPan = &sPan[0][0][0] + z + l;
            if (VAR == 0) {   // as the product: fetch the next slot's operands, then this slot's MFMAs
                double q[2][8];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { q[0][s4] = pPan[(1) * 256 + s4 * 64]; q[0][4 + s4] = pPan[(2) * 256 + s4 * 64]; }
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    if (s + 1 < NT) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) { q[(s + 1) & 1][s4] = pPan[((s + 1) % 13 + 1) * 256 + s4 * 64]; q[(s + 1) & 1][4 + s4] = pPan[((s + 2) % 13 + 1) * 256 + s4 * 64]; }
                    }
                    d4 x = acc[s];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) x = __builtin_amdgcn_mfma_f64_16x16x4f64(-q[s & 1][s4], q[s & 1][4 + s4], x, 0, 0, 0);
                    acc[s] = x;
                }
            } else if (VAR == 1) {   // no LDS: operands are registers
                const double a = pPan[256], b = pPan[512];
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    d4 x = acc[s];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, x, 0, 0, 0);
                    acc[s] = x;
                }
            } else if (VAR == 2) {   // two tiles in flight: interleave the MFMAs of slots s and s+1
                double q[2][16];
#pragma unroll
                for (int s = 0; s < NT; s += 2) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        q[0][s4] = pPan[((s) % 13 + 1) * 256 + s4 * 64]; q[0][4 + s4] = pPan[((s + 1) % 13 + 1) * 256 + s4 * 64];
                        q[0][8 + s4] = pPan[((s + 2) % 13 + 1) * 256 + s4 * 64]; q[0][12 + s4] = pPan[((s + 3) % 13 + 1) * 256 + s4 * 64];
                    }
                    d4 x = acc[s], y = acc[s + 1];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        x = __builtin_amdgcn_mfma_f64_16x16x4f64(-q[0][s4], q[0][4 + s4], x, 0, 0, 0);
                        y = __builtin_amdgcn_mfma_f64_16x16x4f64(-q[0][8 + s4], q[0][12 + s4], y, 0, 0, 0);
                    }
                    acc[s] = x; acc[s + 1] = y;
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0;
#pragma unroll
    for (int s = 0; s < NT; ++s) r += acc[s][0] + acc[s][1] + acc[s][2] + acc[s][3];
    out[tid] = r;
    if (l == 0) stamps[wave] = t1 - t0;
}

int main() {
    double* d; unsigned long long* st;
    hipMalloc(&d, 1 << 16); hipMalloc(&st, 64);
    const unsigned masks[] = {0x02, 0x0e, 0x22, 0xee, 0xff};
    const char* mn[] = {"wave 1 only", "waves 1-3 (one per SIMD)", "waves 1,5 (same SIMD)", "waves 1,2,3,5,6,7", "all 8"};
    for (int var = 0; var < 3; ++var)
        for (int m = 0; m < 5; ++m) {
            const int reps = 200;
            for (int it = 0; it < 2; ++it) {
                if (var == 0) hipLaunchKernelGGL(k_trail<0>, dim3(1), dim3(512), 0, 0, d, reps, masks[m], st);
                if (var == 1) hipLaunchKernelGGL(k_trail<1>, dim3(1), dim3(512), 0, 0, d, reps, masks[m], st);
                if (var == 2) hipLaunchKernelGGL(k_trail<2>, dim3(1), dim3(512), 0, 0, d, reps, masks[m], st);
            }
            hipDeviceSynchronize();
            unsigned long long h[8]; hipMemcpy(h, st, 64, hipMemcpyDeviceToHost);
            printf("var %d  %-28s cycles per tile per wave:", var, mn[m]);
            for (int w = 0; w < 8; ++w) if ((masks[m] >> w) & 1) printf(" %.0f", (double)h[w] / (reps * NT));
            printf("\n");
        }
    return 0;
}

// Micro-benchmarks used to calibrate the latency model of the chain-bound kernels (diagnostic only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_chain(double* out, int iters, int mode, unsigned long long* stamps) {
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.999999;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) {          // dependent f64 FMA chain
        for (int i = 0; i < iters; ++i) x = fma(x, y, 1e-9);
    } else if (mode == 1) {   // rsqrt (v_rsq_f64 + 2 Newton) chain
        for (int i = 0; i < iters; ++i) {
            double r = __builtin_amdgcn_rsq(x);
            double h = 0.5 * x;
            r = r * (1.5 - h * r * r);
            r = r * (1.5 - h * r * r);
            x = r + 1.0;
        }
    } else if (mode == 2) {   // 1/sqrt via libm chain
        for (int i = 0; i < iters; ++i) x = 1.0 / sqrt(x) + 1.0;
    } else if (mode == 3) {   // readlane broadcast + fma chain
        for (int i = 0; i < iters; ++i) {
            int lo = __builtin_amdgcn_readlane(__double2loint(x), 5);
            int hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
            double b = __hiloint2double(hi, lo);
            x = fma(b, y, x * 1e-3);
        }
    } else if (mode == 4) {   // shfl (ds_bpermute) broadcast + fma chain
        for (int i = 0; i < iters; ++i) {
            double b = __shfl(x, 5);
            x = fma(b, y, x * 1e-3);
        }
    } else if (mode == 5) {   // LDS write -> read round trip chain
        __shared__ double sh[64];
        for (int i = 0; i < iters; ++i) {
            sh[threadIdx.x & 63] = x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            x = sh[(threadIdx.x + 1) & 63] * y;
        }
    } else if (mode == 6) {   // independent f64 FMAs (8 chains)
        double a[8];
        for (int k = 0; k < 8; ++k) a[k] = x + k;
        for (int i = 0; i < iters; ++i)
            for (int k = 0; k < 8; ++k) a[k] = fma(a[k], y, 1e-9);
        for (int k = 0; k < 8; ++k) x += a[k];
    } else if (mode == 7) {   // dependent f64 MFMA chain
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 c = {x, x, x, x};
        for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, c, 0, 0, 0);
        x = c[0] + c[1] + c[2] + c[3];
    } else if (mode == 8) {   // __syncthreads chain
        for (int i = 0; i < iters; ++i) { __syncthreads(); x = x * y; }
    } else if (mode == 9) {   // 8 independent DPP fnmacs (row_newbcast) per iteration
        double a[8];
        for (int k = 0; k < 8; ++k) a[k] = x + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(y), "v"(y));
        }
        for (int k = 0; k < 8; ++k) x += a[k];
    } else if (mode == 10) {   // 8 independent asm v_fma_f64 per iteration
        double a[8];
        for (int k = 0; k < 8; ++k) a[k] = x + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(a[k]) : "v"(y), "v"(y));
        }
        for (int k = 0; k < 8; ++k) x += a[k];
    } else if (mode == 11) {   // 4 independent f64 MFMAs per iteration
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 c0 = {x, x, x, x}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, c3, 0, 0, 0);
        }
        x = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (mode == 12) {   // 8 independent f64 FMAs with a scalar (SGPR) multiplier
        double a[8];
        for (int k = 0; k < 8; ++k) a[k] = x + k;
        const int lo = __builtin_amdgcn_readfirstlane(__double2loint(y)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(y));
        const double ys = __hiloint2double(hi, lo);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(a[k]) : "s"(ys), "v"(y));
        }
        for (int k = 0; k < 8; ++k) x += a[k];
    } else if (mode == 13) {   // 8 independent f32 FMAs
        float a[8];
        for (int k = 0; k < 8; ++k) a[k] = (float)x + k;
        const float yf = (float)y;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[k]) : "v"(yf), "v"(yf));
        }
        for (int k = 0; k < 8; ++k) x += a[k];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

void run_dpp_check();
__global__ void k_pair(double* out, int iters, int partner, int kind, unsigned long long* stamps) {
    const int wave = threadIdx.x >> 6;
    if (wave != 0 && wave != partner) return;
    double y = 0.999999, x = 1.0 + threadIdx.x * 1e-9;
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = x + k;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (kind == 0 || wave == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(y), "v"(y));
        }
    } else {   // partner runs dependent f64 MFMAs
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 c = {x, x, x, x};
        for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, c, 0, 0, 0);
        a[0] = c[0] + c[1];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < 8; ++k) x += a[k];
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) stamps[0] = t1 - t0;
    if (threadIdx.x == 64 * partner) stamps[1] = t1 - t0;
}
// how does the FP64 MFMA rate scale with the number of wavefronts of one workgroup (one CU) issuing them?
__global__ void k_mfma_scale(double* out, int iters, int nactive, unsigned long long* stamps) {
    const int wave = threadIdx.x >> 6;
    typedef double d4 __attribute__((ext_vector_type(4)));
    double y = 0.999999, x = 1.0 + threadIdx.x * 1e-9;
    d4 c = {x, x, x, x};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < nactive)
        for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, c, 0, 0, 0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c[0] + c[1];
    if ((threadIdx.x & 63) == 0) stamps[wave] = t1 - t0;
}
int main() {
    run_dpp_check();
    double* d; unsigned long long* st;
    hipMalloc(&d, 1 << 20); hipMalloc(&st, 64);
    const char* names[] = {"dep fma f64", "rsq+2 newton", "1/sqrt libm", "readlane bcast+fma", "shfl bcast+fma", "lds write->read", "8 indep fma f64 (per 8)", "dep mfma f64 16x16x4", "__syncthreads (512 thr)", "8 indep fnmac_dpp f64", "8 indep asm fma f64", "4 indep mfma f64", "8 indep fma f64 sgpr mult", "8 indep fma f32"};
    for (int grid : {1, 256}) {
        for (int mode = 0; mode < 14; ++mode) {
            const int iters = 20000;
            const int threads = (mode == 8) ? 512 : 64;
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_chain, dim3(grid), dim3(threads), 0, 0, d, iters, mode, st);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_chain, dim3(grid), dim3(threads), 0, 0, d, iters, mode, st);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
            double clk = (double)h[0] / (double)h[1] * 100.0;   // MHz
            printf("grid %3d  %-26s  %8.1f ns/iter  %7.1f cyc/iter  clock %.0f MHz  (event %.3f ms)\n", grid, names[mode],
                   h[1] * 10.0 / iters, (double)h[0] / iters, clk, ms);
        }
    }
    for (int na : {1, 2, 4, 6, 8}) {
        const int iters = 20000;
        hipLaunchKernelGGL(k_mfma_scale, dim3(1), dim3(512), 0, 0, d, iters, na, st);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, st, 64, hipMemcpyDeviceToHost);
        printf("mfma f64 chains on %d waves of one workgroup: cycles per MFMA per wave:", na);
        for (int w = 0; w < na; ++w) printf(" %.1f", (double)h[w] / iters);
        printf("\n");
    }
    for (int kind = 0; kind < 2; ++kind)
        for (int partner : {9, 4, 1}) {   // 9 = nobody (block has 8 waves)
            const int iters = 20000;
            hipLaunchKernelGGL(k_pair, dim3(1), dim3(512), 0, 0, d, iters, partner, kind, st);
            hipDeviceSynchronize();
            unsigned long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
            printf("pair kind %d (0 = both DPP fnmac streams, 1 = partner runs dependent MFMAs) partner wave %d: wave0 %.1f cyc/iter(8 fnmac), partner %.1f cyc/iter\n",
                   kind, partner, (double)h[0] / iters, partner < 8 ? (double)h[1] / iters : 0.0);
        }
    return 0;
}
// ---- DPP row_newbcast semantics check for 64-bit ops -------------------------------------------
__global__ void k_dpp(double* out) {
    const int l = threadIdx.x;
    double a = 100.0 * (l >> 4) + (l & 15);   // row*100 + lane-in-row
    double b = 2.0, c = 0.5, d;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(a));
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(b));
    out[l] = d;        // expect row*100 + 7
    out[64 + l] = c;   // expect 0.5 - (row*100 + 5) * 2
}
struct DppCheck {
    DppCheck() {
        double* d; hipMalloc(&d, 128 * 8);
        hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, d);
        double h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        bool ok = true;
        for (int l = 0; l < 64; ++l) {
            const double row = l >> 4;
            if (h[l] != row * 100 + 7) ok = false;
            if (h[64 + l] != 0.5 - (row * 100 + 5) * 2.0) ok = false;
        }
        printf("DPP row_newbcast f64 semantics: %s  (lane 20: mov %.1f fmac %.1f)\n", ok ? "OK" : "MISMATCH", h[20], h[84]);
    }
};

void run_dpp_check() { DppCheck c; (void)c; }

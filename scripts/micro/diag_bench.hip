// Diagnostic: cycles of the 16x16 diagonal-tile factor+inverse sweep (DiagStep) in isolation.
#include "../../orcvio_amd/csrc/msckf_kernels.hpp"
#include <cstdio>
using namespace orcvio_amd;

template <int MODE>
__global__ void k_diag(const double* X, double* out, unsigned long long* st, int reps, int lanes) {
    const int l = threadIdx.x, cc = l & 15;
    if (l >= lanes) return;   // (lanes = 16: does a DP instruction get cheaper with three quarters of EXEC off?)
    double v0[16], y0[16];
    for (int c = 0; c < 16; ++c) { v0[c] = (c <= cc) ? X[cc * 16 + c] : 0.0; y0[c] = (c == cc) ? 1.0 : 0.0; }
    double acc = 0;
    int nz = 0, nn = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        double v[16], y[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { v[c] = v0[c] + rep * 1e-12; y[c] = y0[c]; }
        double dmin = 1e300;
        if (MODE == 0) { DiagStep<0>::run(v, y, 1e-300, dmin); acc += dmin; }
        if (MODE == 1) {   // updates only (no pivot math): issue cost of the 240 DPP FMAs
            double m = v[0], x = y[0];
            diag_fill<0, 1>(v, y, m, x); diag_fill<0, 2>(v, y, m, x); diag_fill<0, 3>(v, y, m, x); diag_fill<0, 4>(v, y, m, x);
            diag_fill<0, 5>(v, y, m, x); diag_fill<0, 6>(v, y, m, x); diag_fill<0, 7>(v, y, m, x); diag_fill<0, 8>(v, y, m, x);
            diag_fill<0, 9>(v, y, m, x); diag_fill<0, 10>(v, y, m, x); diag_fill<0, 11>(v, y, m, x); diag_fill<0, 12>(v, y, m, x);
            diag_fill<0, 13>(v, y, m, x); diag_fill<0, 14>(v, y, m, x); diag_fill<0, 15>(v, y, m, x);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) acc += v[c] + y[c];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[l] = acc + nz + nn;
    if (l == 0) st[0] = t1 - t0;
}

int main() {
    double h[256];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) h[i * 16 + j] = (i == j) ? 20.0 + i : 1.0 / (1 + i + j);
    double *dX, *dO; unsigned long long* dS;
    hipMalloc(&dX, sizeof(h)); hipMalloc(&dO, 64 * 8); hipMalloc(&dS, 8);
    hipMemcpy(dX, h, sizeof(h), hipMemcpyHostToDevice);
    const int reps = 200;
    for (int lanes = 64; lanes >= 16; lanes -= 48)
    for (int mode = 0; mode < 2; ++mode) {
        for (int it = 0; it < 2; ++it) {
            if (mode == 0) hipLaunchKernelGGL(k_diag<0>, dim3(1), dim3(64), 0, 0, dX, dO, dS, reps, lanes);
            else hipLaunchKernelGGL(k_diag<1>, dim3(1), dim3(64), 0, 0, dX, dO, dS, reps, lanes);
            hipDeviceSynchronize();
        }
        unsigned long long s; hipMemcpy(&s, dS, 8, hipMemcpyDeviceToHost);
        printf("lanes %d mode %d: %.0f cycles per 16x16 tile (mode 1 = 30 DPP fnmacs only)\n", lanes, mode, (double)s / reps);
    }
    return 0;
}

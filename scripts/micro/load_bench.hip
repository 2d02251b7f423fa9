// Micro-benchmark (diagnostic only): how long ONE workgroup of 8 wavefronts takes to bring the 91 upper tiles of a 208 x 208 FP64
// matrix (13 per worker wavefront x 7 loading wavefronts, 190 KB) into registers -- the prologue of the register-resident Cholesky --
// for different access patterns.  Core-clock cycles from the first load to the last value having arrived (s_waitcnt vmcnt(0)), max
// over the wavefronts; cold = the matrix was last written by a 256-workgroup kernel (other XCDs), warm = the same launch repeated.
//   mode 0  row-major matrix, 8 B per lane, lane (kk, cc) -> row kk + 4 r, column cc  (what potrf_reg_body does: 4 x 128 B per load)
//   mode 1  tile-major [tile][r][lane], 8 B per lane: 512 contiguous bytes per load
//   mode 2  tile-major [tile][half][lane][2], 16 B per lane: 1 KB contiguous per load, two loads per tile
//   mode 3  row-major, 16 B per lane: lane -> row l / 8 (+ 8 per load), columns 2 (l % 8) .. +1: 8 x 128 B per load, two loads per tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define NT 13
__global__ __launch_bounds__(512) void k_load(const double* __restrict__ X, int mode, unsigned long long* out, double* sink) {
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, kk = l >> 4, cc = l & 15;
    if (wave == 0) return;   // (the chain wave loads one tile only)
    const int w = wave - 1;  // 7 loading waves
    double acc[NT][4];
    const unsigned long long t0 = clock64();
    if (mode == 0) {
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int t = s * 7 + w, a = t / 13, b = t % 13;   // some tile (block row a, block column b) of a 13 x 13 grid
            const double* base = X + (size_t)(16 * a) * 208 + 16 * b;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s][r] = base[(size_t)(kk + 4 * r) * 208 + cc];
        }
    } else if (mode == 1) {
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const double* base = X + (size_t)(s * 7 + w) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s][r] = base[r * 64 + l];
        }
    } else if (mode == 2) {
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const double2* base = reinterpret_cast<const double2*>(X + (size_t)(s * 7 + w) * 256);
            const double2 v0 = base[l], v1 = base[64 + l];
            acc[s][0] = v0.x; acc[s][1] = v0.y; acc[s][2] = v1.x; acc[s][3] = v1.y;
        }
    } else {
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int t = s * 7 + w, a = t / 13, b = t % 13;
            const double* base = X + (size_t)(16 * a) * 208 + 16 * b;
            const double2 v0 = *reinterpret_cast<const double2*>(base + (size_t)(l >> 3) * 208 + 2 * (l & 7));
            const double2 v1 = *reinterpret_cast<const double2*>(base + (size_t)(8 + (l >> 3)) * 208 + 2 * (l & 7));
            acc[s][0] = v0.x; acc[s][1] = v0.y; acc[s][2] = v1.x; acc[s][3] = v1.y;
        }
    }
    const unsigned long long t1 = clock64();   // all loads issued
    double sum = 0.0;
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += acc[s][r];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = clock64();
    if (l == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = t2 - t0; }
    if (sum == 12345.678) sink[threadIdx.x] = sum;
}
__global__ void k_touch(double* X, int n) {   // rewrites the matrix from every XCD (what the producing GEMM does)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) X[i] = X[i] * 1.0000001 + 1e-9;
}
int main() {
    const int n = 208 * 208;
    double *X, *sink; unsigned long long* out;
    hipMalloc(&X, sizeof(double) * n); hipMalloc(&sink, sizeof(double) * 512); hipMalloc(&out, sizeof(unsigned long long) * 16);
    std::vector<double> h(n, 1.0);
    hipMemcpy(X, h.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        for (int cold = 0; cold < 2; ++cold) {
            std::vector<unsigned long long> issued, done;
            for (int rep = 0; rep < 12; ++rep) {
                if (cold) hipLaunchKernelGGL(k_touch, dim3((n + 255) / 256), dim3(256), 0, 0, X, n);
                hipLaunchKernelGGL(k_load, dim3(1), dim3(512), 0, 0, X, mode, out, sink);
                unsigned long long o[16];
                hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
                unsigned long long mi = 0, md = 0;
                for (int w = 1; w < 8; ++w) { mi = std::max(mi, o[2 * w]); md = std::max(md, o[2 * w + 1]); }
                if (rep >= 2) { issued.push_back(mi); done.push_back(md); }
            }
            std::sort(issued.begin(), issued.end()); std::sort(done.begin(), done.end());
            printf("mode %d %s: issued %llu  all data %llu cycles (median of 10; min %llu)\n", mode, cold ? "cold" : "warm",
                   issued[issued.size() / 2], done[done.size() / 2], done[0]);
        }
    }
    return 0;
}

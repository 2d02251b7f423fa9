// cov_ops.hpp -- covariance bookkeeping on a device-resident P: the three places
// besides the update where the reference touches state_cov, so that P never has to cross PCIe between updates.
//   k_cov_propagate_*  OrcVIO::processModel      (src/orcvio.cpp:800-816)
//   k_cov_augment      OrcVIO::stateAugmentation (:962-1010, feature / nuisance states behind the clones included)
//   k_cov_remove       OrcVIO::pruneImuStateBuffer (:2935-2951, non-Schmidt branch)
// All three are HBM-bound element kernels over an n x n matrix (n <= 406): coalesced row-major reads and writes.
#pragma once
#include <hip/hip_runtime.h>

namespace orcvio_amd {

// T[i][j] = sum_k Phi[i][k] P[k][j],  i < leg, j < n   (the first leg rows of Phi * P)
__global__ __launch_bounds__(256) void k_cov_propagate_rows(const double* __restrict__ P, int n, const double* __restrict__ Phi, int leg,
                                                            double* __restrict__ T) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= leg * n) return;
    const int i = idx / n, j = idx - i * n;
    double s = 0.0;
    for (int k = 0; k < leg; ++k) s += Phi[i * leg + k] * P[(size_t)k * n + j];
    T[idx] = s;
}
// out = P with  P_LL <- sym(T[:, :leg] Phi^T + Q),  P_LC <- T[:, leg:],  P_CL <- its transpose,  P_CC unchanged
__global__ __launch_bounds__(256) void k_cov_propagate_finish(const double* __restrict__ P, int n, const double* __restrict__ Phi,
                                                              const double* __restrict__ Q, int leg, const double* __restrict__ T,
                                                              double* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    double v;
    if (i < leg && j < leg) {
        double a = Q[i * leg + j], b = Q[j * leg + i];
        for (int k = 0; k < leg; ++k) { a += T[(size_t)i * n + k] * Phi[j * leg + k]; b += T[(size_t)j * n + k] * Phi[i * leg + k]; }
        v = 0.5 * (a + b);
    } else if (i < leg) {
        v = T[(size_t)i * n + j];
    } else if (j < leg) {
        v = T[(size_t)j * n + i];
    } else {
        v = P[idx];
    }
    out[idx] = v;
}
// out (n+6)^2: the new clone copies rows/cols (0:3, 6:9) of P and is inserted at `pose` = n - rest_rows, i.e. behind the
// clones and in front of the feature / nuisance states (src/orcvio.cpp:976-1003: the rest block moves down / right by 6)
__global__ __launch_bounds__(256) void k_cov_augment(const double* __restrict__ P, int n, int pose, double* __restrict__ out) {
    const int m = n + 6;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= m * m) return;
    const int i = idx / m, j = idx - i * m;
    const bool ni = i >= pose && i < pose + 6, nj = j >= pose && j < pose + 6;
    // source row: old states keep their index (shifted back by 6 behind the new clone); new 0..2 -> theta, 3..5 -> p (6:9)
    const int si = ni ? (i - pose < 3 ? i - pose : i - pose + 3) : (i < pose ? i : i - 6);
    const int sj = nj ? (j - pose < 3 ? j - pose : j - pose + 3) : (j < pose ? j : j - 6);
    // [[P, P12^T], [P12, P11]] then (X + X^T)/2 as :1008-1010: the off-diagonal blocks are transposes of each other already
    double v;
    if (ni == nj) v = 0.5 * (P[(size_t)si * n + sj] + P[(size_t)sj * n + si]);
    else if (ni) v = P[(size_t)si * n + sj];
    else v = P[(size_t)sj * n + si];
    out[idx] = v;
}
// out m^2 = P without the rows/cols flagged in drop[] (drop[k] = 1: state k is removed); map[k'] = k precomputed
__global__ __launch_bounds__(256) void k_cov_remove(const double* __restrict__ P, int n, const int* __restrict__ map, int m,
                                                    double* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= m * m) return;
    const int i = idx / m, j = idx - i * m;
    out[idx] = P[(size_t)map[i] * n + map[j]];
}


// ---- the resident square-root factor S (P = S S^T), stored transposed: S(j, i) = F[i * ld + j], i < k (columns of S),
//      j < n (states) -- the layout of the Cholesky factor of the prior and of Z = L_M^-1 S^T --------------------------------
// commit: S+ = sigma Z^T if the update was applied (apply == nullptr or *apply != 0), else the prior's own factor
__global__ __launch_bounds__(256) void k_fac_commit(const double* __restrict__ Z, int ldz, int k, int n, double sigma,
                                                    const int* __restrict__ apply, const double* __restrict__ prior, long sLi, long sLj,
                                                    double* __restrict__ out, int ldo) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= k * n) return;
    const int i = idx / n, j = idx - i * n;
    const bool app = apply ? (*apply != 0) : true;
    out[(size_t)i * ldo + j] = app ? sigma * Z[(size_t)i * ldz + j] : prior[(long)j * sLi + (long)i * sLj];
}
// stateAugmentation on the factor: the new clone's six rows of S are copies of the IMU's (theta, p) rows
__global__ __launch_bounds__(256) void k_fac_augment(const double* __restrict__ F, int ld, int k, int n, int pose,
                                                     double* __restrict__ out, int ldo) {
    const int m = n + 6;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= k * m) return;
    const int i = idx / m, j = idx - i * m;
    const bool nj = j >= pose && j < pose + 6;
    const int sj = nj ? (j - pose < 3 ? j - pose : j - pose + 3) : (j < pose ? j : j - 6);
    out[(size_t)i * ldo + j] = F[(size_t)i * ld + sj];
}
// marginalisation on the factor: the rows of S of the removed states are deleted (map[j'] = old index of new state j')
__global__ __launch_bounds__(256) void k_fac_remove(const double* __restrict__ F, int ld, int k, const int* __restrict__ map, int m,
                                                    double* __restrict__ out, int ldo) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= k * m) return;
    const int i = idx / m, j = idx - i * m;
    out[(size_t)i * ldo + j] = F[(size_t)i * ld + map[j]];
}

}  // namespace orcvio_amd

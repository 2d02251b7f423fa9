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
// commit: S+ = sigma Z^T if the update was applied (apply == nullptr or *apply != 0, and *fail == 0), else the prior's own factor
__global__ __launch_bounds__(256) void k_fac_commit(const double* __restrict__ Z, int ldz, int k, int n, double sigma,
                                                    const int* __restrict__ apply, const double* __restrict__ prior, long sLi, long sLj,
                                                    double* __restrict__ out, int ldo, const int* __restrict__ fail) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= k * n) return;
    const int i = idx / n, j = idx - i * n;
    const bool app = (apply ? (*apply != 0) : true) && fail[0] == 0 && fail[1] == 0;   // (fail: the pivot counters of chol(M); k_finish_sqrt kept P)
    out[(size_t)i * ldo + j] = app ? sigma * Z[(size_t)i * ldz + j] : prior[(long)j * sLi + (long)i * sLj];
}
// prefactor: the Cholesky of the REVERSED covariance (potrf_reg_body, rev) gives S(j, i) = L'(n-1-j, i); stored in the resident
// factor's own layout F[i * ld + j] (rows of the state in natural order), padding zeroed
__global__ __launch_bounds__(256) void k_fac_flip(const double* __restrict__ Rrev, int ldr, int n, double* __restrict__ out, int ldo) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * ldo) return;
    const int i = idx / ldo, j = idx - i * ldo;
    out[(size_t)i * ldo + j] = (j < n) ? Rrev[(size_t)i * ldr + (n - 1 - j)] : 0.0;
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


// ---- features entering the state, on the resident covariance: the tail of measurementUpdate_hybrid (src/orcvio.cpp:1818-1821,
//      :1904-1947) from the blocks k_ekf_new left on the device -------------------------------------------------------------
// HH = H_2^-1 H_1 (block back-substitution, H_2 upper triangular d x d per feature), column n: x = H_2^-1 r_1;
// W[j] = (H_2^T H_2)^-1 = R^-1 R^-T.  One thread per (feature, column).
__global__ __launch_bounds__(256) void k_aug_hh(const double* __restrict__ H1, const double* __restrict__ H2, const double* __restrict__ r1,
                                                int n, int n_new, int d, double* __restrict__ HH /* [d n_new][n + 1] */,
                                                double* __restrict__ W /* [n_new][d][d] */, int* __restrict__ singular, int diag_only = 0) {
    // diag_only: the reference's literal H_2.ldlt().solve(..) on an upper-triangular H_2 = division by its diagonal (src/orcvio.cpp:1826-1827)
    const int j = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_new || c > n) return;
    const double* R = H2 + (size_t)j * d * d;
    double v[3] = {0.0, 0.0, 0.0};
    for (int i = d - 1; i >= 0; --i) {
        double m = c < n ? H1[(size_t)(d * j + i) * n + c] : r1[d * j + i];
        if (!diag_only)
            for (int k = i + 1; k < d; ++k) m -= R[i * d + k] * v[k];
        v[i] = m / R[i * d + i];
        HH[(size_t)(d * j + i) * (n + 1) + c] = v[i];
    }
    if (c == 0) {
        double Ri[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        bool bad = false;
        for (int i = 0; i < d; ++i) bad = bad || R[i * d + i] == 0.0;
        if (bad) *singular = 1;
        for (int cc = 0; cc < d; ++cc)
            for (int i = d - 1; i >= 0; --i) {
                double m = i == cc ? 1.0 : 0.0;
                for (int k = i + 1; k < d; ++k) m -= R[i * d + k] * Ri[k * d + cc];
                Ri[i * d + cc] = m / R[i * d + i];
            }
        for (int i = 0; i < d; ++i)
            for (int cc = 0; cc < d; ++cc) {
                double m = 0.0;
                for (int k = 0; k < d; ++k) m += Ri[i * d + k] * Ri[cc * d + k];
                W[(size_t)j * d * d + i * d + cc] = m;
            }
    }
}
// dx_new[r] = x[r] - HH[r][:] dx: one wavefront per row
__global__ __launch_bounds__(64) void k_aug_dx(const double* __restrict__ HH, int n, const double* __restrict__ dx, double* __restrict__ dx_new) {
    const int r = blockIdx.x;
    const double* row = HH + (size_t)r * (n + 1);
    double s = 0.0;
    for (int c = threadIdx.x; c < n; c += 64) s += row[c] * dx[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) dx_new[r] = row[n] - s;
}
// P_aug [(n + sz)^2] in the order [old (n - tail) | new (sz) | tail]: P+ , nHHP = -HH P+ (sz x n), Q = nHHP HH^T (sz x sz, so that
// P22 = -Q + s2 W), symmetrised as :1946 does
__global__ __launch_bounds__(256) void k_aug_assemble(const double* __restrict__ Pp, int n, int sz, int tail, int d, const double* __restrict__ nHHP,
                                                      const double* __restrict__ Q, const double* __restrict__ W, double s2,
                                                      double* __restrict__ out) {
    const int nt = n + sz, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nt * nt) return;
    const int i = idx / nt, j = idx - i * nt, n0 = n - tail;
    auto src = [&](int q) { return q < n0 ? q : (q < n0 + sz ? n + (q - n0) : q - sz); };   // index in the [old + tail | new] order
    const int a = src(i), b = src(j);
    double v;
    if (a < n && b < n) v = 0.5 * (Pp[(size_t)a * n + b] + Pp[(size_t)b * n + a]);
    else if (a >= n && b < n) v = nHHP[(size_t)(a - n) * n + b];
    else if (a < n && b >= n) v = nHHP[(size_t)(b - n) * n + a];
    else {
        const int r = a - n, c = b - n;
        double p = -0.5 * (Q[(size_t)r * sz + c] + Q[(size_t)c * sz + r]);
        if (r / d == c / d) p += 0.5 * s2 * (W[(size_t)(r / d) * d * d + (r % d) * d + (c % d)] + W[(size_t)(r / d) * d * d + (c % d) * d + (r % d)]);
        v = p;
    }
    out[idx] = v;
}

}  // namespace orcvio_amd

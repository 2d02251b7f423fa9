// EKF-SLAM rows of the hybrid filter (existing SLAM features).
//
// The reference evaluates featureJacobian_ekf (src/orcvio.cpp:1575-1651) for every SLAM feature the current state
// observes, gates each with two degrees of freedom (gatingTestFeature(H_xj, r_j, 2), :2457) and stacks what passed
// under the MSCKF rows for one joint update (measurementUpdate_hybrid, :1766-1950).  Here the rows arrive in compact form
// (the four blocks of each row pair and their columns); k_ekf_gate gates them against the prior and writes the accepted
// ones as dense rows [H | r] over the active columns, whose Gram is ADDED to the compressed block
// (A = X^T X - T3^T T3 + E^T E): the joint update with the stacked rows, in the Gram form the solve already uses.
#pragma once
#include <hip/hip_runtime.h>
#include "msckf_math.hpp"

namespace orcvio_amd {

struct EkfGateArgs {
    int F, idp_dim, n, leg, N, NA, NAP, estimate_td, n_nui;
    const int* anchor; const int* state; const int* slot;
    const double* He; const double* Ha; const double* Hx; const double* Hf; const double* zvel; const double* r;
    const double* P;         // [n][n] prior
    double sigma2, threshold;   // chi-square quantile for 2 degrees of freedom
    double* E;               // [2F][NAP] dense rows, zero-filled before the launch
    double* gamma; int* accept;
};

// one wavefront per SLAM feature; lane t < 22 holds non-zero t of the row pair: column c_t and the two values
__device__ __forceinline__ void ekf_gate_body(const EkfGateArgs& p, const int f, const int t) {
    if (f >= p.F) return;
    const int d = p.idp_dim;
    const int a = p.anchor[f], k = p.state[f];
    int col = -1;
    double v0 = 0.0, v1 = 0.0;
    if (t < 6) {                       // H_e -> columns 15..20 (:1641)
        col = 15 + t; v0 = p.He[(size_t)f * 12 + t]; v1 = p.He[(size_t)f * 12 + 6 + t];
    } else if (t == 6) {               // observations_vel -> column 21 under estimate_td (:1642-1643)
        if (p.estimate_td) { col = 21; v0 = p.zvel[(size_t)f * 2]; v1 = p.zvel[(size_t)f * 2 + 1]; }
    } else if (t < 13) {               // H_a -> the anchor clone (:1639); the state clone's block overwrites it if they coincide
        const int e = t - 7;
        // (an anchor index >= N is a Schmidt nuisance state: its 6 columns are in the nuisance block at the end of the state, :1591-1606)
        if (a != k) { col = (a < p.N ? p.leg + 6 * a : p.n - 6 * p.n_nui + 6 * (a - p.N)) + e; v0 = p.Ha[(size_t)f * 12 + e]; v1 = p.Ha[(size_t)f * 12 + 6 + e]; }
    } else if (t < 19) {               // H_x -> the observing clone (:1640)
        const int e = t - 13;
        col = p.leg + 6 * k + e; v0 = p.Hx[(size_t)f * 12 + e]; v1 = p.Hx[(size_t)f * 12 + 6 + e];
    } else if (t < 19 + d) {           // H_f -> the feature's own state columns (:1630 / :1636)
        const int e = t - 19;
        col = p.leg + 6 * p.N + d * p.slot[f] + e; v0 = p.Hf[(size_t)f * 2 * d + e]; v1 = p.Hf[(size_t)f * 2 * d + d + e];
    }
    // w_b[t] = sum_u P[c_t][c_u] v_b[u]
    double w0 = 0.0, w1 = 0.0;
#pragma unroll
    for (int u = 0; u < 22; ++u) {   // (unrolled: the 22 loads of P in flight together; the sums keep their order)
        const int cu = __shfl(col, u);
        const double x0 = __shfl(v0, u), x1 = __shfl(v1, u);
        if (cu >= 0 && col >= 0) {
            const double pv = p.P[(size_t)col * p.n + cu];
            w0 += pv * x0;
            w1 += pv * x1;
        }
    }
    double s00 = v0 * w0, s01 = v0 * w1, s11 = v1 * w1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s00 += __shfl_xor(s00, o); s01 += __shfl_xor(s01, o); s11 += __shfl_xor(s11, o); }
    s00 += p.sigma2; s11 += p.sigma2;
    const double r0 = p.r[(size_t)f * 2], r1 = p.r[(size_t)f * 2 + 1];
    const double det = s00 * s11 - s01 * s01;
    const double g = (r0 * (s11 * r0 - s01 * r1) + r1 * (s00 * r1 - s01 * r0)) / det;   // r^T S^-1 r  (:1953-1976)
    const bool ok = g < p.threshold;   // NaN -> rejected
    if (t == 0) { p.gamma[f] = g; p.accept[f] = ok ? 1 : 0; }
    if (!ok) return;
    if (col >= 15) {
        p.E[(size_t)(2 * f) * p.NAP + col - 15] = v0;
        p.E[(size_t)(2 * f + 1) * p.NAP + col - 15] = v1;
    }
    if (t == 22) {
        p.E[(size_t)(2 * f) * p.NAP + p.NA] = r0;
        p.E[(size_t)(2 * f + 1) * p.NAP + p.NA] = r1;
    }
}
__global__ __launch_bounds__(64) void k_ekf_gate(EkfGateArgs p) { ekf_gate_body(p, blockIdx.x, threadIdx.x); }


// ---- the four blocks themselves, from the SLAM features (measurementJacobian_ekf_3didp :1229-1353, _1didp :1356-1478) ----
struct EkfEvalArgs {
    int F, idp_dim, if_fej;
    const double* poses;     // [N][POSE_STRIDE]
    const int* anchor; const int* state;
    const double* param;     // [F][3]  3-d: invParam (x/z, y/z, 1/z in the anchor camera); 1-d: obs_anchor (u, v, 1)
    const double* inv_depth; // [F]     1-d: invDepth
    const double* p_w;       // [F][3]  Feature::position
    const double* p_fej;     // [F][3]  Feature::position_FEJ (if_fej)
    const double* z;         // [F][2]  the observation in the current state
    double* He; double* Ha; double* Hx; double* Hf; double* r;   // outputs, compact (as orcvio_msckf_ekf_rows)
};

namespace ekfm {
ORC_HD void mat3_mul(const double* A, const double* B, double* C) {   // C = A B
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
ORC_HD void mat3_mul_bt(const double* A, const double* B, double* C) {   // C = A B^T
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}
ORC_HD void mat3_tmul_vec(const double* A, const double* x, double* y) {   // y = A^T x
    for (int i = 0; i < 3; ++i) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}
ORC_HD void mat3_vec(const double* A, const double* x, double* y) {   // y = A x
    for (int i = 0; i < 3; ++i) y[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
}
ORC_HD void skew(const double* w, double* S) {
    S[0] = 0; S[1] = -w[2]; S[2] = w[1]; S[3] = w[2]; S[4] = 0; S[5] = -w[0]; S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}
// out (2 x 3) = J_k (2 x 3) M (3 x 3)
ORC_HD void jk_mul(const double* Jk, const double* M, double* out) {
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) out[i * 3 + j] = Jk[i * 3] * M[j] + Jk[i * 3 + 1] * M[3 + j] + Jk[i * 3 + 2] * M[6 + j];
}
}  // namespace ekfm

// measurementJacobian_ekf_3didp (:1229-1353) / _1didp (:1356-1478) for ONE (feature, observing state) pair: the four blocks
// and the residual.  Pk / Pa: pose records of the observing and of the anchor clone; same = (state == anchor).
ORC_HD void ekf_row_blocks(const double* Pk, const double* Pa, bool same, int d, int if_fej, const double* prm, double inv_depth,
                           const double* pw, const double* pfej, const double* z, double* He, double* Ha, double* Hx, double* Hf,
                           double* r) {
    using namespace ekfm;
    const int k = same ? 0 : 1, a = 0;   // (only their equality matters below)
    const double* R_b2c = Pk + POSE_R_B2C;
    const double* t_c_b = Pk + POSE_T_C_B;
    const double* R_bk2w = Pk + POSE_R_B2W;
    const double* R_ba2w = Pa + POSE_R_B2W;
    double R_w2ck[9], R_w2ca[9];
    mat3_mul_bt(R_b2c, R_bk2w, R_w2ck);     // R_b2c R_w2bk   (:1264)
    mat3_mul_bt(R_b2c, R_ba2w, R_w2ca);     // (:1278)
    double t_ck_w[3], tmp[3];
    mat3_vec(R_bk2w, t_c_b, tmp);
    for (int i = 0; i < 3; ++i) t_ck_w[i] = Pk[POSE_T_B_W + i] + tmp[i];
    const double* pf = if_fej ? pfej : pw;
    const double rho = d == 3 ? prm[2] : inv_depth;
    double p_ca[3];
    if (if_fej) {                          // :1281-1282
        double dv[3], q[3];
        for (int i = 0; i < 3; ++i) dv[i] = pf[i] - Pa[POSE_T_FEJ + i];
        mat3_tmul_vec(R_ba2w, dv, q);
        for (int i = 0; i < 3; ++i) q[i] -= t_c_b[i];
        mat3_vec(R_b2c, q, p_ca);
    } else {
        p_ca[0] = prm[0] / rho; p_ca[1] = prm[1] / rho; p_ca[2] = 1.0 / rho;
    }
    double dv[3], p_ck[3];
    for (int i = 0; i < 3; ++i) dv[i] = pw[i] - t_ck_w[i];
    mat3_vec(R_w2ck, dv, p_ck);
    double r0 = z[0] - p_ck[0] / p_ck[2], r1 = z[1] - p_ck[1] / p_ck[2];   // :1299
    for (int i = 0; i < 12; ++i) { He[i] = 0.0; Ha[i] = 0.0; Hx[i] = 0.0; }
    for (int i = 0; i < 6; ++i) Hf[i] = 0.0;
    if (k == a) {                            // :1302-1310 / :1432-1440
        if (d == 3) { Hf[0] = 1.0; Hf[d + 1] = 1.0; }
        else { r0 = 0.0; r1 = 0.0; }
    } else {
        double Jk[6] = {1.0 / p_ck[2], 0.0, -p_ck[0] / (p_ck[2] * p_ck[2]), 0.0, 1.0 / p_ck[2], -p_ck[1] / (p_ck[2] * p_ck[2])};
        double p_baf[3], p_bkf[3];
        for (int i = 0; i < 3; ++i) {
            p_baf[i] = if_fej ? pf[i] - Pa[POSE_T_FEJ + i] : pw[i] - Pa[POSE_T_B_W + i];   // :1320-1323
            p_bkf[i] = if_fej ? pf[i] - Pk[POSE_T_FEJ + i] : pw[i] - Pk[POSE_T_B_W + i];
        }
        double S[9], M[9], J[6];
        skew(p_baf, S); mat3_mul(R_w2ck, S, M);
        jk_mul(Jk, M, J);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) Ha[i * 6 + j] = -J[i * 3 + j];       // -R_w2ck [p_baf]x
        jk_mul(Jk, R_w2ck, J);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) { Ha[i * 6 + 3 + j] = J[i * 3 + j]; Hx[i * 6 + 3 + j] = -J[i * 3 + j]; }
        skew(p_bkf, S); mat3_mul(R_w2ck, S, M);
        jk_mul(Jk, M, J);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) Hx[i * 6 + j] = J[i * 3 + j];        //  R_w2ck [p_bkf]x
        // J_e (:1333-1337)
        double q[3], u[3], Rka[9], SkM[9], Mx[9], T[9];
        mat3_tmul_vec(R_bk2w, p_bkf, q);
        for (int i = 0; i < 3; ++i) q[i] -= t_c_b[i];
        skew(q, SkM);
        mat3_tmul_vec(R_b2c, p_ca, u);
        skew(u, S);
        for (int i = 0; i < 3; ++i)          // Rka = R_w2bk R_ba2w = R_bk2w^T R_ba2w
            for (int j = 0; j < 3; ++j) Rka[i * 3 + j] = R_bk2w[i] * R_ba2w[j] + R_bk2w[3 + i] * R_ba2w[3 + j] + R_bk2w[6 + i] * R_ba2w[6 + j];
        mat3_mul(Rka, S, Mx);
        for (int i = 0; i < 9; ++i) T[i] = SkM[i] - Mx[i];
        mat3_mul(R_b2c, T, M);
        jk_mul(Jk, M, J);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) He[i * 6 + j] = J[i * 3 + j];
        for (int i = 0; i < 9; ++i) T[i] = Rka[i] - ((i % 4 == 0) ? 1.0 : 0.0);
        mat3_mul(R_b2c, T, M);
        jk_mul(Jk, M, J);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) He[i * 6 + 3 + j] = J[i * 3 + j];
        // H_f
        double Jp[9];
        mat3_mul_bt(R_w2ck, R_w2ca, Jp);     // :1318
        if (d == 3) {                        // :1339-1345
            const double Jf[9] = {1.0 / rho, 0.0, -prm[0] / rho / rho, 0.0, 1.0 / rho, -prm[1] / rho / rho, 0.0, 0.0, -1.0 / rho / rho};
            mat3_mul(Jp, Jf, M);
            jk_mul(Jk, M, J);
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) Hf[i * 3 + j] = J[i * 3 + j];
        } else {                             // :1447, :1469-1471
            double Jd[3];
            mat3_vec(Jp, prm, Jd);
            const double Jrho = -1.0 / (rho * rho);
            for (int i = 0; i < 2; ++i) Hf[i] = (Jk[i * 3] * Jd[0] + Jk[i * 3 + 1] * Jd[1] + Jk[i * 3 + 2] * Jd[2]) * Jrho;
        }
    }
    r[0] = r0; r[1] = r1;
}

// one thread per SLAM feature (a handful to a few dozen per frame)
__device__ __forceinline__ void ekf_eval_body(const EkfEvalArgs& p, const int f) {
    if (f >= p.F) return;
    const int d = p.idp_dim, a = p.anchor[f], k = p.state[f];
    double He[12], Ha[12], Hx[12], Hf[6], r[2];
    ekf_row_blocks(p.poses + (size_t)k * POSE_STRIDE, p.poses + (size_t)a * POSE_STRIDE, k == a, d, p.if_fej, p.param + (size_t)3 * f,
                   d == 1 ? p.inv_depth[f] : 0.0, p.p_w + (size_t)3 * f, p.p_fej ? p.p_fej + (size_t)3 * f : nullptr, p.z + (size_t)2 * f,
                   He, Ha, Hx, Hf, r);
    for (int i = 0; i < 12; ++i) { p.He[(size_t)f * 12 + i] = He[i]; p.Ha[(size_t)f * 12 + i] = Ha[i]; p.Hx[(size_t)f * 12 + i] = Hx[i]; }
    for (int i = 0; i < 2 * d; ++i) p.Hf[(size_t)f * 2 * d + i] = Hf[i];
    p.r[2 * f] = r[0]; p.r[2 * f + 1] = r[1];
}
__global__ __launch_bounds__(64) void k_ekf_eval(EkfEvalArgs p) { ekf_eval_body(p, blockIdx.x * 64 + threadIdx.x); }

// ---- features ENTERING the state: featureJacobian_ekf_new (src/orcvio.cpp:1481-1572) and the W = [V | U] split (:2416-2436) ----
// One workgroup per new feature.  Its 2 M rows [H_x | r] are written dense over the active columns (rows of d_dense, the
// rows the caller-projected path uses); H_f (2M x d) stays in LDS.  H_f of different features share no column, so the split
// is one Householder QR of H_f per feature applied to its own rows (U = the first d columns of Q, V the rest; only the two
// subspaces matter): the first d rows are the U part -- H_1, r_1, and H_2 = the R factor -- and leave the stack (zeroed);
// the other 2M - d rows are the V part, zero in the new columns, and are stacked under everything else as they are.
struct EkfNewArgs {
    int n_new, idp_dim, if_fej, estimate_td, leg, NA, NAP, n, N, n_nui;
    const double* poses;
    const int* anchor;          // [n_new]
    const double* param;        // [n_new][3]
    const double* inv_depth;    // [n_new] (idp 1)
    const double* p_w;          // [n_new][3]
    const double* p_fej;        // [n_new][3] (if_fej)
    const int* obs_ptr; const int* obs_clone; const double* obs_z; const double* obs_zvel;
    const int* row0;            // [n_new + 1] first row of every feature in `dense`
    double* dense;              // [rows][NAP]: H(:, 15 : 15 + NA) | r | 0
    double* H1; double* H2; double* r1;   // [d n_new][n], [n_new][d][d], [d n_new]
};
#define EKF_NEW_MAXROWS 64   /* 2 x ORCVIO_MAX_TRACK */
__global__ __launch_bounds__(256) void k_ekf_new(EkfNewArgs p) {
    __shared__ double sHf[EKF_NEW_MAXROWS][3];
    __shared__ double sv[EKF_NEW_MAXROWS];
    __shared__ double sRed[4];
    __shared__ int sSkip;
    const int j = blockIdx.x, tid = threadIdx.x, d = p.idp_dim;
    const int o0 = p.obs_ptr[j], nobs = p.obs_ptr[j + 1] - o0, a = p.anchor[j];
    const int r0 = p.row0[j], m = p.row0[j + 1] - r0;   // m = 2 x kept observations
    double* M = p.dense + (size_t)r0 * p.NAP;
    if (tid == 0) sSkip = -1;
    for (int i = tid; i < m * p.NAP; i += 256) M[i] = 0.0;
    __syncthreads();
    if (d == 1 && tid < nobs && p.obs_clone[o0 + tid] == a) sSkip = tid;   // the anchor's own observation is not used (:1494-1496)
    __syncthreads();
    if (tid < nobs && tid != sSkip) {
        const int c = tid - ((sSkip >= 0 && tid > sSkip) ? 1 : 0), o = o0 + tid, k = p.obs_clone[o];
        double He[12], Ha[12], Hx[12], Hf[6], rr[2];
        ekf_row_blocks(p.poses + (size_t)k * POSE_STRIDE, p.poses + (size_t)a * POSE_STRIDE, k == a, d, p.if_fej, p.param + (size_t)3 * j,
                       d == 1 ? p.inv_depth[j] : 0.0, p.p_w + (size_t)3 * j, p.p_fej ? p.p_fej + (size_t)3 * j : nullptr, p.obs_z + (size_t)2 * o,
                       He, Ha, Hx, Hf, rr);
        for (int b = 0; b < 2; ++b) {
            double* row = M + (size_t)(2 * c + b) * p.NAP;
            const int ca = (a < p.N ? p.leg + 6 * a : p.n - 6 * p.n_nui + 6 * (a - p.N)) - 15;   // (a >= N: a Schmidt nuisance state)
            for (int e = 0; e < 6; ++e) row[ca + e] = Ha[6 * b + e];                  // :1561
            for (int e = 0; e < 6; ++e) row[p.leg - 15 + 6 * k + e] = Hx[6 * b + e];   // :1562 (overwrites if k == a)
            for (int e = 0; e < 6; ++e) row[e] = He[6 * b + e];                       // :1563 (columns 15..20)
            if (p.estimate_td) row[6] = p.obs_zvel[(size_t)2 * o + b];                // :1564-1565 (column 21)
            row[p.NA] = rr[b];
            for (int e = 0; e < 3; ++e) sHf[2 * c + b][e] = e < d ? Hf[b * d + e] : 0.0;
        }
    }
    __syncthreads();
    // Householder QR of H_f (m x d, dgeqr2 convention), applied to [H_x | r]: one thread per column
    for (int q = 0; q < d; ++q) {
        if (tid < 64) {
            double part = (tid > q && tid < m) ? sHf[tid][q] * sHf[tid][q] : 0.0;
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (tid == 0) sRed[0] = part;
        }
        __syncthreads();
        const double nrm2 = sRed[0], alpha = sHf[q][q];
        if (nrm2 != 0.0) {   // (block-uniform)
            const double nu = sqrt(alpha * alpha + nrm2), bk = alpha >= 0.0 ? -nu : nu;
            const double beta = (bk - alpha) / bk, sc = 1.0 / (alpha - bk);
            if (tid < m) sv[tid] = tid < q ? 0.0 : (tid == q ? 1.0 : sHf[tid][q] * sc);
            __syncthreads();
            for (int c = tid; c < p.NAP + 3; c += 256) {
                if (c < p.NAP) {
                    if (c > p.NA) continue;   // (padding columns stay zero)
                    double w = 0.0;
                    for (int i = q; i < m; ++i) w += sv[i] * M[(size_t)i * p.NAP + c];
                    w *= beta;
                    for (int i = q; i < m; ++i) M[(size_t)i * p.NAP + c] -= w * sv[i];
                } else {
                    const int e = c - p.NAP;   // columns of H_f
                    if (e < q || e >= d) continue;
                    double w = 0.0;
                    for (int i = q; i < m; ++i) w += sv[i] * sHf[i][e];
                    w *= beta;
                    for (int i = q; i < m; ++i) sHf[i][e] -= w * sv[i];
                }
            }
        }
        __syncthreads();
    }
    // the U part leaves the stack
    for (int i = 0; i < d; ++i) {
        double* h1 = p.H1 + (size_t)(d * j + i) * p.n;
        for (int c = tid; c < p.n; c += 256) h1[c] = (c >= 15 && c < 15 + p.NA) ? M[(size_t)i * p.NAP + c - 15] : 0.0;
        if (tid == 0) p.r1[d * j + i] = M[(size_t)i * p.NAP + p.NA];
        if (tid < d) p.H2[(size_t)j * d * d + i * d + tid] = tid >= i ? sHf[i][tid] : 0.0;
    }
    __syncthreads();
    for (int i = tid; i < d * p.NAP; i += 256) M[i] = 0.0;
}

__global__ __launch_bounds__(256) void k_add_inplace(double* __restrict__ dst, const double* __restrict__ src, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

}  // namespace orcvio_amd

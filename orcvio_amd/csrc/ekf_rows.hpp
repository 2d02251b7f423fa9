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

namespace orcvio_amd {

struct EkfGateArgs {
    int F, idp_dim, n, leg, N, NA, NAP, estimate_td;
    const int* anchor; const int* state; const int* slot;
    const double* He; const double* Ha; const double* Hx; const double* Hf; const double* zvel; const double* r;
    const double* P;         // [n][n] prior
    double sigma2, threshold;   // chi-square quantile for 2 degrees of freedom
    double* E;               // [2F][NAP] dense rows, zero-filled before the launch
    double* gamma; int* accept;
};

// one wavefront per SLAM feature; lane t < 22 holds non-zero t of the row pair: column c_t and the two values
__global__ __launch_bounds__(64) void k_ekf_gate(EkfGateArgs p) {
    const int f = blockIdx.x, t = threadIdx.x;
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
        if (a != k) { col = p.leg + 6 * a + e; v0 = p.Ha[(size_t)f * 12 + e]; v1 = p.Ha[(size_t)f * 12 + 6 + e]; }
    } else if (t < 19) {               // H_x -> the observing clone (:1640)
        const int e = t - 13;
        col = p.leg + 6 * k + e; v0 = p.Hx[(size_t)f * 12 + e]; v1 = p.Hx[(size_t)f * 12 + 6 + e];
    } else if (t < 19 + d) {           // H_f -> the feature's own state columns (:1630 / :1636)
        const int e = t - 19;
        col = p.leg + 6 * p.N + d * p.slot[f] + e; v0 = p.Hf[(size_t)f * 2 * d + e]; v1 = p.Hf[(size_t)f * 2 * d + d + e];
    }
    // w_b[t] = sum_u P[c_t][c_u] v_b[u]
    double w0 = 0.0, w1 = 0.0;
    for (int u = 0; u < 22; ++u) {
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

}  // namespace orcvio_amd

// feature_split.hpp -- the tracks' front end for MANY tracks per GPU (more than the fused k_front holds co-resident: F > 2 (CUs - 1)).
//
// k_feature keeps one four-wavefront team per track and is bound by how many teams a CU can hold: the gate wavefront carries the
// 16 x 16 tiles of the 2M x 2M gate matrix in registers (223 VGPRs: two wavefronts per SIMD) and E = J P J^T sits in LDS (57 KB per
// track at 30 observations): two tracks per CU, 512 in flight, ~25 us per round -- 2 000 tracks are four rounds, ~100 us
// (profiles/r2*_configs.json; raising the occupancy of that kernel by launch bounds spills 116 / 286 VGPRs).  Here the
// two register- and LDS-hungry roles live in two kernels, each with the occupancy its own role allows:
//
//   k_feature_e     one workgroup (four wavefronts) per track, FOUR workgroups per CU (<= 128 VGPRs, ~24 KB of LDS): the
//                   per-observation Jacobians (wave 0, measurementJacobian_msckf src/orcvio.cpp:1071-1168), the three-reflector
//                   Householder QR of H_f (wave 0), E = J P J^T from the 13 non-zeros of every row (all four wavefronts,
//                   observations handed out through an LDS counter; no software prefetch: sixteen wavefronts per CU hide the
//                   latency instead), written to HBM (upper triangle, by columns: coalesced); then the outputs as for an accepted
//                   track (T3 = rows 0..2 of Q^T [J | r], the sparse rows Xobs)
//   k_feature_gate  one WAVEFRONT per track, eight per CU: the tiles of E + s2 I come back in ONE round trip, the left-looking tile
//                   Cholesky, the right-hand sides [r | Q1] and the projection run from registers exactly as phase G of
//                   feature_body does (gatingTestFeature :1953-1976 through the generalised-least-squares identity); a rejected
//                   track zeroes the outputs k_feature_e wrote for it.
// Same arithmetic, same order of operations as feature_body: the results are the same bits.
#pragma once
#include "msckf_kernels.hpp"

namespace orcvio_amd {

__host__ __device__ inline size_t feat_split_lds_bytes(int Mmax, int NAP, int N) {
    const int R2 = 2 * Mmax;
    const size_t dbl = (size_t)R2 * 7 + (size_t)R2 * 6 + R2 + (size_t)R2 * 4 + 16 + 8 * (size_t)NAP;
    const size_t bytes = dbl * 8 + (size_t)(N + Mmax + 4 + Mmax) * 4;
    return (bytes + 15) & ~(size_t)15;
}
// doubles of scratch per track: E by columns [2 Mmax][2 Mmax + 1] (rows <= column + 1 of every column are written), then the
// right-hand sides of the gate [64][4]
__host__ __device__ inline size_t feat_split_track_doubles(int Mmax) { return (size_t)2 * Mmax * feat_lde(Mmax) + 256; }

template <int NPASS>
__global__ __launch_bounds__(256, 4) void k_feature_e(FeatArgs p, double* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NPD = (NPASS + 3) / 4;   // passes of the 256 threads over the NAP columns
    const int j = blockIdx.x, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), t = tid & 63;
    const int lo = p.obs_ptr[j];
    const int M = p.obs_ptr[j + 1] - lo;
    if (M < 2 || (p.skip && p.skip[j])) {   // whole workgroup (k_feature_gate skips the track on the same test)
        if (tid == 0) { p.gamma[j] = NAN; p.accept[j] = 0; }
        for (int e = tid; e < 3 * p.NAP; e += 256) p.T3[(size_t)3 * j * p.NAP + e] = 0.0;
        for (int e = tid; e < 32 * M; e += 256) p.Xobs[(size_t)32 * p.obs_pos[lo + (e >> 5)] + (e & 31)] = 0.0;
        return;
    }
    const int M2 = 2 * M;
    const int R2 = 2 * p.Mmax;
    const int LDE = feat_lde(p.Mmax);
    const int NA = p.NA, NAP = p.NAP, n = p.n;
    const int cb0 = p.leg - 15;
    const int NAc = cb0 + 6 * p.N;
    double* sJe = smem;               // [R2][7]  ext(6)+td
    double* sJx = sJe + R2 * 7;       // [R2][6]
    double* sR = sJx + R2 * 6;        // [R2]
    double* sV = sR + R2;             // [R2][4]  Householder vectors
    double* sQ = sV + R2 * 4;         // [16]     beta(3), g10, g20, g21
    double* sUall = sQ + 16;          // [4 waves][NAP][2]
    int* sC2O = (int*)(sUall + 8 * (size_t)NAP);   // [N]
    int* sOC = sC2O + p.N;                          // [Mmax]
    int* sFlag = sOC + p.Mmax;                      // [4]  [1] next observation of phase E
    int* sLim = sFlag + 4;                          // [Mmax]
    double* Eg = scratch + (size_t)j * feat_split_track_doubles(p.Mmax);   // E(row, col) at Eg[col * LDE + row]
    double* Bg = Eg + (size_t)R2 * LDE;                                    // [64][4] right-hand sides of the gate

    for (int i = tid; i < p.N; i += 256) sC2O[i] = -1;
    if (tid == 0) sFlag[1] = 0;
    __syncthreads();

    // ---- B: per-observation Jacobians (wave 0, lane t <-> observation t) -----------------
    double a0[3] = {0, 0, 0}, a1[3] = {0, 0, 0};   // rows 2t, 2t+1 of H_f
    if (wave == 0) {
        if (t < M) {
            const int o = lo + t;
            const int ci = p.obs_clone[o];
            double Hx[12], He[12], Hf[6], rr[2];
            double pw[3] = {p.p_w[3 * j], p.p_w[3 * j + 1], p.p_w[3 * j + 2]};
            double z[2] = {p.obs_z[2 * o], p.obs_z[2 * o + 1]};
            ObsFlags f{p.use_larvio, p.use_left, p.if_fej};
            obs_jacobian(p.poses + (size_t)ci * POSE_STRIDE, pw, z, f, Hx, He, Hf, rr);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int row = 2 * t + s;
#pragma unroll
                for (int e = 0; e < 6; ++e) sJe[row * 7 + e] = He[s * 6 + e];
                sJe[row * 7 + 6] = p.estimate_td ? p.obs_zvel[2 * o + s] : 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) sJx[row * 6 + c] = Hx[s * 6 + c];
                sR[row] = rr[s];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { a0[c] = Hf[c]; a1[c] = Hf[3 + c]; }
            sOC[t] = ci;
            sC2O[ci] = t;
        }
        {   // columns of P observation l needs: the clones of observations <= l (upper triangle of E only), feature_body phase B
            int mx = (t < M) ? p.obs_clone[lo + t] : -1;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { const int v = __shfl_up(mx, o); if (t >= o) mx = v > mx ? v : mx; }
            if (t < M) sLim[t] = cb0 + 6 * mx + 6;
        }
    }
    __syncthreads();   // the Jacobians are all phase E needs

    if (wave == 0) {
        // ---- C: Householder QR of H_f (2M x 3), LAPACK dgeqr2 convention (feature_body phase C) ------------------
        const int g0 = 2 * t, g1 = 2 * t + 1;
        double v0[3], v1[3], beta[3], rdiag[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double s = 0.0;
            if (g0 > q) s += a0[q] * a0[q];
            if (g1 > q) s += a1[q] * a1[q];
            const double nrm2 = wave_sum(s);
            const double alpha = __shfl((q & 1) ? a1[q] : a0[q], q >> 1);
            double bq = 0.0, sc = 0.0;
            rdiag[q] = alpha;
            if (nrm2 != 0.0) {
                const double nu = sqrt(alpha * alpha + nrm2);
                const double bk = (alpha >= 0.0) ? -nu : nu;
                bq = (bk - alpha) / bk;
                sc = 1.0 / (alpha - bk);
                rdiag[q] = bk;
            }
            beta[q] = bq;
            v0[q] = (g0 > q) ? a0[q] * sc : ((g0 == q) ? 1.0 : 0.0);
            v1[q] = (g1 > q) ? a1[q] * sc : ((g1 == q) ? 1.0 : 0.0);
#pragma unroll
            for (int c = q + 1; c < 3; ++c) {
                const double w = wave_sum(v0[q] * a0[c] + v1[q] * a1[c]) * bq;
                a0[c] -= w * v0[q];
                a1[c] -= w * v1[q];
            }
        }
        if (t < M) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { sV[g0 * 4 + q] = v0[q]; sV[g1 * 4 + q] = v1[q]; }
            sV[g0 * 4 + 3] = 0.0;
            sV[g1 * 4 + 3] = 0.0;
        }
        const double g10 = wave_sum(v0[1] * v0[0] + v1[1] * v1[0]);
        const double g20 = wave_sum(v0[2] * v0[0] + v1[2] * v1[0]);
        const double g21 = wave_sum(v0[2] * v0[1] + v1[2] * v1[1]);
        if (t == 0) { sQ[0] = beta[0]; sQ[1] = beta[1]; sQ[2] = beta[2]; sQ[3] = g10; sQ[4] = g20; sQ[5] = g21; }
        if (p.Rf && t == 0) {
            double* rf = p.Rf + (size_t)6 * j;
            rf[0] = rdiag[0]; rf[1] = a0[1]; rf[2] = a0[2]; rf[3] = rdiag[1]; rf[4] = a1[2]; rf[5] = rdiag[2];
        }
        wave_sync();
        // right-hand sides of the gate, row t: [r | Q1] (feature_body): to HBM for k_feature_gate
        {
            double b4[4] = {0, 0, 0, 0};
            if (t < M2) {
                b4[0] = sR[t];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double w0 = sV[q * 4 + 0], w1 = sV[q * 4 + 1], w2 = sV[q * 4 + 2];
                    const double z2 = beta[2] * w2;
                    const double z1 = beta[1] * (w1 - g21 * z2);
                    const double z0 = beta[0] * (w0 - g10 * z1 - g20 * z2);
                    b4[1 + q] = ((t == q) ? 1.0 : 0.0) - (sV[t * 4 + 0] * z0 + sV[t * 4 + 1] * z1 + sV[t * 4 + 2] * z2);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) Bg[t * 4 + q] = b4[q];
        }
    }

    // ---- E = J P_aa J^T: all four wavefronts (wave 0 joins behind the QR); one observation at a time per wavefront --------
    {
        double* sU = sUall + (size_t)wave * 2 * NAP;
        const bool rowlane = t < M2;
        double je[7], jx[6];
        int ja0 = 0;
#pragma unroll
        for (int e = 0; e < 7; ++e) je[e] = rowlane ? sJe[t * 7 + e] : 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) jx[c] = rowlane ? sJx[t * 6 + c] : 0.0;
        if (rowlane) ja0 = cb0 + 6 * sOC[t >> 1];
        // extrinsic / td rows of P: zero in every shipped configuration (src/orcvio.cpp:213-221) -- tested once, re-read when live
        bool pe_nz = false;
#pragma unroll 1
        for (int ps = 0; ps < NPASS; ++ps) {   // (rolled: this kernel trades instruction-level parallelism for wavefronts per SIMD)
            const int a = t + 64 * ps;
#pragma unroll
            for (int e = 0; e < 7; ++e) pe_nz |= (p.P[(size_t)(15 + e) * n + 15 + (a < NA ? a : NA - 1)] != 0.0);
        }
        const bool ext_live = __any(pe_nz);
        const int lend = M;
        for (;;) {
            int l = 0;
            if (t == 0) l = atomicAdd(&sFlag[1], 1);
            l = __builtin_amdgcn_readfirstlane(l);
            if (l >= lend) break;
            int lim = __builtin_amdgcn_readfirstlane(sLim[l]);
            lim = lim < NA ? lim : NA;
            const double* Prow = p.P + (size_t)(p.leg + 6 * sOC[l]) * n + 15;
            double jl0e[7], jl1e[7], jl0x[6], jl1x[6];
#pragma unroll
            for (int e = 0; e < 7; ++e) { jl0e[e] = sJe[(2 * l) * 7 + e]; jl1e[e] = sJe[(2 * l + 1) * 7 + e]; }
#pragma unroll
            for (int c = 0; c < 6; ++c) { jl0x[c] = sJx[(2 * l) * 6 + c]; jl1x[c] = sJx[(2 * l + 1) * 6 + c]; }
#pragma unroll 1
            for (int ps = 0; ps < NPASS; ++ps) {
                const int a = t + 64 * ps;
                if (64 * ps < lim) {   // (wave-uniform)
                    const int ac = a < NA ? a : NA - 1;
                    double pc6[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) pc6[c] = Prow[(size_t)c * n + ac];
                    double u0 = 0.0, u1 = 0.0;
                    if (ext_live) {
#pragma unroll
                        for (int e = 0; e < 7; ++e) {
                            const double pv = p.P[(size_t)(15 + e) * n + 15 + ac];
                            u0 += jl0e[e] * pv; u1 += jl1e[e] * pv;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        u0 += jl0x[c] * pc6[c];
                        u1 += jl1x[c] * pc6[c];
                    }
                    if (a < lim) { sU[2 * a] = u0; sU[2 * a + 1] = u1; }
                }
            }
            wave_sync();
            if (rowlane && t <= 2 * l + 1) {   // rows <= columns 2l, 2l+1: the upper triangle, stored by columns
                double e0 = 0.0, e1 = 0.0;
                if (ext_live) {
#pragma unroll
                    for (int e = 0; e < 7; ++e) { e0 += je[e] * sU[2 * e]; e1 += je[e] * sU[2 * e + 1]; }
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) { e0 += jx[c] * sU[2 * (ja0 + c)]; e1 += jx[c] * sU[2 * (ja0 + c) + 1]; }
                Eg[(size_t)(2 * l) * LDE + t] = e0;
                Eg[(size_t)(2 * l + 1) * LDE + t] = e1;
            }
            wave_sync();
        }
    }
    __syncthreads();   // (sV / sQ of the QR are final as well: wave 0 passed through here)

    // ---- D: compact-WY coefficients of Q^T [J | r] for the columns this thread owns; I: the outputs, as for an accepted track --
    double yq[NPD][3];
    {
        const double be0 = sQ[0], be1 = sQ[1], be2 = sQ[2], g10 = sQ[3], g20 = sQ[4], g21 = sQ[5];
#pragma unroll
        for (int ps = 0; ps < NPD; ++ps) {
            const int a = tid + 256 * ps;
            double w0 = 0.0, w1 = 0.0, w2 = 0.0;
            if (a < 7 || a == NA) {
                for (int i = 0; i < M2; ++i) {
                    const double val = (a < 7) ? sJe[i * 7 + a] : sR[i];
                    w0 += sV[i * 4 + 0] * val;
                    w1 += sV[i * 4 + 1] * val;
                    w2 += sV[i * 4 + 2] * val;
                }
            } else if (a >= cb0 && a < NAc) {
                const int cl = (a - cb0) / 6, cc = (a - cb0) - 6 * cl;
                const int k = sC2O[cl];
                if (k >= 0) {
                    const double x0 = sJx[(2 * k) * 6 + cc], x1 = sJx[(2 * k + 1) * 6 + cc];
                    w0 = sV[(2 * k) * 4 + 0] * x0 + sV[(2 * k + 1) * 4 + 0] * x1;
                    w1 = sV[(2 * k) * 4 + 1] * x0 + sV[(2 * k + 1) * 4 + 1] * x1;
                    w2 = sV[(2 * k) * 4 + 2] * x0 + sV[(2 * k + 1) * 4 + 2] * x1;
                }
            }
            const double y0 = be0 * w0;
            const double y1 = be1 * (w1 - g10 * y0);
            const double y2 = be2 * (w2 - g20 * y0 - g21 * y1);
            yq[ps][0] = y0; yq[ps][1] = y1; yq[ps][2] = y2;
        }
    }
#pragma unroll
    for (int ps = 0; ps < NPD; ++ps) {
        const int a = tid + 256 * ps;
        if (a < NAP) {
            int kobs = -1, cc = 0;
            if (a >= cb0 && a < NAc) {
                const int cl = (a - cb0) / 6;
                cc = (a - cb0) - 6 * cl;
                kobs = sC2O[cl];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double val = 0.0;
                if (a <= NA) {
                    double jv = 0.0;
                    if (a < 7) jv = sJe[i * 7 + a];
                    else if (a == NA) jv = sR[i];
                    else if (kobs == (i >> 1)) jv = sJx[i * 6 + cc];
                    val = jv - (sV[i * 4 + 0] * yq[ps][0] + sV[i * 4 + 1] * yq[ps][1] + sV[i * 4 + 2] * yq[ps][2]);
                }
                p.T3[((size_t)3 * j + i) * NAP + a] = val;
            }
        }
    }
    for (int e = tid; e < 16 * M2; e += 256) {
        const int row = e >> 4, c = e & 15;
        double v = 0.0;
        if (c < 14) v = (c < 7) ? sJe[row * 7 + c] : ((c < 13) ? sJx[row * 6 + (c - 7)] : sR[row]);
        p.Xobs[(size_t)32 * p.obs_pos[lo + (row >> 1)] + 16 * (row & 1) + c] = v;
    }
}

// One wavefront per track: phase G of feature_body on E and the right-hand sides k_feature_e left in HBM.
__global__ __launch_bounds__(256, 2) void k_feature_gate(FeatArgs p, const double* __restrict__ scratch) {
    __shared__ double sTile[4][2][272];   // per wavefront: the diagonal tile (row view, 17-double rows) and its inverse factor
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), t = threadIdx.x & 63;
    const int j = 4 * (int)blockIdx.x + wave;
    if (j >= p.F) return;
    const int lo = p.obs_ptr[j];
    const int M = p.obs_ptr[j + 1] - lo;
    if (M < 2 || (p.skip && p.skip[j])) return;   // (k_feature_e wrote gamma = NaN, accept = 0, zero outputs)
    const int M2 = 2 * M;
    const int LDE = feat_lde(p.Mmax);
    const double* Eg = scratch + (size_t)j * feat_split_track_doubles(p.Mmax);
    const double* Bg = Eg + (size_t)2 * p.Mmax * LDE;
    double* sD = &sTile[wave][0][0];
    double* sDi = &sTile[wave][1][0];
    const int kk = t >> 4, cc = t & 15;
    const int nbk = (M2 + 15) >> 4;
    d4 S[4][4], Bt[4];
    double li[4][4];
    // every tile of the upper triangle and the right-hand sides: one round trip
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int a = 0; a <= b; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * a + kk + 4 * r, c = 16 * b + cc;
                const bool in = b < nbk && i < M2 && c < M2;
                const int iu = i <= c ? i : c, cu = i <= c ? c : i;   // (k_feature_e fills rows <= columns only)
                const double ev = Eg[(size_t)(in ? cu : 0) * LDE + (in ? iu : 0)];
                S[a][b][r] = in ? (ev + ((i == c) ? p.sigma2 : 0.0)) : ((i == c) ? 1.0 : 0.0);   // (padding: unit diagonal)
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * a + kk + 4 * r;
            Bt[a][r] = (i < M2 && cc < 4) ? Bg[i * 4 + (cc & 3)] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) li[a][q] = 0.0;
    }
    double dmin = INFINITY;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b < nbk) {
#pragma unroll
            for (int k = 0; k < b; ++k) {
                d4 x = {0, 0, 0, 0};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(li[k][s4], S[k][b][s4], x);
                S[k][b] = x;
#pragma unroll
                for (int a = k + 1; a <= b; ++a) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) S[a][b] = mfma_f64(-S[k][a][s4], S[k][b][s4], S[a][b]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sD[(kk + 4 * r) * 17 + cc] = S[b][b][r];
            wave_sync();
            double v[16], y[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double av = sD[cc * 17 + c];
                v[c] = (c <= cc) ? av : 0.0;
                y[c] = (c == cc) ? 1.0 : 0.0;
            }
            DiagStep<0, false>::run(v, y, 0.0, dmin);
            if (t < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) sDi[c * 17 + t] = y[c];
            }
            wave_sync();
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) li[b][s4] = sDi[cc * 17 + kk + 4 * s4];
            wave_sync();
#pragma unroll
            for (int k = 0; k < b; ++k) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) Bt[b] = mfma_f64(-S[k][b][s4], Bt[k][s4], Bt[b]);
            }
            {
                d4 x = {0, 0, 0, 0};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(li[b][s4], Bt[b][s4], x);
                Bt[b] = x;
            }
        }
    }
    double c0[16], c1[16], c2[16], c3[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = Bt[k][r];
            c0[4 * k + r] = dpp_row_bcast<0>(v);
            c1[4 * k + r] = dpp_row_bcast<1>(v);
            c2[4 * k + r] = dpp_row_bcast<2>(v);
            c3[4 * k + r] = dpp_row_bcast<3>(v);
        }
    }
    auto red = [&](double x) -> double {
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        return x;
    };
    auto dot = [&](const double (&a)[16], const double (&b)[16]) -> double {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i += 2) { s0 += a[i] * b[i]; s1 += a[i + 1] * b[i + 1]; }
        return s0 + s1;
    };
    auto axpy = [&](double (&y)[16], double al, const double (&x)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] -= al * x[i];
    };
    {
        const double n1 = red(dot(c1, c1)), d12 = red(dot(c1, c2)), d13 = red(dot(c1, c3)), d1y = red(dot(c1, c0));
        const double i1 = (n1 > 0.0) ? 1.0 / n1 : 0.0;
        axpy(c2, d12 * i1, c1); axpy(c3, d13 * i1, c1); axpy(c0, d1y * i1, c1);
        const double n2 = red(dot(c2, c2)), d23 = red(dot(c2, c3)), d2y = red(dot(c2, c0));
        const double i2 = (n2 > 0.0) ? 1.0 / n2 : 0.0;
        axpy(c3, d23 * i2, c2); axpy(c0, d2y * i2, c2);
        const double n3 = red(dot(c3, c3)), d3y = red(dot(c3, c0));
        const double i3 = (n3 > 0.0) ? 1.0 / n3 : 0.0;
        axpy(c0, d3y * i3, c3);
    }
    const double gam = red(dot(c0, c0));
    const bool fail = !(dmin > 0.0) || !(gam == gam);
    const bool ok = (!fail) && (gam < p.chi2[M2 - 3]);
    if (t == 0) {
        p.gamma[j] = fail ? NAN : gam;
        p.accept[j] = ok ? 1 : 0;
    }
    if (!ok) {   // a rejected track takes no part in the compression: its three dense rows and its sparse rows become zero
        for (int e = t; e < 3 * p.NAP; e += 64) p.T3[(size_t)3 * j * p.NAP + e] = 0.0;
        for (int e = t; e < 32 * M; e += 64) p.Xobs[(size_t)32 * p.obs_pos[lo + (e >> 5)] + (e & 31)] = 0.0;
    }
}

}  // namespace orcvio_amd

// msckf_kernels.hpp -- hand-written gfx950 (CDNA4) kernels of the MSCKF update path.
//
// Data layout in HBM (all FP64):
//   poses  [N][28]        clone pose records (msckf_math.hpp)
//   P      [n][n]         prior covariance (symmetric)
//   T3     [3F][NAP]      first three rows of Q^T [J | r] of every track; column a <-> state column 15+a (a < NA = n-15),
//                         column NA = residual
//   Xobs   [2 nobs][16]   un-projected rows [H_e 6, td, H_x 6, r, 0, 0], grouped by clone
//   S, Gpart              partial Gram tiles of Xobs (per clone chunk) and of T3 (per row chunk)
//   Ab / A [NAP][NAP]     Gram block [A b; b^T c] of this rank / summed over ranks
//   R_P, R_M [NP][NP]     upper Cholesky factors (X = R^T R) of P and of M = s2 I + L_a^T A L_a
//   U, M, Z               solve intermediates
//
// Kernels (one wavefront = 64 lanes everywhere):
//   k_feature      four wavefronts per track: Jacobians (reference src/orcvio.cpp:1071-1226), Householder QR of H_f
//                  (math_utils.hpp:287-312), E = J P J^T, chi-square gate (:1953-1976) by tile Cholesky, outputs
//   k_gram_pair    FP64-MFMA (v_mfma_f64_16x16x4_f64) Grams of Xobs and T3 -- the compression of :2532-2552 in
//                  Gram form; k_gram (one of the two), k_gram_reduce (sum of the ranks' blocks)
//   k_assemble_A   A = scatter(sum S) - sum Gpart
//   k_potrf_reg    register-resident single-workgroup Cholesky; k_potrf_solve: the same + the triangular solve
//                  in one launch; k_potrf (LDS panels) and k_trsm_rl / k_trsm_lds for large windows and batches
//   k_gemm, k_finish_sqrt   split-K MFMA products of the Kalman solve (:1682-1753)
//   k_obj_*        object blocks (:2154-2193): in object_kernels.hpp (included at the end of this file)
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "msckf_math.hpp"
#include "host/ipc_protocol.hpp"

namespace orcvio_amd {

typedef double d4 __attribute__((ext_vector_type(4)));

// Stores / loads of bytes that another workgroup of the SAME launch consumes (k_potrf_solve, k_front): agent-scope relaxed
// atomics = global_store/load ... sc1 (write-through, L1-bypassing); MI355X_MICROARCH.md "inter-workgroup visibility".
template <bool PUB>
__device__ __forceinline__ void st_pub(double* p, double v) {
    if (PUB) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ double ld_pub(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool PUB>
__device__ __forceinline__ double ld_sel(const double* p) {
    if (PUB) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    // D(16x16) += A(16x4) * B(4x16).  lane l: a = A[l&15][l>>4], b = B[l>>4][l&15],
    // c[r] = C[(l>>4) + 4r][l&15]   (cdna_hip_programming.md section 3, f64 layout)
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ double wave_sum_shfl(double x) {   // (round 1's form: six ds_bpermute pairs; kept for reference)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// broadcast of a double from a lane known at compile time: v_readlane (no LDS traffic)
__device__ __forceinline__ double bcast_lane(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes with DPP moves instead of ds_bpermute shuffles: per step two v_mov_b32_dpp (the halves of the double)
// and one v_add_f64 -- quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror give every 16-lane row its total;
// row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3) carry the totals along, lane 63 ends up with the wave's total, which
// v_readlane broadcasts.  About 20 instructions against ~150 for six __shfl_xor of a double.  (Order of summation differs
// from wave_sum: results agree to rounding, not bit for bit.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double x) {
    x += dpp_move<0xB1, 0xF>(x);    // quad_perm [1,0,3,2]
    x += dpp_move<0x4E, 0xF>(x);    // quad_perm [2,3,0,1]
    x += dpp_move<0x141, 0xF>(x);   // row_half_mirror
    x += dpp_move<0x140, 0xF>(x);   // row_mirror: every lane of a row holds the row's total
    x += dpp_move<0x142, 0xA>(x);   // row_bcast:15 into rows 1 and 3 (the other rows add 0)
    x += dpp_move<0x143, 0xC>(x);   // row_bcast:31 into rows 2 and 3
    return bcast_lane(x, 63);
}
__device__ __forceinline__ double wave_sum(double x) { return wave_sum_dpp(x); }

// 1/sqrt(d) to full double precision: v_rsq_f64 seed + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
}

// ---- FP64 DPP row broadcasts (gfx90a+): lane C of every 16-lane row feeds all lanes of that row.
// One instruction does broadcast + multiply-add; the compiler never emits these.  hipcc does not
// insert VALU->DPP wait states around inline asm, so every consumer block starts with s_nop 1.
template <int C>
__device__ __forceinline__ double dpp_row_bcast(double a) {
    double d;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=&v"(d) : "v"(a), "n"(C));
    return d;
}
template <int C>
__device__ __forceinline__ double dpp_row_bcast_bare(double a) {   // no wait states in front: the caller has placed two instructions behind the write of `a`
    double d;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=&v"(d) : "v"(a), "n"(C));
    return d;
}
template <int C>
__device__ __forceinline__ void dpp_fnmac(double& acc, double a, double b) {   // acc -= a[lane C of the row] * b
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(C));
}
// right-looking Cholesky + inverse of a 16x16 tile: lane (l & 15) holds row r of the tile in v[] and
// column r of the inverse in y[]; the four 16-lane rows of the wavefront work redundantly.
//
// The sweep is a chain of 16 dependent pivots (broadcast, rsqrt, two Newton steps, scale) with 2 (15-J) independent
// rank-1 updates hanging off each.  A wavefront issues in order, so the schedule is written out by hand: pivot J
// first updates column J+1 only, the reciprocal square root of pivot J+1 is started at once, and the remaining
// updates of pivot J are issued between its Newton steps.  Every arithmetic instruction is volatile inline asm to
// pin that order (the compiler would otherwise sink the independent updates behind the dependent chain).
__device__ __forceinline__ double asm_mul(double a, double b) {
    double d;
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double asm_fnma(double a, double b, double c) {   // c - a*b
    double d;
    asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ double asm_rsq(double a) {
    double d;
    asm volatile("v_rsq_f64 %0, %1" : "=v"(d) : "v"(a));
    return d;
}
template <int J, int C>
__device__ __forceinline__ void diag_fill(double (&v)[16], double (&y)[16], double m, double x) {
    if constexpr (C < 16) {
        dpp_fnmac<C>(v[C], m, m);   // A[r][C] -= L[C][J] * L[r][J]
        dpp_fnmac<C>(y[C], m, x);   // Linv[C][r] -= L[C][J] * Linv[J][r]
    }
}
// Pivot bookkeeping of the sweep: a pivot d <= tol is dropped (scale 0: its column of L and its row of the inverse become
// zero -- zero-variance states, rank-deficient Grams).  The sweep itself only keeps the smallest pivot it met (one v_min per
// pivot; the padding rows of a last tile carry 1 on the diagonal, so they are ordinary pivots): "some pivot < -tol" follows from
// it, and the NUMBER of dropped pivots is counted afterwards, off the chain, from the zero diagonal entries of L.  (Per-pivot
// counters were eight more VALU instructions on an issue-bound chain: 4.5 k -> 4.0 k cycles per tile, scripts/micro/diag_bench.)
__device__ __forceinline__ double asm_min(double a, double b) {
    double d;
    asm volatile("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
#ifndef ORCVIO_DIAG_NEWTON
#define ORCVIO_DIAG_NEWTON 1
#endif
template <int J, bool DROP = true>
struct DiagStep {
    // inv: scale 1/sqrt(d_J) of pivot J (0 for a dropped pivot).  a1: entry (J+1, J+1) of the tile as it stands before this step,
    // already broadcast to every lane; entry (J+1, J) is read with a DPP broadcast by the multiply-add itself, so that the next
    // pivot d_{J+1} = a1 - (u inv)^2 and its reciprocal square root hang off `inv` through a chain of dependent operations that is
    // as short as possible (a dependent FP64 operation costs ~25-40 cycles here whatever is issued in between).
    // DROP = false (matrices that are positive definite by construction: M = s2 I + ...): no pivot test, no select on the chain;
    // a non-positive pivot still shows in dmin.
    static __device__ __forceinline__ void step(double (&v)[16], double (&y)[16], double inv, double a1, double c15,
                                                double tol, double& dmin) {
        if constexpr (J + 1 < 16) {
            double w = 0.0;
            asm volatile("" : "+v"(w));
            dpp_fnmac<J + 1>(w, v[J], inv);   // -L[J+1][J] on every lane (only its square is used; v[J] was last written a step ago;
                                              // v_mul_f64 has no DPP form, v_fmac_f64 has)
            const double d = asm_fnma(w, w, a1);
            const double r0 = asm_rsq(d);
            const double m = asm_mul(v[J], inv);
            const double x = asm_mul(y[J], inv);
            v[J] = m;
            y[J] = x;
            const double h = asm_mul(d, 0.5);
            const bool ok = DROP ? d > tol : true;
            dmin = asm_min(dmin, d);
            // (DPP reads need two wait states behind the write of their source: h and the minimum stand between x and the first
            //  update; the update of y[J+2] and t2 between the write of v[J+2] and its broadcast -- no s_nop on this chain)
            dpp_fnmac<J + 1>(v[J + 1], m, m);
            dpp_fnmac<J + 1>(y[J + 1], m, x);
            const double t1 = asm_mul(r0, r0);
            diag_fill<J, J + 2>(v, y, m, x);
            const double t2 = asm_fnma(h, t1, c15);
            double an = 0.0;
            if constexpr (J + 2 < 16) an = dpp_row_bcast_bare<J + 2>(v[J + 2]);
            diag_fill<J, J + 3>(v, y, m, x);
            double r = asm_mul(r0, t2);
            diag_fill<J, J + 4>(v, y, m, x);
            if constexpr (ORCVIO_DIAG_NEWTON >= 2) {
                const double t3 = asm_mul(r, r);
                diag_fill<J, J + 5>(v, y, m, x);
                const double t4 = asm_fnma(h, t3, c15);
                diag_fill<J, J + 6>(v, y, m, x);
                r = asm_mul(r, t4);
            } else {
                diag_fill<J, J + 5>(v, y, m, x);
                diag_fill<J, J + 6>(v, y, m, x);
            }
            diag_fill<J, J + 7>(v, y, m, x);
            diag_fill<J, J + 8>(v, y, m, x);
            diag_fill<J, J + 9>(v, y, m, x);
            diag_fill<J, J + 10>(v, y, m, x);
            diag_fill<J, J + 11>(v, y, m, x);
            diag_fill<J, J + 12>(v, y, m, x);
            diag_fill<J, J + 13>(v, y, m, x);
            diag_fill<J, J + 14>(v, y, m, x);
            diag_fill<J, J + 15>(v, y, m, x);
            DiagStep<J + 1, DROP>::step(v, y, ok ? r : 0.0, an, c15, tol, dmin);
        } else {
            v[J] = asm_mul(v[J], inv);
            y[J] = asm_mul(y[J], inv);
        }
    }
    // dmin: running minimum of the pivots (start it at +infinity); tile rows beyond the matrix must carry 1 on the diagonal
    static __device__ __forceinline__ void run(double (&v)[16], double (&y)[16], double tol, double& dmin) {
        static_assert(J == 0, "the sweep starts at pivot 0");
        double c15 = 1.5;
        asm volatile("" : "+v"(c15));   // keep 1.5 in a register pair (not an inline constant)
        const double d = dpp_row_bcast<0>(v[0]);
        const double a1 = dpp_row_bcast<1>(v[1]);
        const bool ok = DROP ? d > tol : true;
        dmin = asm_min(dmin, d);
        const double r0 = asm_rsq(d);
        const double h = asm_mul(d, 0.5);
        asm volatile("s_nop 1" ::"v"(r0));
        const double r1 = asm_mul(r0, asm_fnma(h, asm_mul(r0, r0), c15));
        const double r2 = asm_mul(r1, asm_fnma(h, asm_mul(r1, r1), c15));
        step(v, y, ok ? r2 : 0.0, a1, c15, tol, dmin);
    }
};
__device__ __forceinline__ void wave_sync() {
    // single-wave workgroups: LDS operations of one wave execute in order; this only
    // stops the compiler from moving LDS accesses across the point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------
// k_feature
// ---------------------------------------------------------------------------------------
struct FeatArgs {
    const double* poses;
    const double* p_w;
    const int* obs_ptr;
    const int* obs_clone;
    const double* obs_z;
    const double* obs_zvel;
    const double* P;
    const int* row_ptr;   // [F+1] row offsets into Hs (rho_j = 2M_j-3, 0 if M_j < 2)
    const int* skip;      // [F] or nullptr: tracks that failed triangulation on the device take no part (k_triangulate)
    const double* chi2;   // [ORCVIO_CHI2_TABLE]
    double* Hs;           // optional: stacked projected blocks (nullptr = not materialised)
    double* T3;           // [3F][NAP]  first three rows of Q^T [J | r] of every track (zero if rejected)
    double* Xobs;         // [2 nobs][16] un-projected rows [H_e(6) td H_x(6) r 0 0], grouped by clone (zero if rejected)
    const int* obs_pos;   // [nobs] position of every observation in the clone-sorted order
    double* gamma;
    int* accept;
    double* Rf;           // optional [F][6]: upper triangle (r00 r01 r02 r11 r12 r22) of the R factor of H_f = Q R (same Q as T3)
    double sigma2;
    int n, leg, N, NA, NAP, Mmax, F;
    int use_larvio, use_left, if_fej, estimate_td;
    int ablate;   // diagnostic only (scripts/gpu_ablate_feature.py): 1 skip E, 2 skip Q^T E Q, 4 skip Cholesky, 8 skip outputs
};

__host__ __device__ inline int feat_lde(int Mmax) { return 2 * Mmax + 1; }
__host__ __device__ inline size_t feat_lds_bytes(int Mmax, int NAP, int N) {
    const int R2 = 2 * Mmax;
    size_t dbl = (size_t)R2 * 7 + (size_t)R2 * 6 + R2 + (size_t)R2 * 4 + 64 * 4 + 16 + 4 + 8 * (size_t)NAP + 272 + 272 +
                 (size_t)R2 * feat_lde(Mmax);
    size_t bytes = dbl * 8 + (size_t)(N + Mmax + 4 + 2 * Mmax) * 4;   // (... | sC2O N | sOC Mmax | sFlag 4 | sDone Mmax | sLim Mmax)
    return (bytes + 15) & ~(size_t)15;
}

// One workgroup of four wavefronts per feature track.
//   B, C  (wave 0)   per-observation Jacobians, Householder QR of H_f (LAPACK dgeqr2 convention)
//   D     (all)      compact-WY coefficients of Q^T [J | r] for the column each thread owns (needed for T3)
//   E     (waves 1-3) E = J P J^T, observations handed out in order through an LDS counter: u_l = J_l P_aa from coalesced P rows
//                    (prefetched one observation ahead), E[:, 2l..2l+1] = J u_l^T with the rows of the block in lanes; a flag
//                    per finished observation
//   G     (wave 0)   the chi-square gate WITHOUT forming the projected block: with Sh = E + s2 I (2M x 2M), Q1 the
//                    first three columns of Q (range of H_f) and N0 the rest,
//                        gamma = r'^T (N0^T Sh N0)^-1 r' = || (I - Pi) L^-1 r ||^2,   Sh = L L^T,
//                    Pi the orthogonal projector onto range(L^-1 Q1)  (generalised least squares identity
//                    N0 (N0^T Sh N0)^-1 N0^T = Sh^-1 - Sh^-1 Q1 (Q1^T Sh^-1 Q1)^-1 Q1^T Sh^-1).  Sh is factored as
//                    16x16 MFMA tiles held in registers (DPP diagonal sweep of the Cholesky kernels, panel and
//                    trailing tiles by MFMA on accumulator-layout operands), the four right-hand sides
//                    [r | Q1] ride along as one more tile column, the projection is modified Gram-Schmidt.  The
//                    factorisation is LEFT-looking, one block column (eight observations) at a time, and starts on a block
//                    column as soon as its observations are flagged: it runs under phase E, one block column behind.
//   I     (all)      outputs: T3 (three dense rows), the un-projected sparse rows Xobs, optionally H'.
// feature_body: track j on the 256 threads `tid` = 0..255 of one four-wavefront team with its own LDS block; the
// team is a whole workgroup (k_feature) or half of one (k_front).  Its three workgroup barriers are unconditional for a
// live track, so two teams of one workgroup stay in step.
template <int NPASS, bool PAD_BARRIERS = false>
__device__ __forceinline__ void feature_body(const FeatArgs& p, const int j, const int tid, double* __restrict__ smem) {
    constexpr int NPD = (NPASS + 2) / 3;   // passes of the 192 threads of waves 1..3 over the NAP columns
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), t = tid & 63;
    const int lo = p.obs_ptr[j];
    const int M = p.obs_ptr[j + 1] - lo;
    if (M < 2 || (p.skip && p.skip[j])) {   // whole workgroup
        if (tid == 0) { p.gamma[j] = NAN; p.accept[j] = 0; }
        for (int e = tid; e < 3 * p.NAP; e += 256) st_pub<PAD_BARRIERS>(&p.T3[(size_t)3 * j * p.NAP + e], 0.0);
        for (int e = tid; e < 32 * M; e += 256) st_pub<PAD_BARRIERS>(&p.Xobs[(size_t)32 * p.obs_pos[lo + (e >> 5)] + (e & 31)], 0.0);
        if (p.Hs) {   // a track dropped by the triangulation still owns rows of the materialised stack: zero them
            const size_t r0 = (size_t)p.row_ptr[j], r1 = (size_t)p.row_ptr[j + 1];
            for (size_t e = r0 * p.NAP + tid; e < r1 * p.NAP; e += 256) p.Hs[e] = 0.0;
        }
        if (PAD_BARRIERS) {   // the other team of the workgroup runs a live track: keep its three barriers company
            __syncthreads(); __syncthreads(); __syncthreads();
        }
        return;
    }
    const int M2 = 2 * M;
    const int R2 = 2 * p.Mmax;
    const int LDE = feat_lde(p.Mmax);
    const int NA = p.NA, NAP = p.NAP, n = p.n;
    const int cb0 = p.leg - 15;   // active index of the first clone column
    const int NAc = cb0 + 6 * p.N;   // end of the clone columns (NA is larger when SLAM feature states follow the clones: their columns are zero here)
    double* sJe = smem;               // [R2][7]  ext(6)+td
    double* sJx = sJe + R2 * 7;       // [R2][6]
    double* sR = sJx + R2 * 6;        // [R2]
    double* sV = sR + R2;             // [R2][4]  Householder vectors
    double* sB = sV + R2 * 4;         // [64][4]  right-hand sides of the gate: r, Q1
    double* sQ = sB + 256;            // [16]     beta(3), g10, g20, g21
    double* sYr = sQ + 16;            // [4]
    double* sUall = sYr + 4;          // [4 waves][NAP][2]
    double* sD = sUall + 8 * NAP;     // [16][17] diagonal tile, row view (17: one row per lane without bank conflicts)
    double* sDi = sD + 272;           // [16][17] its inverse factor
    double* sE = sDi + 272;           // [R2][LDE]
    int* sC2O = (int*)(sE + (size_t)R2 * LDE);   // [N]
    int* sOC = sC2O + p.N;                        // [Mmax]
    int* sFlag = sOC + p.Mmax;                    // [4]  [0] gate verdict, [1] next observation of phase E, [2] QR (phase C) done
    int* sDone = sFlag + 4;                       // [Mmax] observation l of phase E is in sE
    int* sLim = sDone + p.Mmax;                   // [Mmax] columns of P that observation l needs: [0, sLim[l]) (upper triangle of E only)

    for (int i = tid; i < p.N; i += 256) sC2O[i] = -1;
    for (int i = tid; i < p.Mmax; i += 256) sDone[i] = 0;
    if (tid == 0) { sFlag[1] = 0; sFlag[2] = 0; }
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
        {   // Only the upper triangle of E = J P J^T is used (rows <= columns; the gate mirrors inside its diagonal tiles), so the
            // column pair of observation l needs u_l = J_l P only at the columns of the clones of observations <= l (and the
            // extrinsic / td columns in front): a prefix maximum of the clone indices (they ascend in every track the
            // reference produces; the maximum keeps any order correct)
            int mx = (t < M) ? p.obs_clone[lo + t] : -1;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { const int v = __shfl_up(mx, o); if (t >= o) mx = v > mx ? v : mx; }
            if (t < M) sLim[t] = cb0 + 6 * mx + 6;
        }
    }
    __syncthreads();   // the Jacobians are all phase E needs: waves 1..3 start on it while wave 0 does the QR

    if (wave == 0) {
        // ---- C: Householder QR of H_f (2M x 3), LAPACK dgeqr2 convention ------------------
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
        if (p.Rf && t == 0) {   // rows 0 and 1 of the reduced H_f sit in lane 0 (a0, a1): R = Q^T H_f, upper triangle
            double* rf = p.Rf + (size_t)6 * j;
            rf[0] = rdiag[0]; rf[1] = a0[1]; rf[2] = a0[2]; rf[3] = rdiag[1]; rf[4] = a1[2]; rf[5] = rdiag[2];
        }
        wave_sync();
        // right-hand sides of the gate, row t: [r | Q1],  Q = H0 H1 H2,  Q e_q = e_q - V z,
        // z2 = b2 w2, z1 = b1 (w1 - g21 z2), z0 = b0 (w0 - g10 z1 - g20 z2),  w = V^T e_q = row q of V
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
            for (int q = 0; q < 4; ++q) sB[t * 4 + q] = b4[q];
        }
        wave_sync();
        if (t == 0) __hip_atomic_store(&sFlag[2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // sV, sQ, sB are final (phase D reads them)
    }

    // ---- E: E = J P_aa J^T (waves 1..3; wave 0 factors the block columns of E + s2 I as they complete, phase G).
    //         Observations are handed out IN ORDER through an LDS counter; every wave keeps one observation in hand and
    //         the P rows of the next one in flight, and flags each finished observation -----------------
    if (wave > 0) {
        double* sU = sUall + (size_t)wave * 2 * NAP;
        // row-lane data: lane t <-> row t of the 2M-row block
        const bool rowlane = t < M2;
        double je[7], jx[6];
        int ja0 = 0;
#pragma unroll
        for (int e = 0; e < 7; ++e) je[e] = rowlane ? sJe[t * 7 + e] : 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) jx[c] = rowlane ? sJx[t * 6 + c] : 0.0;
        if (rowlane) ja0 = cb0 + 6 * sOC[t >> 1];
        double pe[NPASS][7];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int a = t + 64 * ps;
#pragma unroll
            for (int e = 0; e < 7; ++e) pe[ps][e] = p.P[(size_t)(15 + e) * n + 15 + (a < NA ? a : NA - 1)];
        }
        // With estimate_extrin = estimate_td = 0 (every shipped config) the extrinsic / td rows and columns of P are
        // exactly zero (src/orcvio.cpp:213-221): those 7 of the 13 terms then vanish identically and are skipped.
        bool pe_nz = false;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
            for (int e = 0; e < 7; ++e) pe_nz |= (pe[ps][e] != 0.0);
        const bool ext_live = __any(pe_nz);   // wave-uniform (and the same on every wave: same loads)
        const int lend = (p.ablate & 1) ? 0 : M;
        auto grab = [&]() -> int {
            int v = 0;
            if (t == 0) v = atomicAdd(&sFlag[1], 1);
            return __builtin_amdgcn_readfirstlane(v);
        };
        double pcur[NPASS][6], pnxt[NPASS][6];
        int l = grab();
        int lim = 0, limn = 0;   // (wave-uniform) columns of P this / the next observation needs
        if (l < lend) {
            lim = __builtin_amdgcn_readfirstlane(sLim[l]);
            lim = lim < NA ? lim : NA;
            const double* Prow = p.P + (size_t)(p.leg + 6 * sOC[l]) * n + 15;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                if (64 * ps < lim) {
                    const int a = t + 64 * ps;
#pragma unroll
                    for (int c = 0; c < 6; ++c) pcur[ps][c] = Prow[(size_t)c * n + (a < NA ? a : NA - 1)];
                }
            }
        }
        while (l < lend) {
            const int lnext = grab();
            if (lnext < lend) {
                limn = __builtin_amdgcn_readfirstlane(sLim[lnext]);
                limn = limn < NA ? limn : NA;
                const double* Prow = p.P + (size_t)(p.leg + 6 * sOC[lnext]) * n + 15;
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    if (64 * ps < limn) {
                        const int a = t + 64 * ps;
#pragma unroll
                        for (int c = 0; c < 6; ++c) pnxt[ps][c] = Prow[(size_t)c * n + (a < NA ? a : NA - 1)];
                    }
                }
            }
            double jl0e[7], jl1e[7], jl0x[6], jl1x[6];
#pragma unroll
            for (int e = 0; e < 7; ++e) { jl0e[e] = sJe[(2 * l) * 7 + e]; jl1e[e] = sJe[(2 * l + 1) * 7 + e]; }
#pragma unroll
            for (int c = 0; c < 6; ++c) { jl0x[c] = sJx[(2 * l) * 6 + c]; jl1x[c] = sJx[(2 * l + 1) * 6 + c]; }
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int a = t + 64 * ps;
                if (a < lim) {
                    double u0 = 0.0, u1 = 0.0;
                    if (ext_live) {
#pragma unroll
                        for (int e = 0; e < 7; ++e) { u0 += jl0e[e] * pe[ps][e]; u1 += jl1e[e] * pe[ps][e]; }
                    }
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        u0 += jl0x[c] * pcur[ps][c];
                        u1 += jl1x[c] * pcur[ps][c];
                    }
                    sU[2 * a] = u0;
                    sU[2 * a + 1] = u1;
                }
            }
            wave_sync();
            if (rowlane && t <= 2 * l + 1) {   // (rows <= columns 2l, 2l+1: the upper triangle)
                double e0 = 0.0, e1 = 0.0;
                if (ext_live) {   // P symmetric: zero rows <=> zero columns, so u at the ext columns is zero too
#pragma unroll
                    for (int e = 0; e < 7; ++e) { e0 += je[e] * sU[2 * e]; e1 += je[e] * sU[2 * e + 1]; }
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) { e0 += jx[c] * sU[2 * (ja0 + c)]; e1 += jx[c] * sU[2 * (ja0 + c) + 1]; }
                sE[t * LDE + 2 * l] = e0;
                sE[t * LDE + 2 * l + 1] = e1;
            }
            wave_sync();
            if (t == 0) __hip_atomic_store(&sDone[l], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // (LDS writes of one wave are in order)
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
                for (int c = 0; c < 6; ++c) pcur[ps][c] = pnxt[ps][c];
            l = lnext;
            lim = limn;
        }
        if (lend == 0 && wave == 1 && t < M) __hip_atomic_store(&sDone[t], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // (diagnostic: phase E ablated)
        // phase D needs the reflectors of phase C (wave 0 finished them long ago: E is several times longer)
#pragma unroll 1
        for (int it = 0; it < (1 << 22); ++it) {   // (bounded)
            if (__hip_atomic_load(&sFlag[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }

    // ---- D (waves 1..3, while wave 0 runs the gate): compact-WY coefficients y_q[a] of Q^T [J | r] for the
    //      columns a = (tid - 64) + 192 ps this thread owns; only the outputs need them
    double yq[NPD][3];
    if (wave > 0) {
        const int td = tid - 64;
        const double be0 = sQ[0], be1 = sQ[1], be2 = sQ[2], g10 = sQ[3], g20 = sQ[4], g21 = sQ[5];
#pragma unroll
        for (int ps = 0; ps < NPD; ++ps) {
            const int a = td + 192 * ps;
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

    // ---- G: the gate (wave 0) ---------------------------------------------------------------
    if (wave == 0) {
        const int kk = t >> 4, cc = t & 15;
        const int nbk = (M2 + 15) >> 4;
        d4 S[4][4], Bt[4];
        double li[4][4];   // inv(L11) of every block step, MFMA A operand
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * a + kk + 4 * r;
                Bt[a][r] = (i < M2 && cc < 4) ? sB[i * 4 + (cc & 3)] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) li[a][q] = 0.0;
        }
        double dmin = INFINITY;   // smallest pivot of the factorisation
        int nneg = 0;             // (a wait that gave up)
        if (!(p.ablate & 4)) {
            // Left-looking by block column b of the upper factor (Sh = R^T R, tiles S[a][b], a <= b): the column is read when
            // its eight observations are flagged, brought up to date against the finished block rows k < b, its diagonal tile
            // factored + inverted in one DPP sweep; the right-hand sides follow one block row behind.
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b < nbk) {
                    {   // observations 8b .. 8b+7 (those that exist) are in sE
                        const int lw = 8 * b + (t & 7);
                        bool done = lw >= M;
#pragma unroll 1
                        for (int it = 0; it < (1 << 22); ++it) {   // (bounded: never more than phase E takes -- a few microseconds)
                            if (!done) done = __hip_atomic_load(&sDone[lw], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
                            if (__all(done)) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (!__all(done)) ++nneg;   // (the track is rejected rather than gated on an incomplete E)
                    }
#pragma unroll
                    for (int a = 0; a <= b; ++a) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * a + kk + 4 * r, c = 16 * b + cc;
                            const bool in = i < M2 && c < M2;
                            const int iu = i <= c ? i : c, cu = i <= c ? c : i;   // (phase E fills rows <= columns only)
                            const double ev = sE[(in ? iu : 0) * LDE + (in ? cu : 0)];
                            S[a][b][r] = in ? (ev + ((i == c) ? p.sigma2 : 0.0)) : ((i == c) ? 1.0 : 0.0);   // (padding: unit diagonal)
                        }
                    }
#pragma unroll
                    for (int k = 0; k < b; ++k) {
                        d4 x = {0, 0, 0, 0};   // row block k of the factor, block column b: inv(L11_k) * (tile, up to date)
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(li[k][s4], S[k][b][s4], x);
                        S[k][b] = x;
#pragma unroll
                        for (int a = k + 1; a <= b; ++a) {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) S[a][b] = mfma_f64(-S[k][a][s4], S[k][b][s4], S[a][b]);
                        }
                    }
                    // diagonal tile: accumulator layout -> rows in lanes, factor + invert in one DPP sweep
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
                    DiagStep<0, false>::run(v, y, 0.0, dmin);   // (Sh = E + s2 I is positive definite unless the prior is not PSD: caught by dmin)
                    if (t < 16) {
#pragma unroll
                        for (int c = 0; c < 16; ++c) sDi[c * 17 + t] = y[c];   // Linv[c][t]
                    }
                    wave_sync();
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) li[b][s4] = sDi[cc * 17 + kk + 4 * s4];
                    wave_sync();
                    // right-hand sides, block row b (off the critical path of the next block column)
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
        }
        // y = L^-1 r (column 0), Y = L^-1 Q1 (columns 1..3): every lane gathers its 16 rows of all four columns, then
        // modified Gram-Schmidt (redundantly on every lane; sums over the four 16-lane rows by two butterflies)
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
        const bool fail = nneg > 0 || !(dmin > 0.0) || !(gam == gam);   // a non-positive (or NaN) pivot: Sh is not positive definite
        const int dof = M2 - 3;
        const bool ok = (!fail) && (gam < p.chi2[dof]);
        if (t == 0) {
            p.gamma[j] = fail ? NAN : gam;
            p.accept[j] = ok ? 1 : 0;
            sFlag[0] = ok ? 1 : 0;
        }
    }
    // ---- I: outputs -------------------------------------------------------------------------
    // The compression needs only  H'^T H' = X^T X - T3^T T3  (X = [J | r] un-projected, T3 = rows 0..2 of
    // Q^T X: Q is orthogonal), so the 2M-3 dense projected rows are NOT needed downstream: the track
    // hands over its 2M sparse rows (14 non-zeros each) and the three dense rows T3.  The projected block
    // itself is materialised only on request (p.Hs != nullptr; tests and callers that want H').
    // Waves 1..3 write T3 and the sparse rows SPECULATIVELY, as for an accepted track, while wave 0 is still in the tail of the
    // gate; a rejected track overwrites them with zeros behind the barrier (same threads, same addresses: ordered).
    const bool outputs = !(p.ablate & 8);
    auto write_t3 = [&](bool live) {
#pragma unroll
        for (int ps = 0; ps < NPD; ++ps) {
            const int a = (tid - 64) + 192 * ps;
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
                    if (live && a <= NA) {
                        double jv = 0.0;
                        if (a < 7) jv = sJe[i * 7 + a];
                        else if (a == NA) jv = sR[i];
                        else if (kobs == (i >> 1)) jv = sJx[i * 6 + cc];
                        val = jv - (sV[i * 4 + 0] * yq[ps][0] + sV[i * 4 + 1] * yq[ps][1] + sV[i * 4 + 2] * yq[ps][2]);
                    }
                    st_pub<PAD_BARRIERS>(&p.T3[((size_t)3 * j + i) * NAP + a], val);   // (k_front: consumed inside the launch)
                }
            }
        }
    };
    // un-projected rows of this track: [H_e(6) td | H_x(6) | r | 0 0], 16 doubles per row, stored at the
    // observation's position in the clone-sorted order (so that k_gram reads every clone contiguously)
    auto write_xobs = [&](bool live) {
        for (int e = tid - 64; e < 16 * M2; e += 192) {
            const int row = e >> 4, c = e & 15;
            double v = 0.0;
            if (live && c < 14) v = (c < 7) ? sJe[row * 7 + c] : ((c < 13) ? sJx[row * 6 + (c - 7)] : sR[row]);
            st_pub<PAD_BARRIERS>(&p.Xobs[(size_t)32 * p.obs_pos[lo + (row >> 1)] + 16 * (row & 1) + c], v);
        }
    };
    if (outputs && wave > 0) { write_t3(true); write_xobs(true); }
    __syncthreads();
    const bool ok = sFlag[0] != 0;
    if (!outputs) return;
    if (!ok && wave > 0) { write_t3(false); write_xobs(false); }
    if (p.Hs && wave > 0) {
#pragma unroll
        for (int ps = 0; ps < NPD; ++ps) {
            const int a = (tid - 64) + 192 * ps;
            if (a < NAP) {
                int kobs = -1, cc = 0;
                if (a >= cb0 && a < NAc) {
                    const int cl = (a - cb0) / 6;
                    cc = (a - cb0) - 6 * cl;
                    kobs = sC2O[cl];
                }
                const size_t row0 = (size_t)p.row_ptr[j];
                for (int i = 3; i < M2; ++i) {
                    double val = 0.0;
                    if (ok && a <= NA) {
                        double jv = 0.0;
                        if (a < 7) jv = sJe[i * 7 + a];
                        else if (a == NA) jv = sR[i];
                        else if (kobs == (i >> 1)) jv = sJx[i * 6 + cc];
                        val = jv - (sV[i * 4 + 0] * yq[ps][0] + sV[i * 4 + 1] * yq[ps][1] + sV[i * 4 + 2] * yq[ps][2]);
                    }
                    p.Hs[(row0 + i - 3) * NAP + a] = val;
                }
            }
        }
    }
}

template <int NPASS>   // (windows beyond 33 clones, NPASS >= 5: one workgroup per CU -- 512 registers per lane instead of 256: no scratch; their E tile fills the LDS anyway)
__global__ __launch_bounds__(256, NPASS >= 5 ? 1 : 2) void k_feature(FeatArgs p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    feature_body<NPASS>(p, blockIdx.x, threadIdx.x, smem);
}

// ---------------------------------------------------------------------------------------
// Sparse part of the compression.  Every row of X = [J | r] touches exactly one clone, so X^T X is
// "arrow + block diagonal": (ext|r) x (ext|r), (ext|r) x clone_i, clone_i x clone_i.  The rows are stored
// grouped by clone, 16 doubles each, so S_i = X_i^T X_i is ONE 16x16 MFMA tile per chunk of rows: k_gram
// with NAP = 16 and ragged chunks (chunk_ptr) computes it.
//
// k_assemble_A:  A (NAP x NAP, full symmetric) = scatter(sum of S chunks) - sum_c Gpart[c], with Gpart the
// partial Grams of T3 (lower tiles valid).  Column map of a sparse-row entry e: 0..6 -> a = e (ext, td),
// 7..12 -> clone block, 13 -> a = NA (right-hand side).
// ---------------------------------------------------------------------------------------
template <bool PUB = false>
__device__ __forceinline__ void assemble_entry(const int idx, const double* __restrict__ Sp, int N, int cb0, int NA, int NAP,
                                               const double* __restrict__ parts, int nparts, size_t part_stride,
                                               double* __restrict__ dst, int dbg, const double* __restrict__ plus = nullptr) {
    // plus: a Gram that is ADDED (lower tiles valid): the EKF-SLAM rows of the hybrid filter (ekf_rows.hpp).  The clone
    // columns end at cb0 + 6N; with SLAM feature states behind them NA is larger and those columns get no S term.
    // Sp: one 16x16 tile per clone (k_gram_pair).  An entry of a clone block or of the arrow reads one tile; the
    // shared (ext|r) x (ext|r) entries sum all N -- sixteen loads in flight (clamped index + select: no serial tail),
    // fixed summation order.
    const int i = idx / NAP, j = idx - i * NAP;
    int ei = -1, ci = -1, ej = -1, cj = -1;
    const int NAc = cb0 + 6 * N;   // end of the clone columns
    if (i < 7) ei = i; else if (i == NA) ei = 13; else if (i >= cb0 && i < NAc) { ci = (i - cb0) / 6; ei = 7 + (i - cb0) - 6 * ci; }
    if (j < 7) ej = j; else if (j == NA) ej = 13; else if (j >= cb0 && j < NAc) { cj = (j - cb0) / 6; ej = 7 + (j - cb0) - 6 * cj; }
    const int src = ((i >> 4) >= (j >> 4)) ? idx : j * NAP + i;
    // (the partial Grams first: their loads are unconditional and stay in flight under the S branch)
    double gv[4];
    const int np0 = (dbg & 2) ? 0 : nparts;
#pragma unroll
    for (int u = 0; u < 4; ++u) gv[u] = ld_sel<PUB>(parts + (size_t)(u < nparts ? u : nparts - 1) * part_stride + src);
    double s = 0.0;
    if (ei >= 0 && ej >= 0 && N > 0 && !(dbg & 1)) {
        // the 16x16 tiles hold both triangles: read [max][min]
        const int e = (ei >= ej) ? ei * 16 + ej : ej * 16 + ei;
        int c0 = 0, c1 = 0;
        if (ci < 0 && cj < 0) { c0 = 0; c1 = N; }
        else if (ci >= 0 && cj >= 0) { if (ci == cj) { c0 = ci; c1 = ci + 1; } }
        else { c0 = ci >= 0 ? ci : cj; c1 = c0 + 1; }
        if (c1 - c0 == 1) {
            s = ld_sel<PUB>(Sp + (size_t)c0 * 256 + e);
        } else if (c1 > c0) {
            double sa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int c = c0; c < c1; c += 16) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int cu = c + u;
                    const double v = ld_sel<PUB>(Sp + (size_t)(cu < c1 ? cu : c1 - 1) * 256 + e);
                    sa[u & 7] += cu < c1 ? v : 0.0;
                }
            }
            s = ((sa[0] + sa[1]) + (sa[2] + sa[3])) + ((sa[4] + sa[5]) + (sa[6] + sa[7]));
        }
    }
    double g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) g[u] = u < np0 ? gv[u] : 0.0;
    for (int c = 4; c < np0; c += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cu = c + u;
            const double v = ld_sel<PUB>(parts + (size_t)(cu < nparts ? cu : nparts - 1) * part_stride + src);
            g[u] += cu < nparts ? v : 0.0;
        }
    }
    dst[idx] = s - ((g[0] + g[1]) + (g[2] + g[3])) + (plus ? plus[src] : 0.0);
}
__global__ __launch_bounds__(256) void k_assemble_A(const double* __restrict__ Sp, int N,
                                                    int cb0, int NA, int NAP, const double* __restrict__ parts, int nparts,
                                                    size_t part_stride, double* __restrict__ dst, int dbg = 0,
                                                    const double* __restrict__ plus = nullptr) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= NAP * NAP) return;
    assemble_entry(idx, Sp, N, cb0, NA, NAP, parts, nparts, part_stride, dst, dbg, plus);
}

// ---------------------------------------------------------------------------------------
// k_gram: partial Gram G_c = X_c^T X_c over a chunk of rows, lower 16x16 tiles only.
// One wavefront per (chunk, tile).  grid.x = ceil(ntiles/4), grid.y = chunks, block 256.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_from_linear(int tl, int& bi, int& bj) {
    // lower-triangular tile index tl -> (bi >= bj)
    int b = (int)((sqrt(8.0 * tl + 1.0) - 1.0) * 0.5);
    while ((b + 1) * (b + 2) / 2 <= tl) ++b;
    while (b * (b + 1) / 2 > tl) --b;
    bi = b;
    bj = tl - b * (b + 1) / 2;
}

// lower tile `tl` of the Gram of the rows [r0, r1) of X, by ONE wavefront (lane l), into `out` (NAP x NAP)
__device__ __forceinline__ void gram_tile(const double* __restrict__ X, int NAP, int r0, int r1, double* __restrict__ out, const int tl, const int l) {
    int bi, bj;
    tile_from_linear(tl, bi, bj);
    const int kk = l >> 4, cc = l & 15;
    const double* pa = X + 16 * bi + cc;
    const double* pb = X + 16 * bj + cc;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    constexpr int GB = 16;   // k-steps (of 4 rows) whose operands are in flight together
    for (int k = r0; k < r1; k += 4 * GB) {
        double a[GB], b[GB];
#pragma unroll
        for (int q = 0; q < GB; ++q) {   // clamped address + select (no branch, no vmcnt(0) per load)
            const int rq = k + 4 * q + kk;
            const int rc = rq < r1 ? rq : r1 - 1;
            const double av = pa[(size_t)rc * NAP], bv = pb[(size_t)rc * NAP];
            a[q] = rq < r1 ? av : 0.0;
            b[q] = rq < r1 ? bv : 0.0;
        }
#pragma unroll
        for (int q = 0; q < GB; q += 2) {
            acc0 = mfma_f64(a[q], b[q], acc0);
            acc1 = mfma_f64(a[q + 1], b[q + 1], acc1);
        }
    }
    if (r1 <= r0) { acc0 = d4{0, 0, 0, 0}; acc1 = d4{0, 0, 0, 0}; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 16 * bi + kk + 4 * r, jj = 16 * bj + cc;
        out[(size_t)i * NAP + jj] = acc0[r] + acc1[r];
    }
}
__device__ __forceinline__ void gram_body(const double* __restrict__ X, int m, int NAP, int rows_per_chunk,
                                          double* __restrict__ Gpart, const int* __restrict__ chunk_ptr, int bx, int chunk) {
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nb = NAP >> 4;
    const int ntiles = nb * (nb + 1) / 2;
    const int tl = bx * 4 + wave;
    if (tl >= ntiles) return;
    int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (chunk_ptr) { r0 = chunk_ptr[chunk]; r1 = chunk_ptr[chunk + 1]; }   // ragged chunks (one per object block)
    if (r1 > m) r1 = m;
    gram_tile(X, NAP, r0, r1, Gpart + (size_t)chunk * NAP * NAP, tl, l);
}
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ X, int m, int NAP, int rows_per_chunk,
                                              double* __restrict__ Gpart, const int* __restrict__ chunk_ptr = nullptr) {
    gram_body(X, m, NAP, rows_per_chunk, Gpart, chunk_ptr, blockIdx.x, blockIdx.y);
}
// Both Grams of the compression in one launch, 1024-thread workgroups: the sixteen wavefronts of a workgroup split
// the rows of ONE output tile between them (one batch of loads in flight per wavefront: a single memory round
// trip), the sixteen partial tiles are summed through LDS in a fixed order.
//   blockIdx.y <  chunks : dense rows T3 (width NAP), lower tile blockIdx.x, row chunk blockIdx.y  -> Gpart[chunk]
//   blockIdx.y == chunks : sparse rows Xobs (width 16, grouped by clone), clone blockIdx.x          -> S[clone]
template <int NWAVES, int GB, bool PUB = false, bool MIRROR = false>
__device__ __forceinline__ void gramw_body(double* __restrict__ sT /* [NWAVES][256] LDS */, const double* __restrict__ X, int ld,
                                           int r0, int r1, int bi, int bj, double* __restrict__ out, int ldo) {
    // GB: k-steps (of 4 rows) whose operands are in flight together
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const int kk = l >> 4, cc = l & 15;
    int rw = (r1 - r0 + NWAVES - 1) / NWAVES;
    rw = (rw + 3) & ~3;   // rows per wavefront, whole k-steps
    const int k0 = r0 + wave * rw;
    const int k1 = (k0 + rw < r1) ? k0 + rw : r1;
    const double* pa = X + 16 * bi + cc;
    const double* pb = X + 16 * bj + cc;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int k = k0; k < k1; k += 4 * GB) {
        double a[GB], b[GB];
#pragma unroll
        for (int q = 0; q < GB; ++q) {   // clamped address + select (no branch, no vmcnt(0) per load)
            const int rq = k + 4 * q + kk;
            const int rc = rq < k1 ? rq : k1 - 1;
            const double av = ld_sel<PUB>(pa + (size_t)rc * ld), bv = ld_sel<PUB>(pb + (size_t)rc * ld);
            a[q] = rq < k1 ? av : 0.0;
            b[q] = rq < k1 ? bv : 0.0;
        }
#pragma unroll
        for (int q = 0; q < GB; q += 2) {
            acc0 = mfma_f64(a[q], b[q], acc0);
            acc1 = mfma_f64(a[q + 1], b[q + 1], acc1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sT[wave * 256 + r * 64 + l] = acc0[r] + acc1[r];
    __syncthreads();
    if (tid < 256) {
        double t[NWAVES];
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) t[w] = sT[w * 256 + tid];
#pragma unroll
        for (int st = 1; st < NWAVES; st *= 2)   // fixed pairwise order
#pragma unroll
            for (int w = 0; w + st < NWAVES; w += 2 * st) t[w] += t[w + st];
        const int r = tid >> 6, lk = (tid & 63) >> 4, lc = tid & 15;   // accumulator element (row lk + 4 r, column lc)
        st_pub<PUB>(out + (size_t)(16 * bi + lk + 4 * r) * ldo + 16 * bj + lc, t[0]);
        if (MIRROR && bi != bj) st_pub<PUB>(out + (size_t)(16 * bj + lc) * ldo + 16 * bi + lk + 4 * r, t[0]);   // the other triangle
    }
}
__device__ __forceinline__ void gram16_body(const double* __restrict__ X, int ld, int r0, int r1, int bi, int bj,
                                            double* __restrict__ out, int ldo) {
    __shared__ __attribute__((aligned(16))) double sT[16 * 256];
    gramw_body<16, 16>(sT, X, ld, r0, r1, bi, bj, out, ldo);
}
__global__ __launch_bounds__(1024) void k_gram_pair(const double* __restrict__ T3, int m3, int NAP, int rows_per_chunk, int chunks,
                                                    double* __restrict__ Gpart, const double* __restrict__ Xobs,
                                                    double* __restrict__ S, const int* __restrict__ clone_rows, int N) {
    const int nb = NAP >> 4;
    if ((int)blockIdx.y < chunks) {
        if ((int)blockIdx.x >= nb * (nb + 1) / 2) return;
        int bi, bj;
        tile_from_linear(blockIdx.x, bi, bj);
        const int r0 = blockIdx.y * rows_per_chunk;
        const int r1 = (r0 + rows_per_chunk < m3) ? r0 + rows_per_chunk : m3;
        gram16_body(T3, NAP, r0, r1 > r0 ? r1 : r0, bi, bj, Gpart + (size_t)blockIdx.y * NAP * NAP, NAP);
    } else {
        if ((int)blockIdx.x >= N) return;
        gram16_body(Xobs, 16, clone_rows[blockIdx.x], clone_rows[blockIdx.x + 1], 0, 0, S + (size_t)blockIdx.x * 256, 16);
    }
}

// Sums `nparts` blocks into dst (full symmetric result); used for the chunk partials of one
// rank (lower tiles valid) and for the all-gathered blocks of all ranks.
// shard_meta (sharded updates): ORCVIO_SHARD_META status words stand behind every rank's block, the first at shard_meta; block
// 0 leaves [first failing rank + 1 (0: none), its status, total dof, total accepted rows, ranks present] in shard_info[0..4] -- the same
// numbers on every rank, from the same gathered bytes (zeros when there are no status words).
__global__ __launch_bounds__(256) void k_gram_reduce(const double* __restrict__ parts, int nparts, size_t part_stride,
                                                     int NAP, double* __restrict__ dst, const double* __restrict__ shard_meta = nullptr,
                                                     int* __restrict__ shard_info = nullptr) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx == 0 && shard_info) {
        int bad = 0, st = 0, seen = 0;
        long dof = 0, rows = 0;
        if (shard_meta)
            for (int r = 0; r < nparts; ++r) {
                const double* m = shard_meta + (size_t)r * part_stride;
                if (bad == 0 && m[0] != 0.0) { bad = r + 1; st = (int)m[0]; }
                dof += (long)m[1]; rows += (long)m[2];
                seen += m[5] == (double)(r + 1) ? 1 : 0;   // (ORCVIO_SHARD_META_RANK: the slot carries its sender's number)
            }
        shard_info[0] = bad; shard_info[1] = st; shard_info[2] = (int)dof; shard_info[3] = (int)rows; shard_info[4] = seen;
    }
    if (idx >= NAP * NAP) return;
    const int i = idx / NAP, jj = idx - i * NAP;
    // parts hold lower tiles only; mirror them so that dst is the full symmetric block
    const int src = ((i >> 4) >= (jj >> 4)) ? idx : jj * NAP + i;
    dst[idx] = rank_ordered_sum(parts, nparts, part_stride, (size_t)src);   // (host/ipc_protocol.hpp: fixed order, the same bits on every rank)
}

// ---------------------------------------------------------------------------------------
// k_potrf: single-workgroup (1024 threads) blocked right-looking Cholesky, in place, lower.
//   A (n x n, ld) symmetric positive SEMI-definite: a pivot <= tol zeroes its column
//   (rank-deficient directions of the Gram block: gauge freedom, unobserved columns).
//   Also writes the inverses of the 16x16 diagonal blocks (Dinv[nblk][16][16], generalised
//   inverse on zero pivots) for k_trsm, zeroes the strict upper triangle, and counts zero
//   pivots in *nzero.
// ---------------------------------------------------------------------------------------
#define POTRF_MAXN 416
__global__ __launch_bounds__(1024) void k_potrf(double* __restrict__ A, int n, int ld, double tol_rel,
                                                double* __restrict__ Dinv, int* __restrict__ nzero, double shift_rel = 0.0) {
    __shared__ double sD[16][17];
    __shared__ double sDi[16][17];
    __shared__ double sdinv[16];
    __shared__ double sP[POTRF_MAXN][17];   // panel rows below the diagonal block
    __shared__ double sred[16];
    __shared__ int szero;
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const int nblk = (n + 15) >> 4;
    // tolerance from the largest diagonal entry
    double mx = 0.0;
    for (int i = tid; i < n; i += 1024) mx = fmax(mx, A[(size_t)i * ld + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if (l == 0) sred[wave] = mx;
    if (tid == 0) szero = 0;
    __syncthreads();
    mx = 0.0;
    for (int w = 0; w < 16; ++w) mx = fmax(mx, sred[w]);
    const double tol = tol_rel * mx;
    const double shift = shift_rel * mx;   // optional Tikhonov shift of the diagonal (factor of A + shift I)

    for (int kb = 0; kb < nblk; ++kb) {
        const int k0 = kb * 16;
        const int kw = (n - k0) < 16 ? (n - k0) : 16;   // width of this block column
        // (1) diagonal block -> LDS, unblocked factorisation by wave 0
        if (tid < 256) {
            const int r = tid >> 4, c = tid & 15;
            double v = 0.0;
            if (r < kw && c < kw && c <= r) v = A[(size_t)(k0 + r) * ld + k0 + c] + ((r == c) ? shift : 0.0);
            sD[r][c] = v;
        }
        __syncthreads();
        if (wave == 0) {
            const int r = l & 15, cg = l >> 4;
            for (int jc = 0; jc < 16; ++jc) {
                const double d = sD[jc][jc];
                double inv = 0.0, sq = 0.0;
                if (jc < kw && d > tol) { sq = sqrt(d); inv = 1.0 / sq; }
                __builtin_amdgcn_wave_barrier();
                if (l == 0) {
                    sdinv[jc] = inv;
                    sD[jc][jc] = sq;
                    if (jc < kw && !(d > tol)) szero += 1;
                }
                if (cg == 0 && r > jc) sD[r][jc] = sD[r][jc] * inv;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // trailing update inside the block: columns c = jc+1+cg, +4, ...
                for (int c = jc + 1 + cg; c <= r; c += 4) sD[r][c] -= sD[r][jc] * sD[c][jc];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            // generalised inverse of the diagonal block: Di = D^-1 (lower), column per lane
            if (l < 16) {
                const int c = l;
                double x[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < i) s -= sD[i][k] * x[k];
                    x[i] = (i >= c) ? s * sdinv[i] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) sDi[i][c] = x[i];
            }
        }
        __syncthreads();
        // write back the factored diagonal block (+ zero strict upper) and its inverse
        if (tid < 256) {
            const int r = tid >> 4, c = tid & 15;
            if (r < kw && c < kw) A[(size_t)(k0 + r) * ld + k0 + c] = (c <= r) ? sD[r][c] : 0.0;
            Dinv[(size_t)kb * 256 + r * 16 + c] = sDi[r][c];
        }
        // (2) panel: rows below, x <- x * D^-T by forward substitution (thread per row)
        const int prow0 = k0 + 16;
        const int np = n - prow0;   // rows below
        for (int pr = tid; pr < np; pr += 1024) {
            double x[16];
            const double* src = A + (size_t)(prow0 + pr) * ld + k0;
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = src[c];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double s = x[c];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < c) s -= x[k] * sD[c][k];
                x[c] = s * sdinv[c];
            }
            double* dst = A + (size_t)(prow0 + pr) * ld + k0;
#pragma unroll
            for (int c = 0; c < 16; ++c) { dst[c] = x[c]; sP[pr][c] = x[c]; }
        }
        // zero the strict-upper part right of the diagonal block (rows of this block)
        for (int idx = tid; idx < 16 * np; idx += 1024) {
            const int r = idx / np, c = idx - r * np;
            if (r < kw) A[(size_t)(k0 + r) * ld + prow0 + c] = 0.0;
        }
        __syncthreads();
        // (3) trailing update A22 -= L21 L21^T (lower tiles) with FP64 MFMA, K = 16
        const int nbt = (np + 15) >> 4;
        const int ntile = nbt * (nbt + 1) / 2;
        for (int tl = wave; tl < ntile; tl += 16) {
            int bi, bj;
            tile_from_linear(tl, bi, bj);
            const int kk = l >> 4, cc = l & 15;
            d4 acc = {0, 0, 0, 0};
            const int ra = 16 * bi + cc, rb = 16 * bj + cc;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = (ra < np) ? sP[ra][kk + 4 * s] : 0.0;
                const double b = (rb < np) ? sP[rb][kk + 4 * s] : 0.0;
                acc = mfma_f64(a, b, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * bi + kk + 4 * r, jj = 16 * bj + cc;
                if (i < np && jj < np && jj <= i) A[(size_t)(prow0 + i) * ld + prow0 + jj] -= acc[r];
            }
        }
        __syncthreads();
    }
    if (tid == 0) *nzero = szero;
}

// ---------------------------------------------------------------------------------------
// generic small FP64-MFMA product: one wavefront per 16x16 output tile.
//   C[i][j] = sum_k A(i,k) * B(k,j),  A(i,k) = A[i*sAi + k*sAk], B(k,j) = B[k*sBk + j*sBj]
// ---------------------------------------------------------------------------------------
template <int TP_BATCH = 12, bool PUB = false>   // PUB: operands another workgroup of this launch stored (st_pub): L1-bypassing loads
__device__ __forceinline__ d4 tile_product(const double* __restrict__ A, long sAi, long sAk, const double* __restrict__ B,
                                           long sBk, long sBj, int M, int N, int K, int i0, int j0, int l) {
    // Operands of TP_BATCH k-steps are loaded before the first MFMA of the batch: the kernels built on
    // this are latency-bound (one tile per wavefront), so loads must be in flight together.
    // Out-of-range rows / columns are loaded from a clamped (valid) address and zeroed by a select (a
    // predicated load is a branch and serialises the loads); full batches advance plain pointers, only the
    // K tail pays for index clamping.
    const int kk = l >> 4, cc = l & 15;
    const bool ia = (i0 + cc) < M, jb = (j0 + cc) < N;
    const double* pa = A + (long)(ia ? i0 + cc : M - 1) * sAi + (long)kk * sAk;
    const double* pb = B + (long)(jb ? j0 + cc : N - 1) * sBj + (long)kk * sBk;
    const long stepA = 4 * sAk, stepB = 4 * sBk;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    int k = 0;
    for (; k + 4 * TP_BATCH <= K; k += 4 * TP_BATCH) {
        double a[TP_BATCH], b[TP_BATCH];
#pragma unroll
        for (int q = 0; q < TP_BATCH; ++q) {
            const double av = ld_sel<PUB>(pa + q * stepA), bv = ld_sel<PUB>(pb + q * stepB);
            a[q] = ia ? av : 0.0;
            b[q] = jb ? bv : 0.0;
        }
#pragma unroll
        for (int q = 0; q < TP_BATCH; q += 2) {
            acc0 = mfma_f64(a[q], b[q], acc0);
            acc1 = mfma_f64(a[q + 1], b[q + 1], acc1);
        }
        pa += TP_BATCH * stepA;
        pb += TP_BATCH * stepB;
    }
    if (k < K) {   // tail batch
        double a[TP_BATCH], b[TP_BATCH];
        const int klast = K - 1 - k - kk;   // last valid k-offset of this lane (may be negative)
#pragma unroll
        for (int q = 0; q < TP_BATCH; ++q) {
            const bool kin = 4 * q <= klast;
            const int qc = kin ? q : 0;
            const bool any = klast >= 0;
            const double av = ld_sel<PUB>(pa + ((any ? qc : 0) * stepA - (any ? 0 : (long)kk * sAk)));
            const double bv = ld_sel<PUB>(pb + ((any ? qc : 0) * stepB - (any ? 0 : (long)kk * sBk)));
            a[q] = (ia && kin) ? av : 0.0;
            b[q] = (jb && kin) ? bv : 0.0;
        }
#pragma unroll
        for (int q = 0; q < TP_BATCH; q += 2) {
            acc0 = mfma_f64(a[q], b[q], acc0);
            acc1 = mfma_f64(a[q + 1], b[q + 1], acc1);
        }
    }
    d4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc0[r] + acc1[r];
    return acc;
}

#define TRSM_MAXBLK 26   // block rows the fallback solve k_trsm_rl keeps in registers (n <= 416)

// =======================================================================================
// Register-resident Cholesky and the square-root form of the Kalman solve
// =======================================================================================
//
// With P = Lf Lf^T (Cholesky of the prior covariance, zero-variance states give zero
// columns) and L_a = Lf[15:n, :], the update of reference src/orcvio.cpp:1682-1753
//     S = H P H^T + s2 I,  K = P H^T S^-1,  dx = K r,  P+ = (I - K H) P
// is, exactly,
//     M  = s2 I + L_a^T A L_a            (A = H^T H, b = H^T r: the compressed block)
//     dx = Lf M^-1 L_a^T b,   P+ = s2 Lf M^-1 Lf^T.
// M is SPD with eigenvalues >= s2 whatever the rank of H, so no rank decision is needed
// (the Gram block A is singular in every update: gauge freedom), and P+ is PSD by
// construction.  Factors are stored as UPPER triangles R (X = R^T R, row-major), so that
// both operands of every trailing update are MFMA accumulator tiles as they stand.

// k_potrf_reg: one workgroup of 8 wavefronts.
//   wave 0 ("chain")   owns every diagonal tile (in LDS) and does nothing but: bring the next diagonal tile up to
//                      date, factor + invert it in one DPP sweep (rows in lanes), publish inv(L11).
//   POTRF_NW workers   hold the off-diagonal tiles of the upper triangle in registers from the first load to the last
//                      store (tiles enumerated row by row and dealt round-robin, so the slots of a worker ascend in the
//                      block row: the panel tiles of a step are a contiguous slot range, the trailing tiles the rest).
//                      Panel tiles become inv(L11) * tile with 4 MFMAs and are published through LDS in the
//                      accumulator layout; trailing tiles subtract panel_a^T panel_b with 4 MFMAs.
//   X  : symmetric input (full diagonal tiles + upper tiles are read), n x n, ldx
//   R  : output upper factor, row-major ldr, FULL 16x16 tiles are stored: R must hold 16*ceil(n/16) rows and columns
//        (lower parts of diagonal tiles and the mirrored tiles are written as 0)
//   Dinv[nb][16][16] : inv(L11) of every diagonal block (generalised inverse on zero pivots)
//   info[0] += pivots <= tol (dropped), info[1] += pivots < -tol (matrix not PSD)
#define POTRF_NW 6
__host__ __device__ inline int potrf_slots_needed(int nb) { return (nb * (nb - 1) / 2 + POTRF_NW - 1) / POTRF_NW; }

// LDS-only workgroup barrier: orders LDS traffic (lgkmcnt) but does not wait for global stores in flight
// (a __syncthreads() would also drain vmcnt, i.e. stall ~1 us per step on the stores of finished tiles).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// "my LDS writes of this step are done": the LDS executes one wavefront's operations in order, so the counter bump only
// has to stay behind them in program order -- a release fence here would also drain the wave's GLOBAL stores (the
// finished tiles on their way to memory), i.e. stall ~1.5 k cycles on every step.
__device__ __forceinline__ void lds_publish_count(int* cnt, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// workgroup barrier that also drains this wave's global stores (publishing waves, one block step after issuing them)
__device__ __forceinline__ void lds_barrier_drain() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// uniform base (SGPR pair) + 32-bit lane offset in bytes: the saddr form of global_store, no 64-bit VALU address math
template <bool PUB>
__device__ __forceinline__ void st_tile(double* ubase, unsigned lane_bytes, double v) {
    st_pub<PUB>(reinterpret_cast<double*>(reinterpret_cast<char*>(ubase) + lane_bytes), v);
}

// PUB: every byte of R and of Dinv is stored sc1, and flag[0] = number of block steps whose row block of R and
// inv(L11) are complete in memory (monotonic, set by one lane behind a workgroup barrier that every storing wave
// reaches after draining its stores).
template <int NSLOT, bool PUB>
#define POTRF_LDS_DOUBLES (816 + 3 * 3584 + 8 + 544)
__device__ __forceinline__ void potrf_reg_body(double* __restrict__ lds, const double* __restrict__ X, int ldx, int n, double tol_rel,
                                               double* __restrict__ R, int ldr, double* __restrict__ Dinv,
                                               int* __restrict__ info, int from_lower, int ablate, int* __restrict__ flag,
                                               unsigned long long* __restrict__ stamps = nullptr, int info_store = 0,
                                               int zero_lower = 1, int rev = 0) {
    // rev: factor the REVERSED matrix X'(i, j) = X(n-1-i, n-1-j) (only with from_lower = 0).  The prior's Cholesky is taken
    // that way: with P = S S^T, S(i, c) = L'(n-1-i, c), the rows of S that belong to the states the measurement rows touch
    // (15 .. n-1) are the FIRST n - 15 rows of the lower-triangular L', so they are zero in the last 15 columns of S --
    // M = s2 I + L_a^T A L_a is then block diagonal, diag(M', s2 I_15), and only the (n-15) x (n-15) block M' needs a
    // factorisation (one block step and a fifth of the flops less at 30 clones).  Loads only: tile (a, b) of X' is read
    // through per-lane offsets from the far corner of X.
    // zero_lower: write zeros to the strictly-lower tiles of R.  0 when the caller keeps them zero itself (the handle's
    // factors: zeroed when the leading dimension changes, never written otherwise).
    // info_store: single-workgroup launches WRITE their two counters (no zeroing launch needed); batched launches add
    // ablate (diagnostic only, scripts/gpu_ablate.py): 1 skip the diagonal sweep, 2 skip trailing MFMAs, 4 skip panel
    // MFMAs, 8 skip the later-diagonal updates.  Results are garbage when non-zero.
    // stamps (diagnostic, scripts/gpu_potrf_stamps.py): core-clock time stamps of wave 0 ([0..63]) and wave 1 ([64..127])
#define POTRF_STAMP(w, idx) do { if (stamps && l == 0 && (w) < 3) stamps[(w) * 64 + (idx)] = clock64(); } while (0)
    // rows of 17 doubles: the row view is read one ROW per lane and inv(L11) one COLUMN block per lane; a stride of 16
    // doubles would put every lane of a 16-lane row on the same two banks
    // LDS (POTRF_LDS_DOUBLES doubles at `lds`: the caller's static buffer, or dynamic LDS when the factorisation
    // shares a launch with other workgroups -- k_front):
    double (*sD)[17] = reinterpret_cast<double (*)[17]>(lds);                         // diagonal tile being factored (row view)
    double (*sDi)[16][17] = reinterpret_cast<double (*)[16][17]>(lds + 272);          // inv(L11) of block step kb in sDi[kb & 1]
    double (*sPan)[4][64] = reinterpret_cast<double (*)[4][64]>(lds + 816);           // published panel tiles, accumulator layout
    double (*sDg)[4][64] = reinterpret_cast<double (*)[4][64]>(lds + 816 + 3584);     // the diagonal tiles (owned by wave 0)
    double (*sStage)[4][64] = reinterpret_cast<double (*)[4][64]>(lds + 816 + 7168);  // block row kb of the trailing matrix, up to date
    int* sCnt = reinterpret_cast<int*>(lds + 816 + 10752);                            // [16] waves that have published their panel tiles of step kb
    double (*sL)[16][17] = reinterpret_cast<double (*)[16][17]>(lds + 816 + 10752 + 8);   // L11 of block step kb (rows) in sL[kb & 1]: wave 4 stores R11 from it
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int kk = l >> 4, cc = l & 15;
    const int nb = (n + 15) >> 4;
    const int ksw = (ablate & 16) ? 0 : (nb > 5 ? (nb - 5) / 2 : 0);   // steps whose later-diagonal updates wave 4 takes over
    POTRF_STAMP(wave == 0 ? 0 : (wave == 1 ? 1 : 3), 0);
    if (stamps && tid == 0) stamps[192] = wall_clock64();
    if (tid < 16) sCnt[tid] = 0;   // first use is behind barrier A of step 0

    if (wave == 0) {
        // =====================================================================================
        // Role 1 -- the critical chain
        // =====================================================================================
        double dmin = INFINITY;   // smallest pivot (-> "some pivot below -tol"; wave 4 counts the dropped ones)
        double mx = 0.0;   // largest diagonal entry -> pivot tolerance
        // Only tile 0 and the n diagonal entries (one gather) stand between the launch and the first sweep: eight loads
        // in flight, one memory latency.  Wave 4 brings the other diagonal tiles into LDS meanwhile.
        {
            double d0[4], dd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = kk + 4 * r, j = cc;
                const bool in = i < n && j < n;
                const double xv = X[(size_t)(in ? (rev ? n - 1 - i : i) : 0) * ldx + (in ? (rev ? n - 1 - j : j) : 0)];
                d0[r] = in ? xv : ((i == j) ? 1.0 : 0.0);   // (rows beyond the matrix: unit diagonal, ordinary pivots)
                const int e = l + 64 * r;
                const double dv = X[(size_t)(e < n ? e : 0) * (ldx + 1)];
                dd[r] = (e < n) ? dv : 0.0;
            }
            int z = 0;
            asm volatile("" : "+v"(z));
#pragma unroll
            for (int r = 0; r < 4; ++r) (&sD[0][0] + z)[(kk + 4 * r) * 17 + cc] = d0[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmax(mx, dd[r]);
        }
        double tol = 0.0;
        if (tol_rel > 0.0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
            tol = tol_rel * mx;
        }
        double v[16];   // rows of the most recent L11 (lane (l & 15) = row), kept until its R11 tile has been written
        // factor + invert the tile that sits in sD (row view): v <- L11, inv(L11) -> sDi[kb & 1]
        auto sweep_tile = [&](int kb) {
            int z = 0;
            asm volatile("" : "+v"(z));   // opaque zero: keeps LDS address arithmetic out of loop-invariant hoisting
            const double* pD = &sD[0][0] + z;
            double* pDi = &sDi[0][0][0] + z + (kb & 1) * 272;
            double y[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double a = pD[cc * 17 + c];
                v[c] = (c <= cc) ? a : 0.0;
                y[c] = (c == cc) ? 1.0 : 0.0;
            }
            POTRF_STAMP(2, 3 * kb);
            if (!(ablate & 1)) DiagStep<0, !PUB>::run(v, y, tol, dmin);   // (PUB: the matrix is M, positive definite by construction)
            POTRF_STAMP(2, 3 * kb + 1);
            if (l < 16) {
                double* pL = &sL[0][0][0] + z + (kb & 1) * 272;
#pragma unroll
                for (int c = 0; c < 16; ++c) pDi[c * 17 + l] = y[c];   // Linv[c][l]
#pragma unroll
                for (int c = 0; c < 16; ++c) pL[l * 17 + c] = v[c];    // L11[l][c] (zero above the diagonal)
            }
        };
        // wave 4 takes inv(L11) and L11 (as R11) of every step from LDS to memory: no global stores of them, and no
        // transposition, on this chain
        __builtin_amdgcn_s_setprio(3);
        POTRF_STAMP(0, 1);
        wave_sync();
        sweep_tile(0);
        POTRF_STAMP(0, 2);
        const unsigned lane_b0 = (unsigned)((kk * ldr + cc) * 8);
        for (int kb = 0; kb < nb; ++kb) {
            // A: the workers have finished the trailing update of step kb-1 (block row kb staged, later diagonal tiles
            // up to date) and see inv(L11) of step kb.  It is the only point where the chain waits for anybody.
            if (PUB) lds_barrier_drain(); else lds_barrier();
            POTRF_STAMP(0, 3 + 4 * kb);
            const int kn = kb + 1;
            if (kn < nb) {
                int z = 0;
                asm volatile("" : "+v"(z));
                // the chain computes the one panel tile its next diagonal tile needs, (kb, kb+1), itself: no hand-off
                const double* pDi = &sDi[0][0][0] + z + (kb & 1) * 272;
                const double* st = &sStage[0][0][0] + z + kn * 256 + l;
                const double* pGn = &sDg[0][0][0] + z + kn * 256 + l;
                double li[4], sv[4];
                d4 t;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { li[s4] = pDi[cc * 17 + kk + 4 * s4]; sv[s4] = st[s4 * 64]; t[s4] = pGn[s4 * 64]; }
                d4 x = {0, 0, 0, 0};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(li[s4], sv[s4], x);
                // next diagonal tile: -= panel^T panel (operands straight from the accumulator registers), straight into
                // the row view for its sweep.  Eight dependent MFMAs back to back; everything else of the step is issued
                // while they run: the panel tile to LDS / memory, the step counter.
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) t = mfma_f64(-x[s4], x[s4], t);
                __builtin_amdgcn_sched_barrier(0);
                double* pPan = &sPan[0][0][0] + z + kn * 256 + l;
                double* ub = R + (size_t)(16 * kb) * ldr + 16 * kn;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pPan[r * 64] = x[r];
                    st_tile<PUB>(ub + (size_t)(4 * r) * ldr, lane_b0, x[r]);
                }
                lds_publish_count(&sCnt[kb], l);
                POTRF_STAMP(0, 4 + 4 * kb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) (&sD[0][0] + z)[(kk + 4 * r) * 17 + cc] = t[r];
                wave_sync();
                POTRF_STAMP(0, 5 + 4 * kb);
                sweep_tile(kn);
                POTRF_STAMP(0, 6 + 4 * kb);
                // the later diagonal tiles (k > kn) are brought up to date by the workers (they live in LDS)
            } else {
                lds_publish_count(&sCnt[kb], l);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (l == 0) {   // info[1]: some pivot below -tol (the matrix is not positive semi-definite); info[0] is wave 4's
            // (PUB: M = s2 I + ... is factored without a pivot test -- any pivot that is not positive is a failure)
            const int nneg = PUB ? (!(dmin > 0.0) ? 1 : 0) : ((dmin < -tol) ? 1 : 0);
            if (info_store) info[1] = nneg;
            else if (nneg) atomicAdd(&info[1], nneg);
        }
    } else if (wave == 4) {
        // =====================================================================================
        // Role 3 -- wave 4 shares SIMD 0 with the chain wave (waves are dealt to the four SIMDs round-robin): no MFMA
        // work here, it only writes inv(L11) and R11 of every step from LDS to memory.
        // =====================================================================================
        const unsigned lane_b = (unsigned)((kk * ldr + cc) * 8);
        {   // diagonal tiles 1.. -> sDg (first needed behind barrier A of step 0); all loads in flight before the first write
            double dg[13][4];
#pragma unroll
            for (int k = 1; k < 14; ++k) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * k + kk + 4 * r, j = 16 * k + cc;
                    const bool in = k < nb && i < n && j < n;
                    const double xv = X[(size_t)(in ? (rev ? n - 1 - i : i) : 0) * ldx + (in ? (rev ? n - 1 - j : j) : 0)];
                    dg[k - 1][r] = in ? xv : ((i == j) ? 1.0 : 0.0);   // (rows beyond the matrix: unit diagonal)
                }
            }
#pragma unroll
            for (int k = 1; k < 14; ++k) {
                if (k < nb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sDg[k][r][l] = dg[k - 1][r];
                }
            }
            POTRF_STAMP(2, 50);   // (wave 4: the diagonal tiles are in LDS)
        }
        int ndrop = 0;   // dropped pivots (zero diagonal entries of L), counted here: off the chain
        for (int kb = 0; kb < nb; ++kb) {
            int z = 0;
            asm volatile("" : "+v"(z));
            if (PUB) lds_barrier_drain(); else lds_barrier();   // A
            const double* pDi = &sDi[0][0][0] + z + (kb & 1) * 272;
#pragma unroll
            for (int r = 0; r < 4; ++r) st_pub<PUB>(Dinv + (size_t)kb * 256 + l + 64 * r, pDi[(kk + 4 * r) * 17 + cc]);   // element l + 64 r = (row kk + 4r, column cc)
            // R11 = L11^T from the row copy the sweep left in sL[kb & 1] (written before this barrier)
            const double* pL = &sL[0][0][0] + z + (kb & 1) * 272;
            double* ub = R + (size_t)(16 * kb) * ldr + 16 * kb;
#pragma unroll
            for (int r = 0; r < 4; ++r) st_tile<PUB>(ub + (size_t)(4 * r) * ldr, lane_b, (kk + 4 * r <= cc) ? pL[cc * 17 + kk + 4 * r] : 0.0);
            {
                const bool dropped = l < 16 && pL[(l & 15) * 18] == 0.0;   // L11[l][l] (rows beyond the matrix hold 1)
                ndrop += __builtin_popcountll(__ballot(dropped));
            }
            if (kb < ksw && !(ablate & 8)) {
                // the later diagonal tiles of the early steps (the workers are the bottleneck there): -= panel^T panel once
                // the panel tiles of this step are all in LDS; two tiles in flight
                while (__hip_atomic_load(&sCnt[kb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < POTRF_NW + 1) __builtin_amdgcn_s_sleep(2);
                const double* pPan = &sPan[0][0][0] + z + l;
                double* pG = &sDg[0][0][0] + z + l;
#pragma unroll 1
                for (int k = kb + 2; k < nb; k += 2) {
                    const bool two = k + 1 < nb;
                    const int k1 = two ? k + 1 : k;
                    double qa[4], qb[4];
                    d4 t0, t1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { qa[r] = pPan[k * 256 + r * 64]; t0[r] = pG[k * 256 + r * 64]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { qb[r] = pPan[k1 * 256 + r * 64]; t1[r] = pG[k1 * 256 + r * 64]; }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        t0 = mfma_f64(-qa[s4], qa[s4], t0);
                        t1 = mfma_f64(-qb[s4], qb[s4], t1);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) pG[k * 256 + r * 64] = t0[r];
                    if (two) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pG[k1 * 256 + r * 64] = t1[r];
                    }
                }
            }
        }
        if (l == 0) {   // info[0]: dropped pivots (zero-variance states of a prior, rank-deficient directions of a Gram)
            if (info_store) info[0] = ndrop;
            else if (ndrop) atomicAdd(&info[0], ndrop);
        }
    } else {
        // =====================================================================================
        // Role 2 -- workers (waves 1,2,3,5,6,7).  Tile t = T(a) + (b - a - 1), T(a) = a nb - a(a+1)/2, of worker t % NW, slot t / NW.
        // =====================================================================================
        const int wi = (wave < 4) ? wave - 1 : wave - 2;   // 0 .. POTRF_NW-1 (wave-uniform, in an SGPR)
        d4 acc[NSLOT];
        int tab[NSLOT];   // packed a | b << 8 (wave-uniform), -1 if the slot is empty
        const unsigned lane_b = (unsigned)((kk * ldr + cc) * 8);   // lane part of every R address, bytes
        // Loads: uniform tile base (SGPRs) + one 32-bit lane offset, so a slot costs a handful of instructions and all
        // loads of the wave are in flight within a few hundred cycles.  Only the last block column can reach past n
        // (rows of an off-diagonal tile never do: a <= nb - 2): its lanes use a clamped column and select 0.
        const int ce = n - 1 - 16 * (nb - 1);   // last valid column inside block column nb - 1
        const bool cin = cc <= ce;
        const int ccE = cin ? cc : ce;
        const unsigned loN = (unsigned)(from_lower ? (cc * ldx + kk) : (kk * ldx + cc)) * 8u;
        const unsigned loE = (unsigned)(from_lower ? (ccE * ldx + kk) : (kk * ldx + ccE)) * 8u;
        const unsigned rstep = from_lower ? 32u : 32u * (unsigned)ldx;   // bytes between accumulator rows r -> r + 1
        // The (block row, block column) of every slot is decoded by ONE lane each (lane s <-> slot s, tile index
        // t = s NW + wi: row a is the largest a with T(a) <= t, T(a) = a (nb-1) - a (a-1)/2) and read back with
        // v_readlane: a per-slot scalar search loop plus 64-bit address arithmetic was ~5 k cycles of issue per wavefront.
        int mytab = -1;
        {
            const int t = l * POTRF_NW + wi;
            if (l < NSLOT && t < nb * (nb - 1) / 2) {
                const float bq = 2.0f * nb - 1.0f;
                int a = (int)((bq - __builtin_sqrtf(bq * bq - 8.0f * t)) * 0.5f);
                a = a < 0 ? 0 : (a > nb - 2 ? nb - 2 : a);
                while (a > 0 && a * (nb - 1) - a * (a - 1) / 2 > t) --a;
                while ((a + 1) * (nb - 1) - (a + 1) * a / 2 <= t) ++a;
                mytab = a | ((a + 1 + (t - (a * (nb - 1) - a * (a - 1) / 2))) << 8);
            }
        }
        unsigned loNr[4], loEr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { loNr[r] = loN + r * rstep; loEr[r] = loE + r * rstep; }
        if (rev) {   // element (kk + 4r, cc) of tile (ta, tb) of X' is X(n-1 - 16 ta - kk - 4r, n-1 - 16 tb - cc): the lane's
                     // offset from X counts DOWN from the far corner, the tile's (uniform) offset is subtracted from it
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                loNr[r] = (unsigned)((n - 1 - kk - 4 * r) * ldx + (n - 1 - cc)) * 8u;
                loEr[r] = (unsigned)((n - 1 - kk - 4 * r) * ldx + (n - 1 - ccE)) * 8u;
            }
        }
        const char* Xb = reinterpret_cast<const char*>(X);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            tab[s] = __builtin_amdgcn_readlane(mytab, s);
            acc[s] = d4{0.0, 0.0, 0.0, 0.0};
            if (tab[s] >= 0) {   // (wave-uniform)
                const int ta = tab[s] & 255, tb = tab[s] >> 8;
                const unsigned off = (unsigned)(from_lower ? (16 * tb) * ldx + 16 * ta : (16 * ta) * ldx + 16 * tb) * 8u;
                if (rev) {
                    if (tb == nb - 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[s][r] = *reinterpret_cast<const double*>(Xb + (size_t)(loEr[r] - off));
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[s][r] = *reinterpret_cast<const double*>(Xb + (size_t)(loNr[r] - off));
                    }
                } else if (tb == nb - 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[s][r] = *reinterpret_cast<const double*>(Xb + (size_t)(off + loEr[r]));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[s][r] = *reinterpret_cast<const double*>(Xb + (size_t)(off + loNr[r]));
                }
            }
        }
        POTRF_STAMP(wave == 1 ? 1 : 3, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const bool dead = tab[s] < 0, edge = (tab[s] >> 8) == nb - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[s][r] = (dead || (edge && !cin)) ? 0.0 : acc[s][r];
        }
        POTRF_STAMP(wave == 1 ? 1 : 3, 56);   // every tile of this wave has arrived
        // every load of this wave is in flight before the first use (loads and stores share one in-order counter)
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            if (tab[s] >= 0 && (tab[s] & 255) == 0) {   // block row 0 is the first panel
#pragma unroll
                for (int r = 0; r < 4; ++r) sStage[tab[s] >> 8][r][l] = acc[s][r];
            }
        }
        if (zero_lower) {   // the mirrored (strictly lower) tile of the output is never touched again
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                if (tab[s] >= 0) {
                    double* ub = R + (size_t)(16 * (tab[s] >> 8)) * ldr + 16 * (tab[s] & 255);
#pragma unroll
                    for (int r = 0; r < 4; ++r) st_tile<false>(ub + (size_t)(4 * r) * ldr, lane_b, 0.0);
                }
            }
        }
        POTRF_STAMP(wave == 1 ? 1 : 3, 57);   // row 0 staged
        auto slots_below = [&](int x) -> int {   // slots of this worker whose tile index is < x
            const int v = x - wi;
            return v <= 0 ? 0 : (v + POTRF_NW - 1) / POTRF_NW;
        };
        for (int kb = 0; kb < nb; ++kb) {
            int z = 0;
            asm volatile("" : "+v"(z));
            double* pDi = &sDi[0][0][0] + z + (kb & 1) * 272;
            double* pPan = &sPan[0][0][0] + z;
            POTRF_STAMP(wave == 1 ? 1 : 3, 2 + 4 * kb);
            if (PUB) {
                lds_barrier_drain();   // A (+ the stores of step kb-1 of this wave have landed)
                if (wave == 1 && l == 0 && kb > 0) __hip_atomic_store(flag, kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                lds_barrier();   // A
            }
            POTRF_STAMP(wave == 1 ? 1 : 3, 3 + 4 * kb);
            // ---- panel tiles (a == kb < b): tile <- inv(L11) * tile, published and stored ---------------
            const int Tkb = kb * nb - kb * (kb + 1) / 2;
            const int s_lo = slots_below(Tkb), s_hi = slots_below(Tkb + (nb - 1 - kb));
            // The panel row was staged in LDS by the trailing update of the previous step (sStage), so the panel is a
            // rolled loop that any worker can run on any tile: tile kb+2+wi, +NW, ...
            (void)s_lo;
            if (kb + 2 + wi < nb && !(ablate & 32)) {   // tile (kb, kb+1) is the chain's
                // at most two tiles per worker (nb <= 14): both operand sets are fetched before the first MFMA
                const int b0 = kb + 2 + wi;
                const bool two = b0 + POTRF_NW < nb;
                const int b1 = two ? b0 + POTRF_NW : b0;
                double li[4], sa[4], sb[4];
                const double* st0 = &sStage[0][0][0] + z + b0 * 256 + l;
                const double* st1 = &sStage[0][0][0] + z + b1 * 256 + l;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { li[s4] = pDi[cc * 17 + kk + 4 * s4]; sa[s4] = st0[s4 * 64]; }   // A operand: Linv[m = cc][k = kk + 4 s4]
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) sb[s4] = st1[s4 * 64];
                double* urow = R + (size_t)(16 * kb) * ldr;
                d4 x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
                if (!(ablate & 4)) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) x0 = mfma_f64(li[s4], sa[s4], x0);
                    if (two) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) x1 = mfma_f64(li[s4], sb[s4], x1);
                    }
                }
                {
                    double* dst = pPan + b0 * 256 + l;
                    double* ub = urow + 16 * b0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[r * 64] = x0[r];
                        st_tile<PUB>(ub + (size_t)(4 * r) * ldr, lane_b, x0[r]);   // final: row block kb of R
                    }
                }
                if (two) {
                    double* dst = pPan + b1 * 256 + l;
                    double* ub = urow + 16 * b1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[r * 64] = x1[r];
                        st_tile<PUB>(ub + (size_t)(4 * r) * ldr, lane_b, x1[r]);
                    }
                }
            }
            POTRF_STAMP(wave == 1 ? 1 : 3, 4 + 4 * kb);
            // B: every panel tile of step kb is in LDS -- a counter, not a barrier: the chain only adds to it
            lds_publish_count(&sCnt[kb], l);
            {
                const int target = POTRF_NW + 1;   // the workers and the chain (which also writes R11 of this step before it adds)
                while (__hip_atomic_load(&sCnt[kb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
            }
            POTRF_STAMP(wave == 1 ? 1 : 3, 5 + 4 * kb);
            // ---- trailing tiles (a > kb): tile -= panel_a^T panel_b.  The FP64 MFMA issues once per 64 cycles, which is
            // also its latency, so one dependent chain per tile runs at the full rate of the pipe; the operands of the
            // next live slot are fetched from LDS while the MFMAs of the current one execute.  Tiles of block row
            // kb+1 (the next panel) are also staged in LDS.
            // later diagonal tiles k > kb+1 live in LDS and are off the critical chain: worker k % NW updates them (at
            // most two per worker); operands of both are fetched before the first MFMA.  In the first ksw steps the
            // workers are the bottleneck and the chain's SIMD has time to spare: wave 4 does these updates there.
            if (!(ablate & 8) && kb >= ksw) {
                int d0 = (wi - (kb + 2)) % POTRF_NW;
                if (d0 < 0) d0 += POTRF_NW;
                const int k0 = kb + 2 + d0;
                if (k0 < nb) {
                    const bool two = k0 + POTRF_NW < nb;
                    const int k1 = two ? k0 + POTRF_NW : k0;
                    const double* qp0 = pPan + k0 * 256 + l;
                    const double* qp1 = pPan + k1 * 256 + l;
                    double* g0 = &sDg[0][0][0] + z + k0 * 256 + l;
                    double* g1 = &sDg[0][0][0] + z + k1 * 256 + l;
                    double qa[4], qb[4];
                    d4 t0, t1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { qa[r] = qp0[r * 64]; t0[r] = g0[r * 64]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { qb[r] = qp1[r * 64]; t1[r] = g1[r * 64]; }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) t0 = mfma_f64(-qa[s4], qa[s4], t0);
                    if (two) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) t1 = mfma_f64(-qb[s4], qb[s4], t1);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) g0[r * 64] = t0[r];
                    if (two) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) g1[r * 64] = t1[r];
                    }
                }
            }
            if (!(ablate & 2)) {
                double q[2][8];
                auto fetch = [&](double (&dst)[8], int tb) {
                    const double* qa = pPan + (tb & 255) * 256 + l;
                    const double* qb = pPan + (tb >> 8) * 256 + l;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) { dst[s4] = qa[s4 * 64]; dst[4 + s4] = qb[s4 * 64]; }
                };
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    if (s >= s_hi && tab[s] >= 0) {
                        if (s == s_hi) fetch(q[s & 1], tab[s]);
                        // the fetch of the next slot is unconditional (an empty slot reads tile 0): the wait in front of
                        // the MFMAs can then count it, and leaves it in flight
                        if (s + 1 < NSLOT) { const int tn = tab[(s + 1 < NSLOT) ? s + 1 : s]; fetch(q[(s + 1) & 1], tn >= 0 ? tn : 0); }
                        __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the fetch below the MFMAs)
                        d4 x = acc[s];
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(-q[s & 1][s4], q[s & 1][4 + s4], x);
                        acc[s] = x;
                        if ((tab[s] & 255) == kb + 1) {
                            double* st = &sStage[0][0][0] + z + (tab[s] >> 8) * 256 + l;
#pragma unroll
                            for (int r = 0; r < 4; ++r) st[r * 64] = x[r];
                        }
                    }
                }
            }
        }
    }
    POTRF_STAMP(wave == 0 ? 0 : (wave == 1 ? 1 : 3), 63);
    if (stamps && tid == 0) stamps[193] = wall_clock64();
    if (PUB) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wave == 1 && l == 0) __hip_atomic_store(flag, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#undef POTRF_STAMP
}

template <int NSLOT>
__global__ __launch_bounds__(512) void k_potrf_reg(const double* __restrict__ X, int ldx, int n, double tol_rel,
                                                   double* __restrict__ R, int ldr, double* __restrict__ Dinv,
                                                   int* __restrict__ info, unsigned long long* __restrict__ stamps = nullptr,
                                                   size_t strideX = 0, size_t strideR = 0, size_t strideD = 0,
                                                   int from_lower = 0, int ablate = 0, int zero_lower = 1, int rev = 0) {
    // batched use: workgroup b factors X + b*strideX into R + b*strideR (Dinv + b*strideD);
    // from_lower: the symmetric input has only its lower tiles filled, read element (i,j) as (j,i)
    __shared__ __attribute__((aligned(16))) double sPotrf[POTRF_LDS_DOUBLES];
    potrf_reg_body<NSLOT, false>(sPotrf, X + (size_t)blockIdx.x * strideX, ldx, n, tol_rel, R + (size_t)blockIdx.x * strideR, ldr,
                                 Dinv + (size_t)blockIdx.x * strideD, info, from_lower, ablate, nullptr, stamps,
                                 gridDim.x == 1 ? 1 : 0, zero_lower, rev);
}

#include "potrf_lookahead.hpp"   // potrf_la_chain_wg / potrf_la_far_wg: the factorisation with its trailing update spread over far workgroups

// k_gemm_asmA: C (M x Nc) = A (M x K) * B (K x Nc) where A is NOT in memory: entry (i,k) of the compressed block is
// assembled on the fly as k_assemble_A would, scatter(S)(i,k) - sum_c Gpart[c](i,k).  Used for U = [A; b^T] L_a right
// behind k_front, whose in-launch compression then ends with the Grams (no second device-wide barrier, no assembly
// pass).  Same tiling as k_gemm: one workgroup per 16x16 tile of C, split-K over its four wavefronts; every load of a
// batch (one S tile entry, <= 4 partial-Gram entries and one B entry per k-step) is issued before the first use.
struct AsmArgs {
    const double* S; int N, cb0, NA, NAP; const double* parts; int nparts; size_t stride; int dbg;
    const double* plus;   // optional Gram added to A (EKF-SLAM rows), lower tiles valid
};
// One 16 x 16 tile of C = [A; b^T] B with the strip of A assembled in LDS: the body of k_gemm_asmA, 256 threads (tid), LDS handed in
// (sA: 16 x 193, sPartF: 3 x 4 x 64, sSharedF: 4 x 64 doubles).  PUB: the Grams, the clone tiles and B were stored by other workgroups
// of THIS launch (k_front with the product inside, FrontUArgs): read past the caches.  !active: the barriers only (no store).
constexpr int ASM_LDS_DOUBLES = 16 * 193 + 3 * 4 * 64 + 4 * 64;
// PLUSW: aa.plus is being written by ANOTHER stream (k_gemm_asmA_w): its loads stand behind a poll of the completion word `wait`, issued
// once every other load of the strip is in flight, and go past the caches (agent-scope loads: no acquire fence, nothing else is re-read).
template <bool PUB, bool PLUSW = false>
__device__ __forceinline__ void asm_gemm_tile(const AsmArgs& aa, const double* __restrict__ B, long sBk, long sBj, int M, int Nc, int K,
                                              double* __restrict__ C, long sCi, long sCj, const int tile_index, const bool active, const int tid,
                                              double* __restrict__ sA, double* __restrict__ sPartF, double* __restrict__ sSharedF,
                                              const unsigned* __restrict__ wait = nullptr, unsigned expect = 0u, int spin_limit = 0, int* __restrict__ lost = nullptr) {
    constexpr int KMAX = 192, LDA = KMAX + 1;   // (193: one row per lane group without bank conflicts)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int kk = l >> 4, cc = l & 15;
    const int ntj = (Nc + 15) >> 4;
    const int bi = tile_index / ntj, bj = tile_index - bi * ntj;
    const int KS = ((K + 15) >> 4) << 2;   // k-slice per wavefront, a multiple of the MFMA depth (<= 48 for K <= 192)
    const int kbeg = wave * KS;
    const int kend = (kbeg + KS < K) ? kbeg + KS : K;
    // B operands of this wavefront's slice: in flight while the strip is assembled
    constexpr int GB = 12;
    const int jc = 16 * bj + cc;
    const bool jb = jc < Nc;
    const double* pb = B + (long)(jb ? jc : Nc - 1) * sBj;
    double bv[GB];
#pragma unroll
    for (int q = 0; q < GB; ++q) {
        const int k = kbeg + 4 * q + kk;
        bv[q] = ld_sel<PUB>(pb + (long)(k < kend ? k : (kend > 0 ? kend - 1 : 0)) * sBk);
    }
    // ---- the (ext|r) x (ext|r) entries sum EVERY clone tile: once per workgroup that owns such rows, 64 entries x N
    //      tiles over the 256 threads (clone c on thread group c % 4, then a fixed-order sum of the four partials)
    const bool has_shared = (bi == 0 || bi == (aa.NA >> 4)) && aa.N > 0;   // rows 0..6 or row NA
    if (has_shared) {
        const int g4 = tid >> 6, en = tid & 63;
        const int ea = en >> 3, eb = en & 7;                      // entry classes 0..6 -> e = 0..6, 7 -> e = 13
        const int e0 = ea == 7 ? 13 : ea, e1 = eb == 7 ? 13 : eb;
        const int e16 = (e0 >= e1) ? e0 * 16 + e1 : e1 * 16 + e0;
        double acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = g4; c < aa.N; c += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cu = c + 4 * u;
                const double v = ld_sel<PUB>(aa.S + (size_t)(cu < aa.N ? cu : aa.N - 1) * 256 + e16);
                acc8[u] += cu < aa.N ? v : 0.0;
            }
        }
        sSharedF[g4 * 64 + en] = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
    }
    // ---- strip: thread (wave w, lane l) owns rows w, w+4, w+8, w+12 and columns l, l+64, l+128 -----------------------
    double sv[4][3], gv[4][3][4], pv[4][3];
    int cls[4][3];   // 0: no S contribution, 1: one clone tile (sv), 2: shared entry (index in cls >> 2)
    int ekj[3], ckj[3];   // column classes: three per thread, shared by its four rows
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = l + 64 * j;
        const int kc = k < K ? k : 0;
        ekj[j] = -1; ckj[j] = -1;
        if (kc < 7) ekj[j] = kc; else if (kc >= aa.cb0 && kc < aa.cb0 + 6 * aa.N) { ckj[j] = (kc - aa.cb0) / 6; ekj[j] = 7 + (kc - aa.cb0) - 6 * ckj[j]; }
    }
    // (32-bit element offsets from wave-uniform bases: the loads take the SGPR-base form, and the row part of every
    //  index is scalar -- the per-element code is a dozen instructions, not a hundred)
    const bool use_S = aa.N > 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave + 4 * rr;
        const int i = 16 * bi + r;          // (wave-uniform)
        const bool rin = i < M;
        const int ic = rin ? i : 0;
        int ei = -1, ci = -1;
        if (ic < 7) ei = ic; else if (ic == aa.NA) ei = 13; else if (ic >= aa.cb0 && ic < aa.cb0 + 6 * aa.N) { ci = (ic - aa.cb0) / 6; ei = 7 + (ic - aa.cb0) - 6 * ci; }
        const unsigned rowoff = (unsigned)(ic * aa.NAP);
        const int ea = ei == 13 ? 7 : ei;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = l + 64 * j;
            const bool in = rin && k < K;
            const unsigned kc = k < K ? (unsigned)k : 0u;
            const int ek = ekj[j], ck = ckj[j];
            // clone tile and kind of S contribution: 0 none, 1 one clone tile, 2 every clone tile (shared block)
            int c0 = ci >= 0 ? ci : ck, n1 = 0;
            if (ei >= 0 && ek >= 0) n1 = (ci < 0 && ck < 0) ? 2 : ((ci >= 0 && ck >= 0 && ci != ck) ? 0 : 1);
            if (!in || !use_S) n1 = 0;
            const int e16 = (ei >= ek) ? ei * 16 + ek : ek * 16 + ei;   // the S tiles hold both triangles: [max][min]
            cls[rr][j] = n1 == 2 ? (2 | ((ea * 8 + (ek == 13 ? 7 : ek)) << 2)) : n1;
            sv[rr][j] = ld_sel<PUB>(aa.S + (n1 == 1 ? (unsigned)(c0 * 256 + e16) : 0u));
            const unsigned src = rowoff + kc;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < aa.nparts) gv[rr][j][u] = ld_sel<PUB>(aa.parts + (size_t)u * aa.stride + src);   // (wave-uniform count and base)
            pv[rr][j] = 0.0;
            if (!PLUSW && aa.plus) pv[rr][j] = aa.plus[((ic >> 4) >= (int)(kc >> 4)) ? src : kc * (unsigned)aa.NAP + (unsigned)ic];
        }
    }
    if constexpr (PLUSW) {
        if (tid == 0) {
            int it = 0;
            while ((int)(__hip_atomic_load(wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expect) < 0 && it < spin_limit) { __builtin_amdgcn_s_sleep(4); ++it; }
            if (it >= spin_limit) atomicAdd(lost, 1);
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 16 * bi + wave + 4 * rr;
            const unsigned ic = i < M ? (unsigned)i : 0u;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int k = l + 64 * j;
                const unsigned kc = k < K ? (unsigned)k : 0u;
                pv[rr][j] = ld_pub(aa.plus + (((ic >> 4) >= (kc >> 4)) ? ic * (unsigned)aa.NAP + kc : kc * (unsigned)aa.NAP + ic));
            }
        }
    }
    if (PUB || has_shared) __syncthreads();   // (workgroup-uniform in k_gemm_asmA: one tile per workgroup; the two teams of k_front hold different tiles)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave + 4 * rr;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = l + 64 * j;
            double sS = cls[rr][j] == 1 ? sv[rr][j] : 0.0;
            if (has_shared && (cls[rr][j] & 3) == 2) {
                const int en = cls[rr][j] >> 2;
                sS = (sSharedF[en] + sSharedF[64 + en]) + (sSharedF[128 + en] + sSharedF[192 + en]);
            }
            double g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = u < aa.nparts ? gv[rr][j][u] : 0.0;
            const bool in = 16 * bi + r < M && k < K;
            sA[r * LDA + k] = in ? sS - ((g[0] + g[1]) + (g[2] + g[3])) + pv[rr][j] : 0.0;
        }
    }
    __syncthreads();
    // ---- split-K product from the strip ---------------------------------------------------------------------------
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    {
        double a[GB];
#pragma unroll
        for (int q = 0; q < GB; ++q) {
            const int k = kbeg + 4 * q + kk;
            const bool kin = k < kend;
            a[q] = kin ? sA[cc * LDA + (kin ? k : 0)] : 0.0;
            bv[q] = (jb && kin) ? bv[q] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < GB; q += 2) {
            acc0 = mfma_f64(a[q], bv[q], acc0);
            acc1 = mfma_f64(a[q + 1], bv[q + 1], acc1);
        }
    }
    d4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc0[r] + acc1[r];
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sPartF[((wave - 1) * 4 + r) * 64 + l] = acc[r];
    }
    __syncthreads();
    if (wave > 0 || !active) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = ((acc[r] + sPartF[r * 64 + l]) + sPartF[(4 + r) * 64 + l]) + sPartF[(8 + r) * 64 + l];
        const int io = 16 * bi + kk + 4 * r, jo = 16 * bj + cc;
        if (io < M && jo < Nc) C[(long)io * sCi + (long)jo * sCj] = v;
    }
}

// (k_gemm_asmA keeps its own copy of the body below rather than calling asm_gemm_tile<false>: through the shared function the compiler
//  allocates it differently -- 66 spilled SGPRs instead of 7 -- and the launch takes 8.65 instead of 8.3 us; asm_gemm_tile serves the
//  opt-in in-launch form of k_front only.  The two must stay the same arithmetic: test_U_inside_k_front_is_bit_identical_...)
__global__ __launch_bounds__(256) void k_gemm_asmA(AsmArgs aa, const double* __restrict__ B, long sBk, long sBj, int M, int Nc, int K,
                                                   double* __restrict__ C, long sCi, long sCj, int* __restrict__ clear) {
    // The 16 x K strip of A this tile needs is assembled into LDS first, with the k index along the lanes (coalesced
    // reads of the partial Grams, whose two triangles are both written by k_front), then read back as MFMA operands.
    if (clear && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(clear, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int KMAX = 192, LDA = KMAX + 1;   // (193: one row per lane group without bank conflicts)
    __shared__ double sA[16 * LDA];
    __shared__ double sPart[3][4][64];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int kk = l >> 4, cc = l & 15;
    const int ntj = (Nc + 15) >> 4;
    const int bi = (int)blockIdx.x / ntj, bj = (int)blockIdx.x - bi * ntj;
    const int KS = ((K + 15) >> 4) << 2;   // k-slice per wavefront, a multiple of the MFMA depth (<= 48 for K <= 192)
    const int kbeg = wave * KS;
    const int kend = (kbeg + KS < K) ? kbeg + KS : K;
    // B operands of this wavefront's slice: in flight while the strip is assembled
    constexpr int GB = 12;
    const int jc = 16 * bj + cc;
    const bool jb = jc < Nc;
    const double* pb = B + (long)(jb ? jc : Nc - 1) * sBj;
    double bv[GB];
#pragma unroll
    for (int q = 0; q < GB; ++q) {
        const int k = kbeg + 4 * q + kk;
        bv[q] = pb[(long)(k < kend ? k : (kend > 0 ? kend - 1 : 0)) * sBk];
    }
    // ---- the (ext|r) x (ext|r) entries sum EVERY clone tile: once per workgroup that owns such rows, 64 entries x N
    //      tiles over the 256 threads (clone c on thread group c % 4, then a fixed-order sum of the four partials)
    __shared__ double sShared[4][64];
    const bool has_shared = (bi == 0 || bi == (aa.NA >> 4)) && aa.N > 0;   // rows 0..6 or row NA
    if (has_shared) {
        const int g4 = tid >> 6, en = tid & 63;
        const int ea = en >> 3, eb = en & 7;                      // entry classes 0..6 -> e = 0..6, 7 -> e = 13
        const int e0 = ea == 7 ? 13 : ea, e1 = eb == 7 ? 13 : eb;
        const int e16 = (e0 >= e1) ? e0 * 16 + e1 : e1 * 16 + e0;
        double acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = g4; c < aa.N; c += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cu = c + 4 * u;
                const double v = aa.S[(size_t)(cu < aa.N ? cu : aa.N - 1) * 256 + e16];
                acc8[u] += cu < aa.N ? v : 0.0;
            }
        }
        sShared[g4][en] = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
    }
    // ---- strip: thread (wave w, lane l) owns rows w, w+4, w+8, w+12 and columns l, l+64, l+128 -----------------------
    double sv[4][3], gv[4][3][4], pv[4][3];
    int cls[4][3];   // 0: no S contribution, 1: one clone tile (sv), 2: shared entry (index in cls >> 2)
    int ekj[3], ckj[3];   // column classes: three per thread, shared by its four rows
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = l + 64 * j;
        const int kc = k < K ? k : 0;
        ekj[j] = -1; ckj[j] = -1;
        if (kc < 7) ekj[j] = kc; else if (kc >= aa.cb0 && kc < aa.cb0 + 6 * aa.N) { ckj[j] = (kc - aa.cb0) / 6; ekj[j] = 7 + (kc - aa.cb0) - 6 * ckj[j]; }
    }
    // (32-bit element offsets from wave-uniform bases: the loads take the SGPR-base form, and the row part of every
    //  index is scalar -- the per-element code is a dozen instructions, not a hundred)
    const bool use_S = aa.N > 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave + 4 * rr;
        const int i = 16 * bi + r;          // (wave-uniform)
        const bool rin = i < M;
        const int ic = rin ? i : 0;
        int ei = -1, ci = -1;
        if (ic < 7) ei = ic; else if (ic == aa.NA) ei = 13; else if (ic >= aa.cb0 && ic < aa.cb0 + 6 * aa.N) { ci = (ic - aa.cb0) / 6; ei = 7 + (ic - aa.cb0) - 6 * ci; }
        const unsigned rowoff = (unsigned)(ic * aa.NAP);
        const int ea = ei == 13 ? 7 : ei;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = l + 64 * j;
            const bool in = rin && k < K;
            const unsigned kc = k < K ? (unsigned)k : 0u;
            const int ek = ekj[j], ck = ckj[j];
            // clone tile and kind of S contribution: 0 none, 1 one clone tile, 2 every clone tile (shared block)
            int c0 = ci >= 0 ? ci : ck, n1 = 0;
            if (ei >= 0 && ek >= 0) n1 = (ci < 0 && ck < 0) ? 2 : ((ci >= 0 && ck >= 0 && ci != ck) ? 0 : 1);
            if (!in || !use_S) n1 = 0;
            const int e16 = (ei >= ek) ? ei * 16 + ek : ek * 16 + ei;   // the S tiles hold both triangles: [max][min]
            cls[rr][j] = n1 == 2 ? (2 | ((ea * 8 + (ek == 13 ? 7 : ek)) << 2)) : n1;
            sv[rr][j] = aa.S[n1 == 1 ? (unsigned)(c0 * 256 + e16) : 0u];
            const unsigned src = rowoff + kc;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < aa.nparts) gv[rr][j][u] = (aa.parts + (size_t)u * aa.stride)[src];   // (wave-uniform count and base)
            pv[rr][j] = 0.0;
            if (aa.plus) pv[rr][j] = aa.plus[((ic >> 4) >= (int)(kc >> 4)) ? src : kc * (unsigned)aa.NAP + (unsigned)ic];
        }
    }
    if (has_shared) __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave + 4 * rr;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = l + 64 * j;
            double sS = cls[rr][j] == 1 ? sv[rr][j] : 0.0;
            if (has_shared && (cls[rr][j] & 3) == 2) {
                const int en = cls[rr][j] >> 2;
                sS = (sShared[0][en] + sShared[1][en]) + (sShared[2][en] + sShared[3][en]);
            }
            double g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = u < aa.nparts ? gv[rr][j][u] : 0.0;
            const bool in = 16 * bi + r < M && k < K;
            sA[r * LDA + k] = in ? sS - ((g[0] + g[1]) + (g[2] + g[3])) + pv[rr][j] : 0.0;
        }
    }
    __syncthreads();
    // ---- split-K product from the strip ---------------------------------------------------------------------------
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    {
        double a[GB];
#pragma unroll
        for (int q = 0; q < GB; ++q) {
            const int k = kbeg + 4 * q + kk;
            const bool kin = k < kend;
            a[q] = kin ? sA[cc * LDA + (kin ? k : 0)] : 0.0;
            bv[q] = (jb && kin) ? bv[q] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < GB; q += 2) {
            acc0 = mfma_f64(a[q], bv[q], acc0);
            acc1 = mfma_f64(a[q + 1], bv[q + 1], acc1);
        }
    }
    d4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc0[r] + acc1[r];
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sPart[wave - 1][r][l] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = ((acc[r] + sPart[0][r][l]) + sPart[1][r][l]) + sPart[2][r][l];
        const int io = 16 * bi + kk + 4 * r, jo = 16 * bj + cc;
        if (io < M && jo < Nc) C[(long)io * sCi + (long)jo * sCj] = v;
    }
}

// k_gemm_asmA behind a completion word: the Gram of the in-state features' rows (aa.plus) is formed on ANOTHER stream while k_front
// runs (orcvio_msckf_io_step_frame, capi_step.inc) and a one-word launch behind it stores the frame's sequence number; every workgroup
// polls the word (bounded: a wait that gives up flags the update, which is then run again in separate launches) once its other loads
// are in flight and reads that Gram past the caches; the shared body -- the same arithmetic as k_gemm_asmA, the same bits.
__global__ __launch_bounds__(256) void k_gemm_asmA_w(AsmArgs aa, const double* __restrict__ B, long sBk, long sBj, int M, int Nc, int K,
                                                     double* __restrict__ C, long sCi, long sCj, const unsigned* __restrict__ wait, unsigned expect,
                                                     int spin_limit, int* __restrict__ lost) {
    __shared__ double sU[ASM_LDS_DOUBLES];
    asm_gemm_tile<false, true>(aa, B, sBk, sBj, M, Nc, K, C, sCi, sCj, (int)blockIdx.x, true, (int)threadIdx.x, sU, sU + 16 * 193, sU + 16 * 193 + 768,
                               wait, expect, spin_limit, lost);
}

// ---------------------------------------------------------------------------------------
// k_front: the two independent front ends of the update in ONE launch -- workgroup 0 factors the prior
// (potrf_reg_body, depends on P only), every other workgroup runs two feature tracks (feature_body, one
// four-wavefront team each).  Replaces a cross-stream fork / join around k_potrf_reg (8 + 10 us of dependency
// latency in the replayed graph) by plain co-residency: 1 + ceil(F/2) workgroups of 512 threads, all resident
// at once for F <= 510.  The second team takes its roles two wavefronts on, so that the two chain-like
// wavefronts of a workgroup (QR + gate) sit on different SIMDs.
// ---------------------------------------------------------------------------------------
struct FrontPotrfArgs {
    const double* X; int ldx; int n; double tol_rel; double* R; int ldr; double* Dinv; int* info;
    int skip;   // the prior's factor is resident (orcvio_msckf_cov_commit): workgroup 0 has nothing to do
    int rev;    // factor the reversed matrix (potrf_reg_body): the last 15 columns of the factor are zero in the active rows
    int nfar;   // > 0: the look-ahead form (potrf_lookahead.hpp, depth 3) -- workgroups 1 .. nfar are the far workgroups of the block rows 4 ..
    int* la_flag; int* la_rdy; int la_spin;   // its step counter, its block-row words (both zero between launches), its bound on the waits
};
// The compression (both Grams, then the assembly of A) can run in the same launch: the feature workgroups meet at a
// device-wide counter (they are all resident: the launch has at most as many workgroups as the device has CUs, one
// per CU by its LDS size), take (tile, chunk) / clone work items, meet again and assemble.  The factorisation in
// workgroup 0 takes no part; it outlasts all of this, so the compression is free.
struct FrontGramArgs {
    int enabled;
    int chunks, rows_per_chunk;   // T3 row chunks of this launch (<= 8 x 80 rows each: one batch of loads per wavefront)
    double* Gpart; double* S; const int* clone_rows;
    int* counter;                 // zero between launches (the last workgroup through resets it)
    int* lost;                    // incremented if a wait gives up (reported as an error by the host)
    double* A_dst; int cb0;
    const double* plus;           // optional Gram added to A (EKF-SLAM rows), lower tiles valid
    int spin_limit;               // polls of the flag line before a waiting workgroup gives up (a few tens of ms)
    unsigned* started;            // optional: workgroup 0 stores started_val here with its first instruction -- "everything enqueued in front
    unsigned started_val;         //   of this launch is complete", read by a wait launch on another stream (the frame call's in-state rows)
};
// U = [A; b^T] L_a inside the launch (g.enabled == 3): behind the Grams and a second device-wide barrier, every team takes one tile of U
// with k_gemm_asmA's body -- same arithmetic, same bits -- as soon as the prior's factorisation has published the block row of R the
// tile's columns of L_a come from (its step counter q.la_flag; nothing to wait for when the factor is resident).  The launch then ends
// ~3 us later than the factorisation does, and the k_gemm_asmA launch behind it (8 us + a launch gap) is gone.  The factorising
// workgroup puts its hand-off words back to zero only when every feature workgroup has said it is through (u.done).
struct FrontUArgs {
    AsmArgs aa;
    const double* B; long sBk, sBj;
    int M, Nc, K;
    double* C; long sCi, sCj;
    int* done;   // zero between launches
};
__device__ __forceinline__ void front_grid_barrier(int* counter, int target, int* lost, int spin_limit, unsigned long long* dbg = nullptr) {
    // Everything that crosses this barrier is written with write-through (sc1) stores and read with sc1 loads
    // (st_pub / ld_sel<true>), so no cache write-back or invalidate is needed: every wavefront waits for its own stores
    // to be acknowledged, one thread bumps the counter and polls it.  (A release/acquire fence pair per workgroup --
    // L2 write-back and invalidate -- made each barrier 11-14 us.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (dbg) dbg[0] = wall_clock64();
        // the arrivals are read-modify-writes on one line; the waiting is done on ANOTHER line with plain loads, so the
        // polls of 200 workgroups do not queue up with the arrivals: the last one in raises the flag (monotonic phase)
        int* flag = counter + 32;
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == target - 1) __hip_atomic_store(flag, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {   // bounded: a workgroup that never arrives (its CU is held by somebody else's kernel) must not hang the GPU;
                 // the update is then flagged and the host re-runs it on the forked path (orcvio_msckf_download)
            int it = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && it < spin_limit) { __builtin_amdgcn_s_sleep(4); ++it; }
            if (it >= spin_limit && lost) atomicAdd(lost, 1);
        }
    }
    __syncthreads();
}
template <int NPASS, int NSLOT, bool WITH_U = false>   // WITH_U: the instantiation that can form U behind the Grams (g.enabled == 3, FrontUArgs)
__global__ __launch_bounds__(512) void k_front(FeatArgs p, FrontPotrfArgs q, int team_doubles, FrontGramArgs g, FrontUArgs u) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (blockIdx.x == 0) {
        if (g.started && threadIdx.x == 0) __hip_atomic_store(g.started, g.started_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q.skip) return;
        if (g.enabled && threadIdx.x == 0) (reinterpret_cast<unsigned long long*>(g.counter) + 8)[0] = wall_clock64();
        if (WITH_U && q.nfar > 0 && g.enabled == 3) {   // (the feature workgroups read the step counter until their tiles of U are out)
            potrf_la_chain_wg<3, false, true, true>(smem, LaIn{q.X, q.ldx, q.n, q.rev}, q.tol_rel, q.R, q.ldr, q.Dinv, q.info, q.la_flag, q.la_rdy, g.lost, q.la_spin, nullptr);
            __syncthreads();
            if ((threadIdx.x >> 6) == 4) {   // (the wavefront that stored the step counter)
                const int l = threadIdx.x & 63, nfb = (int)gridDim.x - 1 - q.nfar;
                int it = 0;
                while (__hip_atomic_load(u.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nfb && it < g.spin_limit) { __builtin_amdgcn_s_sleep(8); ++it; }
                if (it >= g.spin_limit && l == 0) atomicAdd(g.lost, 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (l < 16) __hip_atomic_store(q.la_rdy + l, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (l == 16) __hip_atomic_store(q.la_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (l == 17) __hip_atomic_store(u.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        else if (q.nfar > 0) potrf_la_chain_wg<3, false, true>(smem, LaIn{q.X, q.ldx, q.n, q.rev}, q.tol_rel, q.R, q.ldr, q.Dinv, q.info, q.la_flag, q.la_rdy, g.lost, q.la_spin, nullptr);
        else potrf_reg_body<NSLOT, false>(smem, q.X, q.ldx, q.n, q.tol_rel, q.R, q.ldr, q.Dinv, q.info, 0, 0, nullptr, nullptr, 1, 0, q.rev);
        if (g.enabled && threadIdx.x == 0) (reinterpret_cast<unsigned long long*>(g.counter) + 8)[6] = wall_clock64();
        return;
    }
    if ((int)blockIdx.x <= q.nfar) {   // the far workgroups of the prior's factorisation (block rows 4 .. nb-2)
        if (q.skip) return;
        potrf_la_far_wg<3, false>(3 + (int)blockIdx.x, LaIn{q.X, q.ldx, q.n, q.rev}, q.R, q.ldr, q.la_flag, q.la_rdy, g.lost, q.la_spin, nullptr);
        return;
    }
    // (Persistent teams -- a loop over track pairs so that any track count stays in this launch -- were tried in round 2: with
    // feature_body inside a loop the kernel spills 141 VGPRs and 2000 tracks took 273 us against 125 + 24 + 19 us for the forked
    // k_feature / k_gram_pair / k_assemble_A, whose two workgroups per CU also keep more tracks in flight.  Not kept: beyond
    // 2 (CUs - 1) tracks the update takes the forked form.)
    const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int j = 2 * ((int)blockIdx.x - 1 - q.nfar) + team;
    const int local = threadIdx.x & 255;
    if (j < p.F) {
        feature_body<NPASS, true>(p, j, team ? ((local + 128) & 255) : local, smem + (size_t)team * team_doubles);
    } else {   // (an odd track count: the last workgroup has one team)
        __syncthreads(); __syncthreads(); __syncthreads();
    }
    if (!g.enabled) return;
    const int nfb = (int)gridDim.x - 1 - q.nfar, me = (int)blockIdx.x - 1 - q.nfar;
    unsigned long long* stamp = reinterpret_cast<unsigned long long*>(g.counter) + 8;   // diagnostic (bytes 64..): 100 MHz clock
#define FRONT_STAMP(i) do { if (me == 0 && threadIdx.x == 0) stamp[i] = wall_clock64(); } while (0)
    FRONT_STAMP(1);
    front_grid_barrier(g.counter, nfb, g.lost, g.spin_limit);
    FRONT_STAMP(2);
    // ---- Grams: (lower tile, chunk) items of T3, then one item per clone of the sparse rows -----------------
    const int nb = p.NAP >> 4, ntiles = nb * (nb + 1) / 2;
    const int nitems = ntiles * g.chunks + p.N;
    const int m3 = 3 * p.F;
    for (int it = me; it < nitems; it += nfb) {
        if (it < ntiles * g.chunks) {
            const int ch = it / ntiles, tl = it - ch * ntiles;
            int bi, bj;
            tile_from_linear(tl, bi, bj);
            const int r0 = ch * g.rows_per_chunk;
            int r1 = r0 + g.rows_per_chunk;
            if (r1 > m3) r1 = m3;
            gramw_body<8, 20, true, true>(smem, p.T3, p.NAP, r0, r1 > r0 ? r1 : r0, bi, bj, g.Gpart + (size_t)ch * p.NAP * p.NAP, p.NAP);
        } else {
            const int c = it - ntiles * g.chunks;
            gramw_body<8, 20, true>(smem, p.Xobs, 16, g.clone_rows[c], g.clone_rows[c + 1], 0, 0, g.S + (size_t)c * 256, 16);
        }
        __syncthreads();   // the reduction buffer is reused by the next item
    }
    FRONT_STAMP(3);
    int phases = 2;
    if (WITH_U && g.enabled == 3) {   // ---- U = [A; b^T] L_a: one tile per team (FrontUArgs) ------------------------------------------
        front_grid_barrier(g.counter, 2 * nfb, g.lost, g.spin_limit, me == 0 ? stamp + 7 : nullptr);
        FRONT_STAMP(4);
        const int ntj = (u.Nc + 15) >> 4, ntu = ((u.M + 15) >> 4) * ntj;
        double* sU = smem + (size_t)team * ASM_LDS_DOUBLES;
        for (int t0 = 2 * me; t0 < ntu; t0 += 2 * nfb) {
            const int tile = t0 + team;
            const bool active = tile < ntu;
            if (!q.skip && active && local == 0) {   // block row (tile's column block) of R published?  (bounded; one lane polls, the barrier below)
                const int need = tile % ntj + 1;
                int it = 0;
                while (__hip_atomic_load(q.la_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && it < q.la_spin) { __builtin_amdgcn_s_sleep(4); ++it; }
                if (it >= q.la_spin) { atomicAdd(g.lost, 1); }
            }
            __syncthreads();
            asm_gemm_tile<true>(u.aa, u.B, u.sBk, u.sBj, u.M, u.Nc, u.K, u.C, u.sCi, u.sCj, active ? tile : 0, active, local, sU, sU + 16 * 193, sU + 16 * 193 + 768);
            __syncthreads();   // (the strip and the partial tiles are reused by the next item)
        }
        FRONT_STAMP(5);
        if (!q.skip && q.nfar > 0) {   // this workgroup reads the factorisation's words no more
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(u.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        phases = 3;
    }
    if (g.enabled == 1) {   // (enabled == 2: the consumer, k_gemm_asmA, assembles A on the fly)
        front_grid_barrier(g.counter, 2 * nfb, g.lost, g.spin_limit, me == 0 ? stamp + 7 : nullptr);
        FRONT_STAMP(4);
        // ---- A = scatter(S) - sum of the partial Grams --------------------------------------------------------
        for (int idx = me * 512 + (int)threadIdx.x; idx < p.NAP * p.NAP; idx += nfb * 512)
            assemble_entry<true>(idx, g.S, p.N, g.cb0, p.NA, p.NAP, g.Gpart, g.chunks, (size_t)p.NAP * p.NAP, g.A_dst, 0, g.plus);
        FRONT_STAMP(5);
        phases = 3;
    }
#undef FRONT_STAMP
    // the last workgroup through puts the counter (and the flag) back to zero for the next launch
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(g.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == phases * nfb - 1) {
            __hip_atomic_store(g.counter + 32, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// generic strided product C(i,j) = alpha * sum_k A(i,k) B(k,j) [+ diag_add on i == j] [+ Cin(i,j)]; tiles with
// bi <= bj only when upper_only.  One workgroup per 16x16 tile, split-K over its 4 wavefronts (these products
// are latency-bound: a K = 187 chain of dependent MFMAs and its operand loads, cut to a quarter), partial tiles
// summed through LDS in wave order (deterministic).
__global__ __launch_bounds__(256) void k_gemm(const double* __restrict__ A, long sAi, long sAk, const double* __restrict__ B,
                                              long sBk, long sBj, int M, int N, int K, double alpha, double diag_add,
                                              int upper_only, double* __restrict__ C, long sCi, long sCj,
                                              const double* __restrict__ Cin = nullptr, int* __restrict__ clear = nullptr,
                                              const unsigned* __restrict__ wait = nullptr, unsigned expect = 0u, int* __restrict__ lost = nullptr,
                                              int spin_limit = 0, int* __restrict__ clear16 = nullptr, int* __restrict__ clear1 = nullptr) {
    // clear: step counter of the k_potrf_solve launch that follows in the stream (reset here, one kernel ahead); clear16: the sixteen
    // block-row words of k_potrf_solve_la; clear1: the counter its finish workgroups wait for (LaFin; a cache line of its own)
    if (clear1 && blockIdx.x == 0 && threadIdx.x == 16) __hip_atomic_store(clear1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (clear && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(clear, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (clear16 && blockIdx.x == 0 && threadIdx.x < 16) __hip_atomic_store(clear16 + threadIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // wait: completion counter of a launch on ANOTHER stream whose output this product reads (the frame call: A' of the objects'
    // compression, k_gemm_objA).  A stream-level join (hipStreamWaitEvent) costs ~10 us of dispatch even when the event fired long ago;
    // one relaxed poll per workgroup costs nothing when it did.  Bounded: a count that never comes sets *lost (the update is void).
    if (wait) {
        if (threadIdx.x == 0) {
            int spins = 0;
            while ((int)(__hip_atomic_load(wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expect) < 0) {
                if (++spins > spin_limit) { if (lost) atomicExch(lost, 1); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    __shared__ double sPart[3][4][64];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int ntj = (N + 15) >> 4;
    const int tile = blockIdx.x;
    const int bi = tile / ntj, bj = tile - bi * ntj;
    if (upper_only && bi > bj) return;   // whole workgroup
    const int KS = ((K + 15) >> 4) << 2;   // k-slice per wavefront, a multiple of the MFMA depth
    const int k0 = wave * KS;
    const int Kw = (K - k0 < KS) ? (K - k0) : KS;   // may be <= 0: empty slice
    d4 acc = tile_product(A + (long)k0 * sAk, sAi, sAk, B + (long)k0 * sBk, sBk, sBj, M, N, Kw, 16 * bi, 16 * bj, l);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sPart[wave - 1][r][l] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
    const int kk = l >> 4, cc = l & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = ((acc[r] + sPart[0][r][l]) + sPart[1][r][l]) + sPart[2][r][l];
        const int i = 16 * bi + kk + 4 * r, j = 16 * bj + cc;
        if (i < M && j < N)
            C[(long)i * sCi + (long)j * sCj] = alpha * v + ((i == j) ? diag_add : 0.0) + (Cin ? Cin[(long)i * sCi + (long)j * sCj] : 0.0);
    }
}

// Z = L^-1 B with L given through strides (L(i,j) = L[i*sLi + j*sLj]; an upper factor R stored
// row-major is L = R^T: sLi = 1, sLj = ldr).  B(i,c) = B1[i*ldb1 + c] for c < nc1, the single
// extra column c == nc1 is bx[i*sbx].  Right-looking: after Z_kb is final, every later block
// row is updated with independent MFMAs (loads of L tiles are issued a block step ahead).
template <int NBLK>
__global__ __launch_bounds__(64) void k_trsm_rl(const double* __restrict__ L, long sLi, long sLj, const double* __restrict__ Dinv,
                                                int NA, const double* __restrict__ B1, long sB1i, long sB1c, int nc1,
                                                const double* __restrict__ bx, long sbx, double* __restrict__ Z, int ldz) {
    const int l = threadIdx.x;
    const int j0 = blockIdx.x * 16;
    const int kk = l >> 4, cc = l & 15;
    const int nblk = (NA + 15) >> 4;
    const int col = j0 + cc;
    const int ncols = nc1 + (bx ? 1 : 0);
    d4 acc[NBLK];
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * kb + kk + 4 * r;
            double v = 0.0;
            if (kb < nblk && i < NA) {
                if (col < nc1) v = B1[(long)i * sB1i + (long)col * sB1c];
                else if (col == nc1 && bx) v = bx[(long)i * sbx];
            }
            acc[kb][r] = v;
        }
    }
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb) {
        if (kb < nblk) {
            // Z_kb = Dinv_kb * acc_kb
            d4 x = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 4; ++s) x = mfma_f64(Dinv[(size_t)kb * 256 + cc * 16 + kk + 4 * s], acc[kb][s], x);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * kb + kk + 4 * r;
                if (col < ncols && i < NA) Z[(size_t)i * ldz + col] = x[r];
            }
            // acc_ib -= L[ib][kb] * Z_kb for every later block row
#pragma unroll
            for (int ib = 0; ib < NBLK; ++ib) {
                if (ib > kb && ib < nblk) {
                    const int ia = 16 * ib + cc;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const double a = (ia < NA) ? -L[(long)ia * sLi + (long)(16 * kb + kk + 4 * s) * sLj] : 0.0;
                        acc[ib] = mfma_f64(a, x[s], acc[ib]);
                    }
                }
            }
        }
    }
}

// k_trsm_lds: Z = L^-1 B for L = R^T with R an upper factor stored row-major (ldr), nn <= 224.
// One workgroup = 4 wavefronts x 16 right-hand-side columns; the 16-column panel of L needed by
// block step kb (rows 16kb..16kb+15 of R) is staged through LDS once per workgroup, double
// buffered, and prefetched one step ahead; solved tiles stay in registers (accumulator layout =
// B-operand layout), later block rows are updated right-looking with independent MFMAs.
//   B(i,c) = B1[i*sB1i + c*sB1c] for c < nc1; the optional extra column c == nc1 is bx[i*sbx].
#define TRSM_LDP 240
__global__ __launch_bounds__(256) void k_trsm_lds(const double* __restrict__ R, int ldr, const double* __restrict__ Dinv, int nn,
                                                  const double* __restrict__ B1, long sB1i, long sB1c, int nc1,
                                                  const double* __restrict__ bx, long sbx, double* __restrict__ Z, int ldz,
                                                  size_t strideR = 0, size_t strideD = 0, size_t strideB = 0, size_t strideZ = 0,
                                                  size_t strideBx = 0) {
    __shared__ __attribute__((aligned(16))) double sL[2][16][TRSM_LDP];
    // batched use: blockIdx.y selects the system
    if (bx) bx += (size_t)blockIdx.y * strideBx;
    R += (size_t)blockIdx.y * strideR;
    Dinv += (size_t)blockIdx.y * strideD;
    B1 += (size_t)blockIdx.y * strideB;
    Z += (size_t)blockIdx.y * strideZ;
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    const int kk = l >> 4, cc = l & 15;
    const int nblk = (nn + 15) >> 4;
    const int ncols = nc1 + (bx ? 1 : 0);
    const int col = (blockIdx.x * 4 + wave) * 16 + cc;
    const bool wave_live = (int)(blockIdx.x * 4 + wave) * 16 < ncols;
    d4 acc[14];
#pragma unroll
    for (int kb = 0; kb < 14; ++kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * kb + kk + 4 * r;
            double v = 0.0;
            if (kb < nblk && i < nn) {
                if (col < nc1) v = B1[(long)i * sB1i + (long)col * sB1c];
                else if (col == nc1 && bx) v = bx[(long)i * sbx];
            }
            acc[kb][r] = v;
        }
    }
    // panel loader: element pair e = tid + 256 q  <->  (row c = e / npair, columns 2*(e % npair) + c0 ..)
    double2 pf[7];
    auto panel_load = [&](int kb) {
        const int c0 = 16 * (kb + 1);           // first column needed (even)
        const int ncol = nn - c0;
        const int npair = (ncol + 1) >> 1;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int e = tid + 256 * q;
            double2 v = {0.0, 0.0};
            if (ncol > 0 && e < 16 * npair) {
                const int c = e / npair, pi = e - c * npair;
                const double* src = R + (size_t)(16 * kb + c) * ldr + c0 + 2 * pi;
                v = *reinterpret_cast<const double2*>(src);   // ldr and c0 even -> 16-byte aligned
            }
            pf[q] = v;
        }
    };
    auto panel_store = [&](int kb, int buf) {
        const int c0 = 16 * (kb + 1);
        const int ncol = nn - c0;
        const int npair = (ncol + 1) >> 1;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int e = tid + 256 * q;
            if (ncol > 0 && e < 16 * npair) {
                const int c = e / npair, pi = e - c * npair;
                *reinterpret_cast<double2*>(&sL[buf][c][c0 + 2 * pi]) = pf[q];
            }
        }
    };
    panel_load(0);
    panel_store(0, 0);
    double di[4], din[4];   // inv(L11) operands of the current / next block step
#pragma unroll
    for (int s = 0; s < 4; ++s) { di[s] = Dinv[cc * 16 + kk + 4 * s]; din[s] = 0.0; }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < 14; ++kb) {
        if (kb < nblk) {
            if (kb + 1 < nblk) {
                panel_load(kb + 1);
#pragma unroll
                for (int s = 0; s < 4; ++s) din[s] = Dinv[(size_t)(kb + 1) * 256 + cc * 16 + kk + 4 * s];
            }
            if (wave_live) {
                d4 x = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 4; ++s) x = mfma_f64(di[s], acc[kb][s], x);
                acc[kb] = x;   // final; stored after the loop (no global stores ahead of a barrier)
#pragma unroll
                for (int ib = 0; ib < 14; ++ib) {
                    if (ib > kb && ib < nblk) {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            acc[ib] = mfma_f64((16 * ib + cc < nn) ? -sL[kb & 1][kk + 4 * s][16 * ib + cc] : 0.0, x[s], acc[ib]);
                    }
                }
            }
            if (kb + 1 < nblk) panel_store(kb + 1, (kb + 1) & 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) di[s] = din[s];
            __syncthreads();
        }
    }
    if (wave_live && col < ncols) {
#pragma unroll
        for (int kb = 0; kb < 14; ++kb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * kb + kk + 4 * r;
                if (kb < nblk && i < nn) Z[(size_t)i * ldz + col] = acc[kb][r];
            }
        }
    }
}

// k_potrf_solve: Cholesky M = R^T R and the triangular solve Z = R^-T B in ONE launch.  Workgroup 0 is the
// factorisation (potrf_reg_body, publishing); every wavefront of workgroups 1.. owns 16 right-hand-side columns
// (all block rows in registers) and runs the right-looking solve one block step behind the factorisation: it polls
// the step counter, reads row block k of R and inv(L11) with L1-bypassing loads, finishes Z_k and updates the later
// block rows with independent MFMAs.  The solve is off the critical chain: the launch ends a few microseconds after
// the factorisation instead of a whole k_trsm later.  SOLVE_WPB live wavefronts per solver workgroup (one per SIMD).
#define SOLVE_WPB 4
#define SOLVE_CH 4
// One solver workgroup of k_potrf_solve / k_potrf_solve_la (block index `sb` among the solver workgroups): see k_potrf_solve.
// PUB: Z is consumed inside the launch (LaFin): stored past the caches.
template <bool PUB>
__device__ __forceinline__ void potrf_solver_body(const int sb, int n, const double* __restrict__ R, int ldr, const double* __restrict__ Dinv,
                                                int* __restrict__ flag, int* __restrict__ lost_flag,
                                                const double* __restrict__ B1, long sB1i, long sB1c, int nc1,
                                                const double* __restrict__ bx, long sbx, double* __restrict__ Z, int ldz,
                                                int tail, double tail_scale) {
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int kk = l >> 4, cc = l & 15;
    const int nblk = (n + 15) >> 4;
    const int ncols = nc1 + (bx ? 1 : 0);
    const int cb = sb * SOLVE_WPB + wave;
    if (wave >= SOLVE_WPB || cb * 16 >= ncols) return;
    const int col = cb * 16 + cc;
    for (int e = l; e < 16 * tail; e += 64) {   // (independent of the factorisation: out of the way first)
        const int i = n + (e >> 4), c = cb * 16 + (e & 15);
        if (c < ncols) st_pub<PUB>(&Z[(size_t)i * ldz + c], (c < nc1) ? tail_scale * B1[(long)i * sB1i + (long)c * sB1c] : 0.0);
    }
    d4 acc[14];
#pragma unroll
    for (int kb = 0; kb < 14; ++kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * kb + kk + 4 * r;
            double v = 0.0;
            if (kb < nblk && i < n) {
                if (col < nc1) v = B1[(long)i * sB1i + (long)col * sB1c];
                else if (col == nc1 && bx) v = bx[(long)i * sbx];
            }
            acc[kb][r] = v;
        }
    }
    int seen = 0;
    bool lost = false;
    const int l4 = cc * 16 + kk;                    // inv(L11) operand: Linv[m = cc][k = kk + 4s]
    const size_t lane_off = (size_t)kk * ldr + cc;  // lane part of every R address
#pragma unroll
    for (int kb = 0; kb < 14; ++kb) {
        if (kb < nblk) {
            // wait until row block kb of R and inv(L11)_kb are in memory (bounded spin: a lost producer must not hang the GPU)
            if (seen <= kb && !lost) {
#pragma unroll 1
                for (int it = 0; it < (1 << 24); ++it) {
                    // (scalar load past the scalar cache: the flag is one word for the whole wavefront, and the scalar path answers
                    //  sooner than a vector load that has to come back to 64 lanes)
                    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(seen) : "s"(flag) : "memory");
                    if (seen > kb) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                lost = seen <= kb;   // never on a healthy device; the solve then runs through on stale bytes and is flagged
            }
            asm volatile("" ::: "memory");
            double di[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) di[s] = ld_pub(Dinv + (size_t)kb * 256 + l4 + 4 * s);
            // A operands of the later block rows, SOLVE_CH tiles at a time (register budget: 256 per wave), the next
            // chunk in flight while the current one is consumed: element (16kb + kk + 4s, 16ib + cc) of R
            const double* Rk = R + (size_t)(16 * kb) * ldr;   // wave-uniform
            double a[2][SOLVE_CH][4];
#pragma unroll
            for (int q = 0; q < SOLVE_CH; ++q) {
                const int ib = kb + 1 + q;
                if (ib < 14 && ib < nblk) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) a[0][q][s] = ld_pub(Rk + (size_t)(4 * s) * ldr + 16 * ib + lane_off);
                }
            }
            d4 x = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 4; ++s) x = mfma_f64(di[s], acc[kb][s], x);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * kb + kk + 4 * r;
                if (col < ncols && i < n) st_pub<PUB>(&Z[(size_t)i * ldz + col], x[r]);
            }
#pragma unroll
            for (int c = 0; c * SOLVE_CH < 13 - kb; ++c) {
#pragma unroll
                for (int q = 0; q < SOLVE_CH; ++q) {   // prefetch chunk c + 1
                    const int ib = kb + 1 + (c + 1) * SOLVE_CH + q;
                    if (ib < 14 && ib < nblk) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) a[(c + 1) & 1][q][s] = ld_pub(Rk + (size_t)(4 * s) * ldr + 16 * ib + lane_off);
                    }
                }
#pragma unroll
                for (int q = 0; q < SOLVE_CH; ++q) {
                    const int ib = kb + 1 + c * SOLVE_CH + q;
                    if (ib < 14 && ib < nblk) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) acc[ib] = mfma_f64(-a[c & 1][q][s], x[s], acc[ib]);
                    }
                }
            }
        }
    }
    if (lost && l == 0) atomicAdd(lost_flag, 1);   // reported as an error by the host
}
__device__ __forceinline__ void potrf_solver_wg(const int sb, int n, const double* __restrict__ R, int ldr, const double* __restrict__ Dinv,
                                                int* __restrict__ flag, int* __restrict__ lost_flag,
                                                const double* __restrict__ B1, long sB1i, long sB1c, int nc1,
                                                const double* __restrict__ bx, long sbx, double* __restrict__ Z, int ldz,
                                                int tail, double tail_scale) {
    potrf_solver_body<false>(sb, n, R, ldr, Dinv, flag, lost_flag, B1, sB1i, sB1c, nc1, bx, sbx, Z, ldz, tail, tail_scale);
}

template <int NSLOT>
__global__ __launch_bounds__(512) void k_potrf_solve(const double* __restrict__ X, int ldx, int n, double tol_rel,
                                                     double* __restrict__ R, int ldr, double* __restrict__ Dinv,
                                                     int* __restrict__ info, int* __restrict__ flag, int* __restrict__ lost_flag,
                                                     const double* __restrict__ B1, long sB1i, long sB1c, int nc1,
                                                     const double* __restrict__ bx, long sbx, double* __restrict__ Z, int ldz,
                                                     int tail = 0, double tail_scale = 0.0) {
    // tail: M = diag(X (n x n), s2 I_tail) -- the right-hand sides B1 have `tail` more rows behind the n that take part in the
    // factorisation (the last columns of the prior's factor, zero in the active rows: potrf_reg_body, rev); their rows of Z are
    // B1 / sigma (tail_scale), the extra column's are zero.
    if (blockIdx.x == 0) {
        __shared__ __attribute__((aligned(16))) double sPotrf[POTRF_LDS_DOUBLES];
        potrf_reg_body<NSLOT, true>(sPotrf, X, ldx, n, tol_rel, R, ldr, Dinv, info, 0, 0, flag, nullptr, 1, 0);   // lower tiles: kept zero by the handle
        return;
    }
    potrf_solver_wg(blockIdx.x - 1, n, R, ldr, Dinv, flag, lost_flag, B1, sB1i, sB1c, nc1, bx, sbx, Z, ldz, tail, tail_scale);
}

// LaEnd (optional): the factorising workgroup does not END before *word has reached `expect` (a cumulative counter another stream's launch
// stores), and copies four status words -- the frame call's chained object solve: the launch behind this one (k_finish_sqrt) reads what the
// feature half's commit wrote, and its start-of-kernel acquire must come after that commit.  Long satisfied when this kernel ends: no wait
// in practice, and no launch of its own for it.
// `mark` (optional, independent of the rest): a word the factorising workgroup stores as its first instruction -- "every launch ahead of
// this one in the stream is complete", for a launch on another stream that polls it (the chained object solve waits for M this way).
struct LaEnd { const unsigned* word = nullptr; unsigned expect = 0u; const int* info_src = nullptr; int* info_dst = nullptr;
               unsigned* mark = nullptr; unsigned mark_val = 0u; };
// Joint chi-square gate of an object update (gatingTest on the stacked, projected rows, src/orcvio.cpp:2172-2182), decided inside
// k_finish_sqrt by every workgroup for itself (same arithmetic, same order: same decision): gamma = (|r'|^2 - |z|^2) / s2 with
// |r'|^2 = *rr (corner of the compressed block) and z = Z[:, n].  Workgroup 0 writes gamma / accept for the consumers after this
// launch (k_fac_commit, the outputs arena).
struct ObjGate {
    const double* rr = nullptr;   // nullptr: no gate (feature updates)
    double thr = 0.0;
    double *gamma = nullptr, *gamma_out = nullptr;
    int *accept = nullptr, *accept_out = nullptr;
    const int* fail = nullptr;   // the two pivot counters of chol(M) (dropped, non-positive); either != 0: M was not positive definite
                                 // (non-finite or absurdly scaled input) and the update is not applied (P+ = P, dx = 0), for feature
                                 // and object updates alike
};
// LaFin (kernel template argument FIN): P+ = s2 Zn^T Zn and dx = Zn^T z inside this launch instead of a k_finish_sqrt behind it -- a
// feature update's (no chi-square gate).  `nfin` finish workgroups behind the far ones, two tiles each, wait for the counter `done` (zero at
// launch: rdy[15], cleared with the rest): every solver workgroup adds 1 when its stores of Z have landed (stored past the caches), the
// chain workgroup 1, and 1 << 16 more if M was not positive definite (then P+ = P, dx = 0, as k_finish_sqrt decides from the status
// words).  The arithmetic is k_finish_sqrt's tile by tile -- the same split of K over four wavefronts, the same order: the same bits.
struct LaFin {
    int first = 0;              // block index of the first finish workgroup
    unsigned* done = nullptr;
    const int* step = nullptr;  // the factorisation's step counter (flag), last_step: its value when the last but one block step is out
    int last_step = 0;
    int acq = 0;
    int nstate = 0, kdim = 0;   // Z = [Zn | z] is kdim x (nstate + 1)
    double s2 = 0.0;
    double* P_out = nullptr;
    double* dx = nullptr;
    const double* P = nullptr;
    int keep_tail = 0;
    ObjGate gate;               // object updates: the joint chi-square gate, as k_finish_sqrt decides it (gate.fail is not read: the counter carries it)
    const unsigned* wait_word = nullptr;   // the chained frame: P is what ANOTHER stream's launch commits; its word has reached wait_expect
    unsigned wait_expect = 0u;             // when that is done (read past the caches then)
};
__device__ __forceinline__ void la_finish_wg(const int fb, const LaFin f, const double* __restrict__ Z, const int ldz, const int expect,
                                             int* __restrict__ lost_flag, const int spin) {
    __shared__ double sPart[2][3][4][64];
    __shared__ unsigned sSeen;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int half = wave >> 2, w = wave & 3, n = f.nstate;
    const int nbt = (n + 1 + 15) >> 4, t = 2 * fb + half;
    const bool live = t < nbt * (nbt + 1) / 2;
    int bi = 0, bj = 0;
    if (live) tile_from_linear(t, bi, bj);
    if (wave == 0) {   // (one wave polls, the others load behind the barrier it joins: MI355X_MICROARCH.md, valid forms, first row)
        // While the factorisation is under way: a look at its step counter every few microseconds (the solver wavefronts poll that word
        // all the time; the counter this workgroup waits for cannot rise before the last block step).
#pragma unroll 1
        for (int it = 0; it < spin && __hip_atomic_load(f.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < f.last_step; ++it)
            __builtin_amdgcn_s_sleep(64);
        unsigned v = 0u;
#pragma unroll 1
        for (int it = 0; it <= spin; ++it) {
            v = __hip_atomic_load(f.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(v & 0xffffu) >= expect) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (f.wait_word) {
#pragma unroll 1
            for (int it = 0; it <= spin && (int)(__hip_atomic_load(f.wait_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - f.wait_expect) < 0; ++it) {
                if (it == spin) v = 0u;   // reported below
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (l == 0) sSeen = v;
        if (f.acq) {   // (experiment: one agent acquire, then plain loads that the XCD's L2 serves to all its finish workgroups)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    const unsigned seen = sSeen;
    if ((int)(seen & 0xffffu) < expect) {   // never on a healthy device: reported, the host runs the update again in separate launches
        if (threadIdx.x == 0) atomicAdd(lost_flag, 1);
        return;
    }
    const bool failed = (seen >> 16) != 0u;
    __shared__ double sZZ[2][4];
    if (f.gate.rr) {   // |z|^2, by every half workgroup for itself (k_finish_sqrt's partition and order: the same gamma)
        double zz = 0.0;
        for (int i = threadIdx.x & 255; i < f.kdim; i += 256) { const double v = ld_pub(Z + (size_t)i * ldz + n); zz += v * v; }
        zz = wave_sum(zz);
        if (l == 0) sZZ[half][w] = zz;
    }
    d4 acc = {0, 0, 0, 0};
    if (live) {
        const int KS = ((f.kdim + 15) >> 4) << 2;
        const int k0 = w * KS;
        const int Kw = (f.kdim - k0 < KS) ? (f.kdim - k0) : KS;
        // (sixteen k-steps in flight at once: the 50-odd rows of a wavefront in ONE round trip; the k-steps alternate between the two
        //  accumulators whatever the batch size, so the sums are k_finish_sqrt's)
        if (f.acq) acc = tile_product<16, false>(Z + (size_t)k0 * ldz, 1, ldz, Z + (size_t)k0 * ldz, ldz, 1, n + 1, n + 1, Kw, 16 * bi, 16 * bj, l);
        else acc = tile_product<16, true>(Z + (size_t)k0 * ldz, 1, ldz, Z + (size_t)k0 * ldz, ldz, 1, n + 1, n + 1, Kw, 16 * bi, 16 * bj, l);
        if (w > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sPart[half][w - 1][r][l] = acc[r];
        }
    }
    __syncthreads();
    if (!live || w > 0) return;
    const int kk = l >> 4, cc = l & 15;
    bool app = !failed;
    if (f.gate.rr) {
        const double g = (*f.gate.rr - ((sZZ[half][0] + sZZ[half][1]) + (sZZ[half][2] + sZZ[half][3]))) / f.s2;
        app = !failed && (g == g && g < f.gate.thr);
        if (t == 0 && l == 0) { *f.gate.gamma = g; *f.gate.accept = app ? 1 : 0; *f.gate.gamma_out = g; *f.gate.accept_out = app ? 1 : 0; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = ((acc[r] + sPart[half][0][r][l]) + sPart[half][1][r][l]) + sPart[half][2][r][l];
        const int i = 16 * bi + kk + 4 * r, jj = 16 * bj + cc;
        if (i < n && jj < n && jj <= i) {
            const bool prior = !app || (i >= n - f.keep_tail && jj >= n - f.keep_tail);
            double pv = f.s2 * v;
            if (prior) pv = f.wait_word ? 0.5 * (ld_pub(f.P + (size_t)i * n + jj) + ld_pub(f.P + (size_t)jj * n + i))
                                        : 0.5 * (f.P[(size_t)i * n + jj] + f.P[(size_t)jj * n + i]);
            f.P_out[(size_t)i * n + jj] = pv;
            f.P_out[(size_t)jj * n + i] = pv;
        } else if (i == n && jj < n) {
            f.dx[jj] = app ? v : 0.0;
        }
    }
}
// k_potrf_solve_la: k_potrf_solve with the trailing update of the factorisation spread over far workgroups (potrf_lookahead.hpp).
// grid: [0] the chain workgroup, [1 .. nsolve] the solver workgroups (potrf_solver_wg), [nsolve+1 ..] the far workgroups of the
// block rows LA+1 .. nb-2, [fin.first ..] the finish workgroups (FIN).  Dynamic LDS: la_lds_doubles<LA>() doubles.  rdy[16]: zero at
// launch (k_gemm clears it with the step counter, one kernel ahead).
template <int LA, bool ST, bool FIN = false>
__global__ __launch_bounds__(512) void k_potrf_solve_la(const double* __restrict__ X, int ldx, int n,
                                                        double* __restrict__ R, int ldr, double* __restrict__ Dinv,
                                                        int* __restrict__ info, int* __restrict__ flag, int* __restrict__ rdy,
                                                        int* __restrict__ lost_flag, int nsolve,
                                                        const double* __restrict__ B1, long sB1i, long sB1c, int nc1,
                                                        const double* __restrict__ bx, long sbx, double* __restrict__ Z, int ldz,
                                                        int tail, double tail_scale, int spin, unsigned long long* __restrict__ stamps,
                                                        LaEnd end = LaEnd(), LaFin fin = LaFin()) {
    extern __shared__ __attribute__((aligned(16))) double sLaLds[];
    if (FIN && (int)blockIdx.x >= fin.first) {
        la_finish_wg((int)blockIdx.x - fin.first, fin, Z, ldz, nsolve + 1, lost_flag, spin);
        return;
    }
    if (blockIdx.x == 0) {
        if (end.mark && threadIdx.x == 0) __hip_atomic_store(end.mark, end.mark_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        potrf_la_chain_wg<LA, ST, false>(sLaLds, LaIn{X, ldx, n, 0}, 0.0, R, ldr, Dinv, info, flag, rdy, lost_flag, spin, stamps);
        if (FIN) {   // (the two pivot counters: sCnt[30..31] of the chain workgroup's LDS)
            __syncthreads();
            if (threadIdx.x == 0) {
                const int* c = reinterpret_cast<const int*>(sLaLds + 1360);
                __hip_atomic_fetch_add(fin.done, 1u + ((c[30] != 0 || c[31] != 0) ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (end.word && threadIdx.x == 0) {   // (the frame call's chained object solve: see LaEnd)
            int spins = 0;
            while ((int)(__hip_atomic_load(end.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - end.expect) < 0) {
                if (++spins > spin) { atomicExch(lost_flag, 1); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            const int idx[4] = {0, 1, 4, 5};
            for (int q = 0; q < 4; ++q) end.info_dst[idx[q]] = __hip_atomic_load(end.info_src + idx[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if ((int)blockIdx.x <= nsolve) {
        potrf_solver_body<FIN>(blockIdx.x - 1, n, R, ldr, Dinv, flag, lost_flag, B1, sB1i, sB1c, nc1, bx, sbx, Z, ldz, tail, tail_scale);
        if (FIN) {   // every storing wave waits for its stores, the workgroup's barrier, one lane signals
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(fin.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    potrf_la_far_wg<LA, ST>(LA + 1 + ((int)blockIdx.x - nsolve - 1), LaIn{X, ldx, n, 0}, R, ldr, flag, rdy, lost_flag, spin, stamps);
}


// P_out = s2 * Zn^T Zn (symmetric), dx = Zn^T z, with Z = [Zn | z] (kdim x (n+1), ldz).  One workgroup per lower
// tile, split-K over its 4 wavefronts (as k_gemm).
__global__ __launch_bounds__(256) void k_finish_sqrt(const double* __restrict__ Z, int ldz, int n, int kdim, double s2,
                                                     double* __restrict__ P_out, double* __restrict__ dx,
                                                     ObjGate gate = ObjGate(), const double* __restrict__ P = nullptr, int keep_tail = 0) {
    // keep_tail: the trailing keep_tail x keep_tail block of P+ is the prior's (Schmidt nuisance states, src/orcvio.cpp:1740-1751)
    __shared__ double sPart[3][4][64];
    __shared__ double sZZ[4];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    int bi, bj;
    tile_from_linear(blockIdx.x, bi, bj);
    if (gate.rr) {   // (loads in flight together with the tile's)
        double zz = 0.0;
        for (int i = threadIdx.x; i < kdim; i += 256) { const double v = Z[(size_t)i * ldz + n]; zz += v * v; }
        zz = wave_sum(zz);
        if (l == 0) sZZ[wave] = zz;
    }
    const int KS = ((kdim + 15) >> 4) << 2;   // Z is kdim x (n + 1): kdim = dimension of M (= n unless a resident factor is used)
    const int k0 = wave * KS;
    const int Kw = (kdim - k0 < KS) ? (kdim - k0) : KS;
    d4 acc = tile_product(Z + (size_t)k0 * ldz, 1, ldz, Z + (size_t)k0 * ldz, ldz, 1, n + 1, n + 1, Kw, 16 * bi, 16 * bj, l);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sPart[wave - 1][r][l] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
    const int kk = l >> 4, cc = l & 15;
    bool app = true;   // gated object update: leave P and x alone if rejected
    const bool failed = gate.fail && (gate.fail[0] != 0 || gate.fail[1] != 0);
    if (failed) app = false;
    if (gate.rr) {
        const double g = (*gate.rr - ((sZZ[0] + sZZ[1]) + (sZZ[2] + sZZ[3]))) / s2;
        app = !failed && (g == g && g < gate.thr);   // NaN anywhere in the rows makes g NaN -> rejected
        if (blockIdx.x == 0 && l == 0) { *gate.gamma = g; *gate.accept = app ? 1 : 0; *gate.gamma_out = g; *gate.accept_out = app ? 1 : 0; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = ((acc[r] + sPart[0][r][l]) + sPart[1][r][l]) + sPart[2][r][l];
        const int i = 16 * bi + kk + 4 * r, jj = 16 * bj + cc;
        if (i < n && jj < n && jj <= i) {
            const bool prior = !app || (i >= n - keep_tail && jj >= n - keep_tail);
            const double pv = prior ? 0.5 * (P[(size_t)i * n + jj] + P[(size_t)jj * n + i]) : s2 * v;
            P_out[(size_t)i * n + jj] = pv;
            P_out[(size_t)jj * n + i] = pv;
        } else if (i == n && jj < n) {
            dx[jj] = app ? v : 0.0;
        }
    }
}

}  // namespace orcvio_amd

#include "object_kernels.hpp"   // k_obj_front, k_obj_border_solve_assemble (+ obj_refine_body), k_obj_border_qr, k_obj_solve_assemble, k_obj_refine

namespace orcvio_amd {

// Test hook (tests/test_gpu_robustness.py): hold `gridDim.x` compute units for `ticks` of the 100 MHz wall clock (every
// workgroup takes a whole CU's LDS), so that a co-resident launch next to it cannot get all of its workgroups resident.
__global__ __launch_bounds__(64) void k_debug_occupy(unsigned long long ticks, int* sink) {
    extern __shared__ double sOcc[];
    const unsigned long long t0 = wall_clock64();
    sOcc[threadIdx.x] = (double)threadIdx.x;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (sOcc[threadIdx.x] < 0.0) *sink = 1;
}

}  // namespace orcvio_amd

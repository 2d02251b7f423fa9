// object_kernels.hpp -- the object update's compression (reference OrcVIO::removeLostObjects, src/orcvio.cpp:2154-2193, and the
// projection nullspace_project_inplace_svd, include/orcvio/utils/math_utils.hpp:287-312): cross products per (object, clone), the
// structured Householder QR of Hf, Y = Q1^T X by substitution or through an explicit basis, the sum of the clone tiles.
// Included by msckf_kernels.hpp (it uses that file's MFMA / DPP helpers and Gram bodies); not a translation unit of its own.
#pragma once

namespace orcvio_amd {

// ---------------------------------------------------------------------------------------
// Object blocks (reference OrcVIO::removeLostObjects, src/orcvio.cpp:2154-2193)
// ---------------------------------------------------------------------------------------
// Every object row touches one clone (6 non-zeros of Hx) and the object's own state columns Hf.  With X = [Hx | r]
// (scattered into the window's columns) the left-nullspace projection of math_utils.hpp:287-312 is the Schur complement
//     A' = B - C^T F^-1 C,   B = X^T X,  C = Hf^T X,  F = Hf^T Hf
// (any orthonormal basis of the left nullspace of Hf gives the same A'), formed with the Cholesky factor of F:
// Y = L_F^-1 C, A' = B - Y^T Y.  None of B, C, F needs the rows scattered to window width: the rows of an object are
// grouped by clone, and per (object, clone) group only 6 x (6 + no + 1) numbers are new --
//     hx^T hx (6 x 6), hx^T r (6)            -> the clone's 7 x 7 tile of B      (Sg, 8 x 8 per (object, clone))
//     hx^T Hf (6 x no)                       -> the clone's 6 columns of C       (Cd, dense no x NAP per object)
// while F, Hf^T r and r^T r come from ONE Gram of the compact matrix [Hf | r] per object (Gff).  The stack is read
// once (rows x (6 + no + 1) doubles); nothing of window width is written except C itself.  (Round 1 wrote the rows
// scattered to NAP + NOP = 240 columns -- 28.6 MB -- and took a 240 x 240 Gram of that per object -- 94 MB fetched.)
//
// Compact row storage: Hx6 [rows][6];  HfR [rows][ldf], ldf = 16 ceil((no_max + 1)/16): columns [0, no) Hf of the
// object (zero up to no_max), column no_max = the residual, zero behind it.
// k_obj_cross: one wavefront per (object, clone) group {first, last+1 into ridx, clone, object}; ridx lists the rows of
// the update grouped by (object, clone).  MFMA with A = hx^T (6 of 16 rows live), B = [Hf | r] tile by tile and hx.
struct ObjGroup { int r0, r1, clone, obj; };
__device__ __forceinline__ void obj_cross_body(const ObjGroup* __restrict__ groups, int g, int l, const int* __restrict__ ridx,
                                               const double* __restrict__ Hx6, const double* __restrict__ HfR, int ldf,
                                               int no_max, int cb0, int NAP, int NOP, int N,
                                               double* __restrict__ Cd, double* __restrict__ Sg, double* __restrict__ Hr) {
    // A operand = [hx (6) ; r] ^T: seven live rows.  Rows 0..5 give hx^T [Hf | r | hx] as before; row 6 gives r^T Hf (this
    // group's part of Hf^T r), r^T hx and r^T r -- the compact Gram of [Hf | r] is not needed for them (arrow route)
    const ObjGroup grp = groups[g];
    const int m = l & 15, kq = l >> 4;
    const int nt = ldf >> 4;   // <= 8 (object state <= 112 columns)
    d4 acc[8], ahh = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = d4{0, 0, 0, 0};
    for (int k0 = grp.r0; k0 < grp.r1; k0 += 8) {   // two k-steps (8 rows) per trip: their loads are in flight together
        double a[2], b[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = k0 + 4 * u + kq;
            const bool in = k < grp.r1;
            const int row = ridx[in ? k : grp.r1 - 1];
            const double av = m < 6 ? Hx6[(size_t)row * 6 + m] : HfR[(size_t)row * ldf + no_max];   // (m == 6: the residual)
            a[u] = (in && m < 7) ? av : 0.0;
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < nt) { const double bv = HfR[(size_t)row * ldf + 16 * t + m]; b[u][t] = in ? bv : 0.0; }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            ahh = mfma_f64(a[u], a[u], ahh);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < nt) acc[t] = mfma_f64(a[u], b[u][t], acc[t]);
        }
    }
    // D[mm][nn], mm = kq + 4 r (the hx component 0..5, 6 = the residual), nn = m
    double* Co = Cd + (size_t)grp.obj * NOP * NAP;
    double* So = Sg + ((size_t)grp.obj * N + grp.clone) * 64;
    double* Ho = Hr + ((size_t)grp.obj * N + grp.clone) * NOP;
    const int colb = cb0 + 6 * grp.clone;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int mm = kq + 4 * r;
        if (mm < 7) {
            if (m < 7) So[mm * 8 + m] = ahh[r];   // the 7 x 7 tile [hx | r]^T [hx | r] of this (object, clone)
            if (mm < 6) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (t < nt) {
                        const int i = 16 * t + m;   // column of [Hf | r]
                        if (i < no_max) Co[(size_t)i * NAP + colb + mm] = acc[t][r];
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (t < nt) {
                        const int i = 16 * t + m;
                        if (i < NOP) Ho[i] = i < no_max ? acc[t][r] : 0.0;   // this group's part of Hf^T r
                    }
                }
            }
        }
    }
}
// dst (NAP x NAP, full symmetric) = sum over objects of B_o: clone tiles from Sg, |r|^2 from Gff[no_max][no_max]
// (objects summed in index order: deterministic)
__device__ __forceinline__ void obj_assemble_B_body(int idx, const double* __restrict__ Sg, int nobj, int N, const double* __restrict__ rr,
                                                    size_t rr_stride, int cb0, int NA, int NAP, double* __restrict__ dst,
                                                    const double* corner_value = nullptr) {
    // corner_value: sum of |r|^2 over all objects, computed by the caller (the launch that also holds the border QR, where the
    // per-object sums do not exist yet)
    // rr[o * rr_stride] = |r|^2 of object o (arrow route: summed over the clone tiles by k_obj_border_qr; Gram route: the corner of
    // the compact Gram)
    if (idx >= NAP * NAP) return;
    const int i = idx / NAP, j = idx - i * NAP;
    const int ci = (i >= cb0 && i < cb0 + 6 * N) ? (i - cb0) / 6 : -1, cj = (j >= cb0 && j < cb0 + 6 * N) ? (j - cb0) / 6 : -1;
    const int ei = ci >= 0 ? i - cb0 - 6 * ci : (i == NA ? 6 : -1), ej = cj >= 0 ? j - cb0 - 6 * cj : (j == NA ? 6 : -1);
    double s = 0.0;
    const bool corner = ei == 6 && ej == 6;
    const bool tile = !corner && ei >= 0 && ej >= 0 && (ci == cj || ci < 0 || cj < 0);
    if (corner && corner_value) s = *corner_value;
    else if (corner || tile) {   // (everything else of the block is structurally zero: no loads)
        const double* base = corner ? rr : Sg + (size_t)(ci >= 0 ? ci : cj) * 64 + ei * 8 + ej;
        const size_t st = corner ? rr_stride : (size_t)N * 64;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int o = 0;
        for (; o + 4 <= nobj; o += 4) {   // four loads in flight; objects summed in a fixed order (deterministic)
            const double v0 = base[(size_t)o * st], v1 = base[(size_t)(o + 1) * st], v2 = base[(size_t)(o + 2) * st], v3 = base[(size_t)(o + 3) * st];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; o < nobj; ++o) s0 += base[(size_t)o * st];
        s = (s0 + s1) + (s2 + s3);
    }
    dst[idx] = s;
}

// ---- the triangular factor of Hf by STRUCTURED HOUSEHOLDER QR (no Gram of Hf) ------------------------------------------
// Y = R^-T C needs the triangular factor R of Hf = Q R.  chol(Hf^T Hf) squares the condition number, and Hf of a real
// object is badly conditioned by construction: a keypoint row is invariant under "move the object, move every keypoint
// back" (a gauge of the keypoint rows that only the four bbox rows per frame break), so cond(Hf) ~ 1e8 on the reference's
// own data (src/tests/data/one_car: 2.5e8) and the Gram loses that direction altogether -- delta_x off by 10 %.  The
// reference takes the left nullspace from a full-U JacobiSVD (math_utils.hpp:287-312), accurate to cond * eps.
// Hf is an ARROW matrix (include/orcvio/obj/ObjectLM.h:117-123): columns [pose 6 | shape 3 | keypoint k: 3 each], a
// keypoint row touches the border (pose, shape) and ITS keypoint's block only, a bbox row the border only.  So:
//   phase B  (k_obj_kp_qr) one wavefront per keypoint: Householder QR (three reflectors, LAPACK dgeqr2 convention) of its m x 3 block
//            (m <= 128 rows, two per lane), applied to the nine border columns of the same rows: R_kk (3 x 3), R_kb
//            (3 x 9), and the rows' border part after elimination (back into LDS)
//   phase C  (k_obj_border_qr) one workgroup per object: Householder QR of the eliminated border (all rows x 9): R_b (9 x 9)
// R = [[blockdiag R_kk, R_kb], [0, R_b]] up to the column order.  k_obj_arrow_solve then forms Y = R^-T C by forward
// substitution, one thread per column of C.  A pivot that is zero to rounding (a keypoint seen in one frame only, an exactly
// dependent column) is dropped (its row of Y is zero) and counted.
struct ObjArrow { int row0, rows, K, kp_off; };   // rows [row0, row0 + rows) of the update; K keypoint blocks; kp_off: first
                                                  // entry of this object in kp_range
// Rout per object (stride arrow_stride(Kmax)): [K][3 x 3 R_kk | 3 x 9 R_kb] , then 9 x 9 R_b, then the pivot tolerance
__host__ __device__ inline int arrow_stride(int Kmax) { return 36 * Kmax + 81 + 3; }
// Two launches (a first version did both phases in one 1024-thread workgroup per object: 755 spilled VGPRs, 209 us):
//   k_obj_kp_qr      grid (ceil(Kmax / 4), objects) x 256 threads: one wavefront per (object, keypoint), everything in
//                    registers; the border part of its rows is rewritten IN PLACE in HfR (columns 0..8; the three pivot rows are
//                    zeroed) -- k_obj_cross / k_obj_gram_ff have read HfR before
//   k_obj_border_qr  one 256-thread workgroup per object: the nine reflectors of the eliminated border, rows in registers
__device__ __forceinline__ void obj_kp_qr_body(const ObjArrow ob, int obj, int k, int lane, const int2* __restrict__ kp_range,
                                               const int* __restrict__ kp_rows, const double* __restrict__ HfR, int ldf, int Kmax,
                                               double* __restrict__ Rout, double* __restrict__ Bred) {
    const int2 rg = kp_range[ob.kp_off + k];
    if (k == ob.K) {   // the border-only rows (bbox rows): their border goes to Bred as it is
        for (int p = rg.x + lane; p < rg.y; p += 64) {
            const int r = kp_rows[p];
#pragma unroll
            for (int c = 0; c < 9; ++c) Bred[(size_t)r * 9 + c] = HfR[(size_t)r * ldf + c];
        }
        return;
    }
    double* Ro = Rout + (size_t)obj * arrow_stride(Kmax);
    const int m = rg.y - rg.x;   // <= 128 (host-checked)
    int rl[2];
    bool in[2];
    double a[2][3], b[2][9];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int pos = lane + 64 * u;
        in[u] = pos < m;
        rl[u] = in[u] ? kp_rows[rg.x + pos] : ob.row0;
        const double* row = HfR + (size_t)rl[u] * ldf;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const double v = row[9 + 3 * k + c]; a[u][c] = in[u] ? v : 0.0; }
#pragma unroll
        for (int c = 0; c < 9; ++c) { const double v = row[c]; b[u][c] = in[u] ? v : 0.0; }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // rows below the pivot (list positions > j): g_c = sum a_j * col_c over them, for col = a_j (norm^2), a_c>j, b_0..8
        const bool below0 = lane > j;   // slot 0: position = lane; slot 1: position = lane + 64 > j always
        const double x0 = (in[0] && below0) ? a[0][j] : 0.0, x1 = in[1] ? a[1][j] : 0.0;
        double g[12];
        g[0] = x0 * x0 + x1 * x1;
#pragma unroll
        for (int c = 1; c < 3; ++c) g[c] = (c > j) ? x0 * a[0][c] + x1 * a[1][c] : 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) g[3 + c] = x0 * b[0][c] + x1 * b[1][c];
#pragma unroll
        for (int q = 0; q < 12; ++q) g[q] = wave_sum_dpp(g[q]);   // (independent: the twelve reductions interleave)
        const double alpha = bcast_lane(a[0][j], j);   // pivot entry (0 if the list is shorter than j + 1: a is 0 there)
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (g[0] > 0.0) {   // dlarfg: beta = -sign(alpha) ||x||, tau = (beta - alpha) / beta, v = x / (alpha - beta), v_pivot = 1
            const double nrm = sqrt(alpha * alpha + g[0]);
            beta = alpha >= 0.0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        const double v0 = x0 * scale, v1 = x1 * scale;   // the reflector below the pivot (0 outside)
        const bool piv = lane == j;                       // slot 0 of lane j is the pivot row
#pragma unroll
        for (int c = 1; c < 3; ++c) {
            if (c > j) {
                const double w = tau * (bcast_lane(a[0][c], j) + scale * g[c]);
                a[0][c] -= piv ? w : w * v0;
                a[1][c] -= w * v1;
            }
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const double w = tau * (bcast_lane(b[0][c], j) + scale * g[3 + c]);
            b[0][c] -= piv ? w : w * v0;
            b[1][c] -= w * v1;
        }
        if (piv) a[0][j] = beta;
        if (lane > j) a[0][j] = 0.0;
        a[1][j] = 0.0;
    }
    // R_kk rows 0..2 and R_kb rows 0..2 live in slot 0 of lanes 0..2; the other rows go to the reduced border
    if (lane < 3) {
        double* o = Ro + 36 * k + 12 * lane;   // row `lane`: [3 of R_kk | 9 of R_kb]
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (c >= lane && lane < m) ? a[0][c] : 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) o[3 + c] = lane < m ? b[0][c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (in[u]) {
            const bool consumed = (u == 0 && lane < 3);
            double* row = Bred + (size_t)rl[u] * 9;
#pragma unroll
            for (int c = 0; c < 9; ++c) row[c] = consumed ? 0.0 : b[u][c];
        }
    }
}
template <int RPT>   // rows per thread: RPT * 256 >= the rows of the largest object (2, 4 or 8)
__device__ __forceinline__ void obj_border_qr_body(const int obj, const ObjArrow* __restrict__ objs, const double* __restrict__ Bred, int Kmax,
                                                   double* __restrict__ Rout, const double* __restrict__ Hr, const double* __restrict__ Sg,
                                                   int N, int NOP, double* __restrict__ Hfr, double* __restrict__ sArrow = nullptr) {
    // sArrow (the fused launch): LDS copy of the object's arrow factor for the substitution that follows in the same workgroup -- the
    // keypoint blocks (written by the launch before) are fetched HERE, with the rows, and R_b / the tolerance go to it directly: the
    // substitution starts without a round trip through memory for the factor this workgroup has just finished
    if (sArrow) {
        const double* Rg = Rout + (size_t)obj * arrow_stride(Kmax);
        for (int i = threadIdx.x; i < 36 * Kmax; i += 256) sArrow[i] = Rg[i];
    }
    // prologue (independent of the reflectors): Hf^T r and |r|^2 of the object = the sum of its clone groups' parts.  Value i = thread mod 128,
    // clones of one parity per half of the workgroup, sixteen loads in flight per thread: one memory round trip, hidden under the row loads
    // below; the two partial sums are combined in a fixed order at the end of the kernel (deterministic).  Hfr[o] = [Hf^T r | r^T r]
    __shared__ double sSum[2][128];   // (NOP + 1 <= 113 values)
    {
        const int i = threadIdx.x & 127, part = threadIdx.x >> 7;
        double acc = 0.0;
        if (i <= NOP) {
            const double* src = i < NOP ? Hr + (size_t)obj * N * NOP + i : Sg + (size_t)obj * N * 64 + 54;
            const size_t st = i < NOP ? (size_t)NOP : 64;
            for (int c0 = part; c0 < N; c0 += 32) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int c = c0 + 2 * u; const double t = src[(size_t)(c < N ? c : c0) * st]; v[u] = c < N ? t : 0.0; }
                acc += (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
                       (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
            }
        }
        sSum[part][i] = acc;
    }
    __shared__ double sPiv[16];
    __shared__ double sPart[4 * 9];
    const ObjArrow ob = objs[obj];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    double* Ro = Rout + (size_t)obj * arrow_stride(Kmax);
    // thread t holds rows t, t + 256, ... (up to 2048 rows per object, host-checked)
    double x[RPT][9];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = tid + 256 * q;
        const double* row = Bred + (size_t)(ob.row0 + (r < ob.rows ? r : 0)) * 9;
#pragma unroll
        for (int c = 0; c < 9; ++c) { const double v = row[c]; x[q][c] = r < ob.rows ? v : 0.0; }
    }
    double pmax = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        // partial dot products of column j (rows below the pivot) with the columns j..8: ONE block reduction per reflector
        double g[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            double sacc = 0.0;
            if (c >= j) {
#pragma unroll
                for (int q = 0; q < RPT; ++q) sacc += ((tid + 256 * q > j) ? x[q][j] : 0.0) * x[q][c];   // (rows beyond ob.rows hold zeros)
            }
            g[c] = sacc;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c)
            if (c >= j) g[c] = wave_sum_dpp(g[c]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 9; ++c)
                if (c >= j) sPart[wave * 9 + c] = g[c];
        }
        if (tid == j) {   // pivot row j is held by thread j (q = 0): publish it
#pragma unroll
            for (int c = 0; c < 9; ++c) sPiv[c] = x[0][c];
        }
        __syncthreads();
        double pr[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            pr[c] = sPiv[c];
            if (c >= j) g[c] = (sPart[c] + sPart[9 + c]) + (sPart[18 + c] + sPart[27 + c]);   // fixed order: deterministic
        }
        __syncthreads();   // sPart / sPiv are rewritten by the next reflector
        const double alpha = (j < ob.rows) ? pr[j] : 0.0;
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (g[j] > 0.0) {
            const double nrm = sqrt(alpha * alpha + g[j]);
            beta = alpha >= 0.0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = tid + 256 * q;
            const double vq = r > j ? x[q][j] * scale : 0.0;   // the reflector's entry of this row (the pivot row: 1)
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                if (c > j) {
                    const double w = tau * (pr[c] + scale * g[c]);
                    x[q][c] -= (r == j) ? w : w * vq;
                }
            }
            if (r == j) x[q][j] = beta; else if (r > j) x[q][j] = 0.0;
        }
        pmax = fmax(pmax, fabs(beta));
    }
    // R_b row j = row j of the reduced border (thread j, q = 0)
    if (tid < 9) {
        double* o = Ro + 36 * Kmax + 9 * tid;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const double v = (c >= tid && tid < ob.rows) ? x[0][c] : 0.0;
            o[c] = v;
            if (sArrow) sArrow[36 * Kmax + 9 * tid + c] = v;
        }
    }
    // pivot tolerance: a pivot below 1e-11 of the largest one is rounding noise of an exactly dependent column.  (Every pivot of a
    // triangular factor is >= the smallest singular value: with cond(Hf) ~ 2.5e8 on the reference's own data real pivots stay
    // above 4e-9 of the largest; the noise pivot of a dependent border column reached 1e-13 in the randomised soak,
    // scripts/gpu_soak_objects.py seed 1060 -- a tolerance of 1e-13 kept it and the update came back as NaN.)
    if (wave == 0) {   // (the 3 K pivots of the keypoint blocks: one memory round trip for the wavefront, not one per keypoint for a thread)
        double mx = pmax;
        const double* piv = sArrow ? sArrow : Ro;   // (the LDS copy is complete: the reflectors' barriers stand in between)
        for (int e = lane; e < 3 * ob.K; e += 64) mx = fmax(mx, fabs(piv[36 * (e / 3) + 13 * (e % 3)]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
        if (lane == 0) {
            Ro[36 * Kmax + 81] = 1e-11 * mx;
            if (sArrow) sArrow[36 * Kmax + 81] = 1e-11 * mx;
        }
    }
    if (tid <= NOP) Hfr[(size_t)obj * (NOP + 1) + tid] = sSum[0][tid] + sSum[1][tid];   // (behind the barriers of the reflectors)
}
template <int RPT>
__global__ __launch_bounds__(256) void k_obj_border_qr(const ObjArrow* __restrict__ objs, const double* __restrict__ Bred, int Kmax,
                                                       double* __restrict__ Rout, const double* __restrict__ Hr, const double* __restrict__ Sg,
                                                       int N, int NOP, double* __restrict__ Hfr) {
    obj_border_qr_body<RPT>((int)blockIdx.x, objs, Bred, Kmax, Rout, Hr, Sg, N, NOP, Hfr);
}
// Y_o = R^-T C_o for the arrow factor: one thread per column of C (window columns 0..NA-1 from Cd, column NA = Hf^T r
// from the compact Gram's residual row).  Rows of Y in the order of Hf's columns; dropped pivots give zero rows.
__device__ __forceinline__ void obj_arrow_solve_body(double* __restrict__ sR, int o, int col, const ObjArrow* __restrict__ objs,
                                                     const double* __restrict__ Rin, int Kmax,
                                                     const double* __restrict__ Cd, int NOP, int NAP, int NA,
                                                     const double* __restrict__ Hfr,
                                                     double* __restrict__ Y, int* __restrict__ info, bool preloaded = false) {
    // (the factor goes to LDS once per workgroup; every thread has the C entries of a batch of keypoints in flight before it
    // starts substituting: a handful of memory round trips, not one per keypoint)
    const ObjArrow ob = objs[o];
    const double* Ro = Rin + (size_t)o * arrow_stride(Kmax);
    if (!preloaded) {   // (preloaded: the fused launch filled sR in obj_border_qr_body, a barrier stands in between)
        for (int i = threadIdx.x; i < arrow_stride(Kmax); i += 256) sR[i] = Ro[i];
        __syncthreads();
    }
    if (col > NA) {   // padding columns of the block: zero (A' = B - Y^T Y is formed over all NAP columns)
        if (col < NAP)
            for (int i = 0; i < NOP; ++i) Y[(size_t)o * NOP * NAP + (size_t)i * NAP + col] = 0.0;
        return;
    }
    const double tol = sR[36 * Kmax + 81];
    const double* Co = Cd + (size_t)o * NOP * NAP;
    const double* hr = Hfr + (size_t)o * (NOP + 1);   // Hf^T r of the object (k_obj_border_qr's prologue)
    double* Yo = Y + (size_t)o * NOP * NAP;
    const bool rcol = col == NA;
    const double* src = rcol ? hr : Co + col;
    const size_t st = rcol ? 1 : (size_t)NAP;
    double cb[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cb[i] = src[(size_t)i * st];
    int dropped = 0;
    constexpr int KB = 12;   // keypoints per batch of loads
    for (int k0 = 0; k0 < ob.K; k0 += KB) {
        double c[KB][3];
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            const int k = k0 + q < ob.K ? k0 + q : ob.K - 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) c[q][j] = src[(size_t)(9 + 3 * k + j) * st];
        }
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            const int k = k0 + q;
            if (k < ob.K) {
                const double* Rk = sR + 36 * k;   // rows j: [R_kk(j, 0..2) | R_kb(j, 0..8)]
                double y[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double t = c[q][j];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (i < j) t -= Rk[12 * i + j] * y[i];
                    const double p = Rk[12 * j + j];
                    const bool ok = fabs(p) > tol;
                    dropped += ok ? 0 : 1;
                    y[j] = ok ? t / p : 0.0;
                    Yo[(size_t)(9 + 3 * k + j) * NAP + col] = y[j];
                }
#pragma unroll
                for (int cc = 0; cc < 9; ++cc) cb[cc] -= Rk[3 + cc] * y[0] + Rk[12 + 3 + cc] * y[1] + Rk[24 + 3 + cc] * y[2];
            }
        }
    }
    const double* Rb = sR + 36 * Kmax;
    double yb[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double t = cb[j];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            if (i < j) t -= Rb[9 * i + j] * yb[i];
        const double p = Rb[9 * j + j];
        const bool ok = fabs(p) > tol;
        dropped += ok ? 0 : 1;
        yb[j] = ok ? t / p : 0.0;
        Yo[(size_t)j * NAP + col] = yb[j];
    }
    for (int i = 9 + 3 * ob.K; i < NOP; ++i) Yo[(size_t)i * NAP + col] = 0.0;   // rows behind this object's columns (A' sums Y^T Y over NOP rows)
    if (col == 0 && dropped > 0 && info) atomicAdd(info, dropped);
}

// ---- the object compression in three launches -------------------------------------------------------------------------
// Every stage of it is a small latency-bound kernel (~4 us floor each): the independent ones share a launch.
//   k_obj_front           512-thread workgroups with three roles by blockIdx.x: [0, nb_cross) the (object, clone) cross products, eight
//                         groups per workgroup; [.., + ntiles * nobj) one tile of an object's compact Gram [Hf | r]^T [Hf | r], its rows
//                         split over the eight wavefronts; [.., + kp_blocks * nobj) the keypoint blocks of the structured QR, eight per
//                         workgroup (arrow route only)
//   k_obj_border_qr       the nine border reflectors, one workgroup per object (arrow route only)
//   k_obj_solve_assemble  256-thread workgroups: [0, nb_solve) Y = R^-T C (arrow route), then sum_o B_o
__global__ __launch_bounds__(512) void k_obj_front(const ObjGroup* __restrict__ groups, int ngroups, const int* __restrict__ ridx,
                                                   const double* __restrict__ Hx6, const double* __restrict__ HfR, int ldf,
                                                   int no_max, int cb0, int NAP, int NOP, int N, double* __restrict__ Cd, double* __restrict__ Sg,
                                                   double* __restrict__ Hr, const int* __restrict__ row_ptr, double* __restrict__ Gff, int nobj, int gram_tiles,
                                                   const ObjArrow* __restrict__ objs, const int2* __restrict__ kp_range, const int* __restrict__ kp_rows,
                                                   int Kmax, double* __restrict__ Rout, double* __restrict__ Bred, int kp_blocks) {
    __shared__ __attribute__((aligned(16))) double sT[8 * 256];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nb_cross = (ngroups + 7) / 8;
    const int ntiles = gram_tiles;   // tiles of the compact Gram per object: Gram route only (0 in the arrow route)
    int b = blockIdx.x;
    if (b < nb_cross) {
        const int g = b * 8 + wave;
        if (g < ngroups) obj_cross_body(groups, g, lane, ridx, Hx6, HfR, ldf, no_max, cb0, NAP, NOP, N, Cd, Sg, Hr);
        return;
    }
    b -= nb_cross;
    if (b < ntiles * nobj) {
        const int o = b / ntiles, tl = b - o * ntiles;
        int bi, bj;
        tile_from_linear(tl, bi, bj);
        gramw_body<8, 20>(sT, HfR, ldf, row_ptr[o], row_ptr[o + 1], bi, bj, Gff + (size_t)o * ldf * ldf, ldf);
        return;
    }
    b -= ntiles * nobj;
    if (kp_blocks > 0) {
        const int o = b / kp_blocks, k = (b - o * kp_blocks) * 8 + wave;
        const ObjArrow ob = objs[o];
        if (k <= ob.K) obj_kp_qr_body(ob, o, k, lane, kp_range, kp_rows, HfR, ldf, Kmax, Rout, Bred);
    }
}
__global__ __launch_bounds__(256) void k_obj_solve_assemble(int nb_solve, int solve_xblocks, const ObjArrow* __restrict__ objs,
                                                            const double* __restrict__ Rin, int Kmax, const double* __restrict__ Cd, int NOP,
                                                            int NAP, int NA, const double* __restrict__ Hfr,
                                                            double* __restrict__ Y, int* __restrict__ info, const double* __restrict__ Sg,
                                                            int nobj, int N, int cb0, const double* __restrict__ rr, size_t rr_stride,
                                                            double* __restrict__ Bdst) {
    extern __shared__ double sR[];   // arrow_stride(Kmax) doubles (solve role)
    int b = blockIdx.x;
    if (b < nb_solve) {
        const int o = b / solve_xblocks, xb = b - o * solve_xblocks;
        obj_arrow_solve_body(sR, o, xb * 256 + (int)threadIdx.x, objs, Rin, Kmax, Cd, NOP, NAP, NA, Hfr, Y, info);
        return;
    }
    b -= nb_solve;
    obj_assemble_B_body(b * 256 + (int)threadIdx.x, Sg, nobj, N, rr, rr_stride, cb0, NA, NAP, Bdst);
}

// ---- ill-conditioned Hf: the projection through an explicit basis (obj_refine_body) ----------------------------------------
// Y = Q1^T X is what the Schur complement A' = B - Y^T Y needs (Q1 = an orthonormal basis of range(Hf), X = [Hx | r]).  The fast
// route takes it from the semi-normal equations, Y = R^-T (Hf^T X): the rounding of the products Hf^T X (eps |Hf| |X|) is divided by
// the small singular values of R, an error of cond(Hf) eps in Y -- 3e-8 on the reference's own one_car frames (cond(Hf) = 3e8: the
// gauge of the keypoint rows), which the update amplifies to 1.4e-6 in delta_x at a large prior (profiles/r3_conditioning.json) where
// the reference's full-U SVD (math_utils.hpp:287-312) keeps 4e-10.  The subtraction B - Y^T Y itself is NOT the problem (measured:
// with Y from explicit Householder reflectors the same subtraction reaches 4e-10).
// An object whose factor has |R|_F |R^-1|_F above OBJ_REFINE_COND therefore forms the basis EXPLICITLY, row by row,
//     q_i R = h_i   (a backward-stable triangular solve: Q~ R = Hf + E, |E| <= c eps |Q~| |R| -- the backward error of a Householder QR)
// and takes Y = Q~^T X directly from the rows (no product with Hf, nothing divided by a small pivot afterwards).  Q~ spans range(Hf)
// to that backward error but is orthonormal only to cond * eps: T = Q~^T Q~ = I + D.  The projector onto its range is Q~ T^-1 Q~^T,
// so A' = B - Y^T T^-1 Y = B - Y''^T Y'' with Y'' = (I - D/2) Y = 1.5 Y - 0.5 T Y up to D^2 (< 1e-10 even at the pivot tolerance).
// Both Q~ and T keep the arrow shape of Hf, so everything is 12 numbers per row.  One 256-thread workgroup per object, the rows of
// Q~ and of X staged in LDS in the order of the (object, clone) groups (objects of more rows than fit: in global scratch):
//   P  positions: lp = index of a row in the clone-grouped order; the keypoint lists as lists of positions
//   A  q_i for every row: [3 keypoint entries | 9 border entries]
//   B  T (per keypoint 3 x 3 and 3 x 9, border 9 x 9) and Q~^T r -- fixed summation orders (deterministic)
//   C  thread per window column: Y[:, col] from the rows of the column's clone, keypoints four at a time in registers, corrected and
//      written over the object's block of Y
#define OBJ_REFINE_COND 3e6
#define OBJ_REFINE_ROW_DOUBLES 21   // per row of the staging: q 12 (+1 pad), [hx | r] 7, four 16-bit indices (position, list entry, keypoint, clone)
#define OBJ_REFINE_SCRATCH 384      // doubles of static LDS the caller lends (partial tiles of T_bb)
struct RefineArgs {
    const int2* kp_range; const int* kp_rows; const ObjGroup* groups; int ngroups; const int* ridx;
    const double* Hx6; const double* HfR; int ldf, no_max;
    double* scratch;      // [rows_tot][OBJ_REFINE_ROW_DOUBLES] global staging for objects that do not fit the LDS staging
    int lds_rows;         // rows the LDS staging of the launch holds
    int mode;             // 1: objects above OBJ_REFINE_COND, 2: every object
    int* refined;         // counter of refined objects
    int cb0, N;
    unsigned long long* stamps;   // diagnostics (ORCVIO_REFINE_STAMPS): wall-clock stamps of object 0's phases
};
#define REFINE_STAMP(i) do { if (a.stamps && o == 0 && threadIdx.x == 0) a.stamps[i] = wall_clock64(); } while (0)
// cond_F^2 of the arrow factor in sR (sInv: 81 doubles of scratch); every thread of the 256 takes part (two barriers), the result is uniform.
// The diagonal of an unpivoted R says nothing (one_car: pivot ratio 5e-4 at cond 3e8), so the inverse is formed -- 12 numbers per
// keypoint with the arrow shape: R^-1 = [[R_kk^-1, -R_kk^-1 R_kb R_b^-1], [0, R_b^-1]].  Dropped pivots (zero columns of Q~) are left out.
__device__ __forceinline__ double obj_arrow_cond2(const double* __restrict__ sR, double* __restrict__ sInv, double* __restrict__ sOut, int K, int Kmax) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double tol = sR[36 * Kmax + 81];
    const double* Rb0 = sR + 36 * Kmax;
    if (tid < 9) {
        double x[9];
#pragma unroll
        for (int i = 8; i >= 0; --i) {
            double t = (i == tid) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < 9; ++m)
                if (m > i) t -= Rb0[9 * i + m] * x[m];
            const double pv = Rb0[9 * i + i];
            x[i] = fabs(pv) > tol ? t / pv : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) sInv[9 * i + tid] = x[i];
    }
    __syncthreads();
    if (wave == 0) {
        double nr = 0.0, ni = 0.0;
        if (lane < K) {
            const double* Rk = sR + 36 * lane;
            double inv[3][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {   // column j of R_kk^-1
#pragma unroll
                for (int i = 2; i >= 0; --i) {
                    double t = (i == j) ? 1.0 : 0.0;
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                        if (m > i) t -= Rk[12 * i + m] * inv[m][j];
                    const double pv = Rk[12 * i + i];
                    inv[i][j] = fabs(pv) > tol ? t / pv : 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double w[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    w[c] = (inv[i][0] * Rk[3 + c] + inv[i][1] * Rk[15 + c]) + inv[i][2] * Rk[27 + c];
                    nr += Rk[12 * i + 3 + c] * Rk[12 * i + 3 + c];
                }
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    double v = 0.0;
#pragma unroll
                    for (int m = 0; m < 9; ++m)
                        if (m <= c) v += w[m] * sInv[9 * m + c];
                    ni += v * v;
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) { ni += inv[i][j] * inv[i][j]; if (j >= i) nr += Rk[12 * i + j] * Rk[12 * i + j]; }
            }
        }
        for (int e = lane; e < 81; e += 64) { nr += Rb0[e] * Rb0[e]; ni += sInv[e] * sInv[e]; }   // (below the diagonal both hold zeros)
        nr = wave_sum_dpp(nr);
        ni = wave_sum_dpp(ni);
        if (lane == 0) sOut[0] = nr * ni;
    }
    __syncthreads();
    return sOut[0];
}
// Phases B and C of the explicit-basis projection on rows that stand in LDS / scratch (q: the rows of Q~, xr: [hx | r], the keypoint
// lists as positions in sList with ranges sRange[k] - p0, the clone groups sGrp, the (clone, keypoint) table sTab): shared by
// obj_refine_body (rows staged from device memory, four wavefronts) and k_obj_fused (rows evaluated in place, eight wavefronts).
// sPartR: 90 NW doubles of scratch.
template <int NW>
__device__ __forceinline__ void obj_refine_B(const int o, const int K, const int m, const int Kmax, double* __restrict__ sT, double* __restrict__ sYr,
                                             double* __restrict__ sPartR, const double* __restrict__ q, const double* __restrict__ xr,
                                             const unsigned short* __restrict__ sList, const int2* __restrict__ sRange, const int p0,
                                             const RefineArgs& a) {
    static_assert(NW == 4 || NW == 8, "four or eight wavefronts");
    constexpr int QS = 13;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- B: T = Q~^T Q~ and Q~^T r on the matrix cores (D = A B with A = operand rows of Q~ along the lanes' cc, four rows of the
    // object per instruction along kk: mfma_f64's layout, a = A[cc][kk], b = B[kk][cc], D[kk + 4r][cc]) -------------------------------
    // keypoint blocks: A = q_k (3 live rows), B = [q_k (3) | q_b (9) | r] (13 live columns) over the rows of keypoint k in list order;
    // keypoints dealt to the wavefronts round-robin, one accumulation chain each: T_kk, T_kb and q_k^T r of a keypoint in one tile
    {
        const int kk = lane >> 4, cc = lane & 15;
#pragma unroll 1
        for (int k = wave; k < K; k += NW) {
            const int e0 = sRange[k].x - p0, e1 = sRange[k].y - p0;
            d4 acc = {0, 0, 0, 0};
#pragma unroll 1
            for (int e = e0; e < e1; e += 64) {   // sixteen instructions' operands (64 rows) are read before the first of them issues
                int lpv[16];
                double av[16], bv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int ee = e + 4 * u + kk;
                    lpv[u] = sList[ee < e1 ? ee : e0];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const bool ok = e + 4 * u + kk < e1;
                    const double a0 = q[lpv[u] * QS + (cc < 3 ? cc : 0)];
                    const double b0 = cc < 12 ? q[lpv[u] * QS + cc] : xr[lpv[u] * 7 + 6];
                    av[u] = (ok && cc < 3) ? a0 : 0.0;
                    bv[u] = (ok && cc < 13) ? b0 : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (e + 4 * u < e1) acc = mfma_f64(av[u], bv[u], acc);   // (wave-uniform)
            }
            // D[i][j], i = kk + 4 r: rows 0..2 live (r == 0, kk < 3); j = cc: 0..2 T_kk, 3..11 T_kb, 12 q_k^T r
            if (kk < 3) {
                if (cc < 12) sT[36 * k + (cc < 3 ? 3 * kk + cc : 9 + 9 * kk + (cc - 3))] = acc[0];
                else if (cc == 12) sYr[9 + 3 * k + kk] = acc[0];   // (the reciprocals kept there were phase A's: dead behind its barrier)
            }
        }
    }
    REFINE_STAMP(4);
    // border: A = q_b (9 live rows), B = [q_b (9) | r] (10 live columns) over ALL rows, every wavefront a quarter of them (position
    // order), the four partial tiles summed through LDS in wave order
    {
        const int kk = lane >> 4, cc = lane & 15;
        const int steps = (m + 3) >> 2, per = (steps + NW - 1) / NW, s0 = wave * per, s1 = (s0 + per < steps) ? s0 + per : steps;
        d4 acc = {0, 0, 0, 0};
#pragma unroll 1
        for (int st = s0; st < s1; st += 16) {   // sixteen instructions' operands (64 rows) are read before the first of them issues
            double av[16], bv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int lp0 = 4 * (st + u) + kk;
                const bool ok = st + u < s1 && lp0 < m;
                const int lp = ok ? lp0 : 0;
                const double a0 = q[lp * QS + 3 + (cc < 9 ? cc : 0)];
                const double r0 = xr[lp * 7 + 6];
                av[u] = (ok && cc < 9) ? a0 : 0.0;
                bv[u] = ok ? (cc < 9 ? a0 : (cc == 9 ? r0 : 0.0)) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (st + u < s1) acc = mfma_f64(av[u], bv[u], acc);   // (wave-uniform)
        }
        // D[i = kk + 4 r][j = cc]: i < 9, j < 10 live -> sPartR[wave][10 i + j]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = kk + 4 * r;
            if (i < 9 && cc < 10) sPartR[wave * 90 + 10 * i + cc] = acc[r];
        }
    }
    __syncthreads();
    if (tid < 90) {
        double sres = (sPartR[tid] + sPartR[90 + tid]) + (sPartR[180 + tid] + sPartR[270 + tid]);
        if (NW == 8) sres += (sPartR[360 + tid] + sPartR[450 + tid]) + (sPartR[540 + tid] + sPartR[630 + tid]);   // (fixed order: deterministic)
        const int i = tid / 10, j = tid - 10 * i;
        if (j < 9) sT[36 * Kmax + 9 * i + j] = sres; else sYr[i] = sres;
    }
    __syncthreads();
    REFINE_STAMP(5);
}
template <int NW>
__device__ __forceinline__ void obj_refine_C(const int o, const int K, const int Kmax, const double* __restrict__ sT, const double* __restrict__ sYr,
                                             const int2* __restrict__ sTab, const double* __restrict__ q, const double* __restrict__ xr,
                                             const unsigned short* __restrict__ sK, const int2* __restrict__ sGrp, const int* __restrict__ sOvfp,
                                             const RefineArgs& a, double* __restrict__ Y, int NOP, int NAP, int NA) {
    constexpr int QS = 13;
    const int tid = threadIdx.x;
    const int KT = Kmax > 0 ? Kmax : 1;
    const int sOvf = *sOvfp;
    // ---- C: Y[:, col] from the rows, then Y'' = 1.5 Y - 0.5 T Y ----------------------------------------------------------------
    double* Yo = Y + (size_t)o * NOP * NAP;
    const double* Tb = sT + 36 * Kmax;
#pragma unroll 1
    for (int col = tid; col < NAP; col += 64 * NW) {
        if (col > NA) {
            for (int i = 0; i < NOP; ++i) Yo[(size_t)i * NAP + col] = 0.0;
            continue;
        }
        const bool rcol = col == NA;
        const int c = (!rcol && col >= a.cb0 && col < a.cb0 + 6 * a.N) ? (col - a.cb0) / 6 : -1;
        const int e = c >= 0 ? col - a.cb0 - 6 * c : 6;
        const int2 gr = c >= 0 ? sGrp[c] : int2{0, 0};
        double yb[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) yb[j] = rcol ? sYr[j] : 0.0;
#pragma unroll 1
        for (int l = gr.x; l < gr.y; l += 2) {   // (two rows in flight)
            double x[2], qv[2][9];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool ok = l + u < gr.y;
                const int lp = ok ? l + u : gr.x;
                const double x0 = xr[lp * 7 + e];
                x[u] = ok ? x0 : 0.0;
#pragma unroll
                for (int j = 0; j < 9; ++j) qv[u][j] = q[lp * QS + 3 + j];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int j = 0; j < 9; ++j) yb[j] += qv[u][j] * x[u];
            }
        }
        double tb[9];
#pragma unroll
        for (int c1 = 0; c1 < 9; ++c1) {
            double t = 0.0;
#pragma unroll
            for (int c2 = 0; c2 < 9; ++c2) t += Tb[9 * c1 + c2] * yb[c2];
            tb[c1] = t;
        }
        if (!sOvf) {   // (workgroup-uniform) at most two rows per (clone, keypoint): their positions come from the table
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                double yk[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) yk[j] = rcol ? sYr[9 + 3 * k + j] : 0.0;
                if (c >= 0) {
                    const int2 tb2 = sTab[c * KT + k];
                    const int la = tb2.x >= 0 ? tb2.x : 0, lb = tb2.y >= 0 ? tb2.y : 0;
                    const double xa = xr[la * 7 + e], xb = xr[lb * 7 + e];
                    const double wa = tb2.x >= 0 ? xa : 0.0, wb = tb2.y >= 0 ? xb : 0.0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) yk[j] = q[la * QS + j] * wa + q[lb * QS + j] * wb;
                }
                const double* Tk = sT + 36 * k;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double t = (Tk[3 * j] * yk[0] + Tk[3 * j + 1] * yk[1]) + Tk[3 * j + 2] * yk[2];
#pragma unroll
                    for (int c2 = 0; c2 < 9; ++c2) t += Tk[9 + 9 * j + c2] * yb[c2];
                    Yo[(size_t)(9 + 3 * k + j) * NAP + col] = 1.5 * yk[j] - 0.5 * t;
                }
#pragma unroll
                for (int c2 = 0; c2 < 9; ++c2) tb[c2] += (Tk[9 + c2] * yk[0] + Tk[18 + c2] * yk[1]) + Tk[27 + c2] * yk[2];
            }
        } else {
#pragma unroll 1
            for (int kc = 0; kc < K; kc += 4) {
                double yk[4][3];
    #pragma unroll
                for (int u = 0; u < 4; ++u) {
    #pragma unroll
                    for (int j = 0; j < 3; ++j) yk[u][j] = (rcol && kc + u < K) ? sYr[9 + 3 * (kc + u) + j] : 0.0;
                }
    #pragma unroll 1
                for (int l = gr.x; l < gr.y; l += 4) {   // (four rows in flight; a row adds to the slot of its keypoint, if that is one of these four)
                    int kq[4];
                    double q0[4], q1[4], q2[4];
    #pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const bool ok = l + w < gr.y;
                        const int lp = ok ? l + w : gr.x;
                        const int kv = (int)sK[lp];
                        kq[w] = ok ? kv - kc : -1;
                        const double x = xr[lp * 7 + e];
                        q0[w] = q[lp * QS] * x; q1[w] = q[lp * QS + 1] * x; q2[w] = q[lp * QS + 2] * x;
                    }
    #pragma unroll
                    for (int w = 0; w < 4; ++w) {
    #pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool hit = kq[w] == u;
                            yk[u][0] += hit ? q0[w] : 0.0; yk[u][1] += hit ? q1[w] : 0.0; yk[u][2] += hit ? q2[w] : 0.0;
                        }
                    }
                }
    #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kc + u;
                    if (k < K) {
                        const double* Tk = sT + 36 * k;
    #pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            double t = (Tk[3 * j] * yk[u][0] + Tk[3 * j + 1] * yk[u][1]) + Tk[3 * j + 2] * yk[u][2];
    #pragma unroll
                            for (int c2 = 0; c2 < 9; ++c2) t += Tk[9 + 9 * j + c2] * yb[c2];
                            Yo[(size_t)(9 + 3 * k + j) * NAP + col] = 1.5 * yk[u][j] - 0.5 * t;
                        }
    #pragma unroll
                        for (int c2 = 0; c2 < 9; ++c2) tb[c2] += (Tk[9 + c2] * yk[u][0] + Tk[18 + c2] * yk[u][1]) + Tk[27 + c2] * yk[u][2];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) Yo[(size_t)j * NAP + col] = 1.5 * yb[j] - 0.5 * tb[j];
        for (int i = 9 + 3 * K; i < NOP; ++i) Yo[(size_t)i * NAP + col] = 0.0;
    }
    REFINE_STAMP(6);
}
template <int NW>
__device__ __forceinline__ void obj_refine_BC(const int o, const int K, const int m, const int Kmax, double* __restrict__ sT, double* __restrict__ sYr,
                                              const int2* __restrict__ sTab, double* __restrict__ sPartR, const double* __restrict__ q,
                                              const double* __restrict__ xr, const unsigned short* __restrict__ sList,
                                              const unsigned short* __restrict__ sK, const int2* __restrict__ sRange, const int p0,
                                              const int2* __restrict__ sGrp, const int* __restrict__ sOvfp, const RefineArgs& a,
                                              double* __restrict__ Y, int NOP, int NAP, int NA) {
    obj_refine_B<NW>(o, K, m, Kmax, sT, sYr, sPartR, q, xr, sList, sRange, p0, a);
    obj_refine_C<NW>(o, K, Kmax, sT, sYr, sTab, q, xr, sK, sGrp, sOvfp, a, Y, NOP, NAP, NA);
}

// sR: the object's arrow factor (with the pivot tolerance); sT (arrow_stride doubles), sYr (NOP doubles): LDS scratch; rowbuf: the
// staging of this object's rows (LDS or global), m * OBJ_REFINE_ROW_DOUBLES doubles.  All 256 threads; writes the object's block of Y.
template <bool INLDS>
__device__ __forceinline__ void obj_refine_body(const int o, const ObjArrow ob, const int Kmax, const double* __restrict__ sR,
                                                double* __restrict__ sT, double* __restrict__ sYr, int2* __restrict__ sTab, double* __restrict__ sPartR,
                                                double* rowbuf, const RefineArgs& a, double* __restrict__ Y, int NOP, int NAP, int NA) {
    __shared__ int2 sRange[36];
    __shared__ int2 sGrp[ORCVIO_MAX_CLONES];
    __shared__ int sOvf;   // some (clone, keypoint) pair has more than two rows (two frames of the object share a clone, or rows handed over
                           // through orcvio_msckf_update_objects in another shape): phase C then takes the general form
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = ob.K, m = ob.rows, row0 = ob.row0;
    const int KT = Kmax > 0 ? Kmax : 1;   // sTab[c * KT + k] = the (at most two) positions of the rows of keypoint k in clone c, -1: none
    const double tol = sR[36 * Kmax + 81];
    constexpr int QS = 13;              // row stride of q: odd, so that the groups of different clones start on different LDS banks
    double* q = rowbuf;                 // [m][QS]  3 keypoint entries, 9 border entries
    double* xr = rowbuf + (size_t)m * QS;   // [m][7]  hx (6), r
    typedef unsigned short u16;         // (an object has at most 2 048 rows, 34 keypoints, 60 clones)
    u16* sPos = reinterpret_cast<u16*>(rowbuf + (size_t)m * (QS + 7));   // [m] row - row0 -> position
    u16* sList = sPos + m;              // [m] the keypoint lists as positions
    u16* sK = sList + m;                // [m] keypoint block of the row at a position (K: border only)
    u16* sCl = sK + m;                  // [m] clone of the row at a position
    REFINE_STAMP(1);
    if (tid <= K) sRange[tid] = a.kp_range[ob.kp_off + tid];
    if (tid < ORCVIO_MAX_CLONES) sGrp[tid] = int2{0, 0};
    if (tid == 0) sOvf = 0;
    for (int i = tid; i < a.N * KT; i += 256) sTab[i] = int2{-1, -1};
    __syncthreads();
    for (int g = tid; g < a.ngroups; g += 256) {
        const ObjGroup gq = a.groups[g];
        if (gq.obj == o) {
            sGrp[gq.clone] = int2{gq.r0 - row0, gq.r1 - row0};
            for (int lp = gq.r0 - row0; lp < gq.r1 - row0; ++lp) sCl[lp] = (u16)gq.clone;
        }
    }
    // ---- P: positions, [hx | r] in group order; reciprocals of the kept pivots ---------------------------------------------------
    if (tid < 9 + 3 * K) {
        const double pv = tid < 9 ? sR[36 * Kmax + 10 * tid] : sR[36 * ((tid - 9) / 3) + 13 * ((tid - 9) % 3)];
        sYr[tid] = fabs(pv) > tol ? 1.0 / pv : 0.0;
    }
    for (int lp = tid; lp < m; lp += 256) {
        const int row = a.ridx[row0 + lp];
        sPos[row - row0] = (u16)lp;
        double v[7];
#pragma unroll
        for (int e = 0; e < 6; ++e) v[e] = a.Hx6[(size_t)row * 6 + e];
        v[6] = a.HfR[(size_t)row * a.ldf + a.no_max];
#pragma unroll
        for (int e = 0; e < 7; ++e) xr[lp * 7 + e] = v[e];
    }
    __syncthreads();
    REFINE_STAMP(2);
    // ---- A: the rows of Q~ (thread per entry of the keypoint lists).  (Straight-line code that runs once is bound by instruction
    // fetch: every loop here is kept rolled except where a register array needs constant indices.) ------------------------------------
    const int p0 = sRange[0].x, p1 = sRange[K].y;   // (the K + 1 row lists of an object are contiguous in kp_rows)
    const double* Rb = sR + 36 * Kmax;
    const double* sRi = sYr;   // reciprocals of the pivots (0: dropped), written in phase P: [9 border | 3 per keypoint]
#pragma unroll 1
    for (int p = p0 + tid; p < p1; p += 256) {
        int k = 0;
        while (k < K && p >= sRange[k].y) ++k;
        const int row = a.kp_rows[p];
        const int lp = sPos[row - row0];
        sList[p - p0] = (u16)lp;
        sK[lp] = (u16)k;
        if (k < K) {   // the row's slot in the (clone, keypoint) table
            int* slot = reinterpret_cast<int*>(sTab + (int)sCl[lp] * KT + k);
            if (atomicCAS(slot, -1, lp) != -1 && atomicCAS(slot + 1, -1, lp) != -1) sOvf = 1;
        }
        const double* h = a.HfR + (size_t)row * a.ldf;
        double hb[9], qk[3] = {0.0, 0.0, 0.0}, qb[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) hb[c] = h[c];
        if (k < K) {
            const double* Rk = sR + 36 * k;
            const double* rik = sRi + 9 + 3 * k;
            double hk[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) hk[j] = h[9 + 3 * k + j];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double t = hk[j];
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (i < j) t -= qk[i] * Rk[12 * i + j];
                qk[j] = t * rik[j];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) hb[c] -= (qk[0] * Rk[3 + c] + qk[1] * Rk[15 + c]) + qk[2] * Rk[27 + c];
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            double t = hb[j];
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i < j) t -= qb[i] * Rb[9 * i + j];
            qb[j] = t * sRi[j];
        }
        double* qo = q + lp * QS;
#pragma unroll
        for (int j = 0; j < 3; ++j) qo[j] = qk[j];
#pragma unroll
        for (int j = 0; j < 9; ++j) qo[3 + j] = qb[j];
    }
    __syncthreads();
    REFINE_STAMP(3);
    obj_refine_BC<4>(o, K, m, Kmax, sT, sYr, sTab, sPartR, q, xr, sList, sK, sRange, p0, sGrp, &sOvf, a, Y, NOP, NAP, NA);
}
// sDyn: [factor (arrow_stride) | T (arrow_stride) | Q~^T r (NOP) | (clone, keypoint) table (N * max(Kmax, 1) int2) | row staging
// (lds_rows * OBJ_REFINE_ROW_DOUBLES)]; the factor is in place.
// Returns true if the object was refined (uniform over the workgroup).
__device__ __forceinline__ bool obj_refine_if_needed(const int o, const ObjArrow ob, const int Kmax, double* __restrict__ sDyn,
                                                     double* __restrict__ sScratch /* [OBJ_REFINE_SCRATCH] */, const RefineArgs& a,
                                                     double* __restrict__ Y, int NOP, int NAP, int NA) {
    __shared__ double sCond[1];
    const int astr = arrow_stride(Kmax);
    double* sR = sDyn;
    double* sT = sDyn + astr;
    double* sYr = sT + astr;
    int2* sTab = reinterpret_cast<int2*>(sYr + NOP);   // [N][max(Kmax, 1)]
    double* sRows = sYr + NOP + (size_t)a.N * (Kmax > 0 ? Kmax : 1);
    if (a.mode != 2) {
        const double c2 = obj_arrow_cond2(sR, sT, sCond, ob.K, Kmax);
        if (!(c2 > OBJ_REFINE_COND * OBJ_REFINE_COND)) return false;   // (NaN: the fast route's result stands)
        __syncthreads();   // (sT is rewritten)
    }
    if (threadIdx.x == 0 && a.refined) atomicAdd(a.refined, 1);
    if (ob.rows <= a.lds_rows) obj_refine_body<true>(o, ob, Kmax, sR, sT, sYr, sTab, sScratch, sRows, a, Y, NOP, NAP, NA);
    else obj_refine_body<false>(o, ob, Kmax, sR, sT, sYr, sTab, sScratch, a.scratch + (size_t)ob.row0 * OBJ_REFINE_ROW_DOUBLES, a, Y, NOP, NAP, NA);
    return true;
}
// The stand-alone launch (windows wider than 256 columns, where border QR and substitution are separate launches): one workgroup per
// object, over the Y the substitution wrote.
__global__ __launch_bounds__(256) void k_obj_refine(const ObjArrow* __restrict__ objs, int Kmax, const double* __restrict__ Rarrow, RefineArgs a,
                                                    double* __restrict__ Y, int NOP, int NAP, int NA) {
    extern __shared__ double sRef[];
    __shared__ double sScratch[OBJ_REFINE_SCRATCH];
    const int o = blockIdx.x, astr = arrow_stride(Kmax);
    const double* Ro = Rarrow + (size_t)o * astr;
    for (int i = threadIdx.x; i < astr; i += 256) sRef[i] = Ro[i];
    __syncthreads();
    obj_refine_if_needed(o, objs[o], Kmax, sRef, sScratch, a, Y, NOP, NAP, NA);
}

// Border QR, Y = R^-T C and sum_o B_o in ONE launch (windows with NAP <= 256: one solve workgroup per object).  Workgroup o < nobj:
// the nine border reflectors of object o, then -- R_b and H_f^T r of the object are this workgroup's own writes -- the forward
// substitution; the others: the sum of the clone tiles.  |r|^2 summed over all (object, clone) tiles by the workgroup that owns the
// corner element (fixed assignment, LDS tree: deterministic).
template <int RPT>
__global__ __launch_bounds__(256) void k_obj_border_solve_assemble(const ObjArrow* __restrict__ objs, const double* __restrict__ Bred, int Kmax,
                                                                   double* __restrict__ Rarrow, const double* __restrict__ Hr,
                                                                   const double* __restrict__ Sg, int N, int NOP, double* __restrict__ Hfr,
                                                                   const double* __restrict__ Cd, int NAP, int NA, double* __restrict__ Y,
                                                                   int* __restrict__ info, int nobj, int cb0, double* __restrict__ Bdst, RefineArgs ra) {
    extern __shared__ double sR[];   // arrow_stride(Kmax) doubles (solve role) [+ the scratch of obj_refine_if_needed when ra.mode != 0]
    __shared__ double sShare[OBJ_REFINE_SCRATCH];   // object role: partial tiles of the explicit-basis route; assemble role: the corner's tree
    int b = blockIdx.x;
    if (b < nobj) {
        if (ra.stamps && b == 0 && threadIdx.x == 0) ra.stamps[7] = wall_clock64();
        obj_border_qr_body<RPT>(b, objs, Bred, Kmax, Rarrow, Hr, Sg, N, NOP, Hfr, sR);
        __syncthreads();   // (R_b and the tolerance in LDS; drains this workgroup's stores of H_f^T r)
        if (ra.stamps && b == 0 && threadIdx.x == 0) ra.stamps[0] = wall_clock64();
        // an ill-conditioned Hf takes Y from the explicit basis (and counts its dropped pivots like the substitution would)
        if (ra.mode != 0 && obj_refine_if_needed(b, objs[b], Kmax, sR, sShare, ra, Y, NOP, NAP, NA)) {
            if (threadIdx.x == 0 && info) {
                const ObjArrow ob = objs[b];
                const double tol = sR[36 * Kmax + 81];
                int dropped = 0;
                for (int e = 0; e < 3 * ob.K + 9; ++e) {
                    const double pv = e < 3 * ob.K ? sR[36 * (e / 3) + 13 * (e % 3)] : sR[36 * Kmax + 10 * (e - 3 * ob.K)];
                    dropped += fabs(pv) > tol ? 0 : 1;
                }
                if (dropped > 0) atomicAdd(info, dropped);
            }
            return;
        }
        obj_arrow_solve_body(sR, b, (int)threadIdx.x, objs, Rarrow, Kmax, Cd, NOP, NAP, NA, Hfr, Y, info, true);
        return;
    }
    b -= nobj;
    double* sCorner = sShare;
    const int corner = NA * NAP + NA;
    const bool corner_block = corner >= b * 256 && corner < b * 256 + 256;   // (workgroup-uniform)
    if (corner_block) {
        double sacc = 0.0;
        for (int q = threadIdx.x; q < nobj * N; q += 256) sacc += Sg[(size_t)q * 64 + 54];
        sCorner[threadIdx.x] = sacc;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) sCorner[threadIdx.x] += sCorner[threadIdx.x + w];
            __syncthreads();
        }
    }
    obj_assemble_B_body(b * 256 + (int)threadIdx.x, Sg, nobj, N, nullptr, 0, cb0, NA, NAP, Bdst, corner_block ? sCorner : nullptr);
}

}  // namespace orcvio_amd

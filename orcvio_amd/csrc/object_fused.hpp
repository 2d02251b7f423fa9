// object_fused.hpp -- the object tracks' compression in ONE launch, one workgroup per object (VERDICT r4 'next' 2).
// Reference: the export block of ObjectFeatureInitializer::single_levenberg_marquardt (src/obj/ObjectFeatureInitializer.cpp:394-434),
// OrcVIO::constructObjectResidualJacobians (src/orcvio.cpp:2017-2151) and the left-nullspace projection of removeLostObjects
// (src/orcvio.cpp:2154-2193; include/orcvio/utils/math_utils.hpp:287-312).
//
// Round 4 ran this as three launches over materialised rows: k_object_rows_batch wrote every row scattered to the object's 48 columns
// (8.7 MB for config 3, 10 structural non-zeros of 48 per row), k_obj_front read them back for the cross products and the keypoint
// blocks of the structured QR, k_obj_border_solve_assemble finished the QR and the projection (50 us in all, ~100 x the 0.13 MB of
// track data in traffic).  Here the eight wavefronts of a workgroup evaluate the object's rows straight into LDS -- 12 structural
// numbers of H_f, the six window columns and the residual per row; an object of config 3 is 840 rows, 141 KB -- and everything else
// happens on them where they stand:
//   P1  rows: 64 / lpf frames per wavefront side by side (object_rows_lane), lpf = 16 lanes per frame for the 12-keypoint car class
//   P2  the keypoint blocks of the structured Householder QR of H_f (three reflectors each, rows in registers), one per wavefront
//   P3  the nine border reflectors over all rows, which stay in the registers of the threads that hold them after P2
//   P4  Y = Q1^T [Hx | r] through the explicit basis Q~ = H_f R^-1 with the first-order correction for its loss of orthogonality
//       (obj_refine_BC; round 4 took this route for ill-conditioned objects only and the semi-normal equations otherwise: one route
//       now, the accurate one), the clone tiles of B = X^T X
// Nothing of an object's rows touches HBM; the launch writes Y (NOP x NAP per object), the 7 x 7 clone tiles and |r|^2.
// The launch serves objects whose in-window frames map to distinct clones, with at most 64 rows per keypoint (<= 32 frames), at most
// 16 keypoints and rows that fit the LDS staging; anything else takes the three-launch pipeline (capi_objects.inc decides per update).
#pragma once

namespace orcvio_amd {

#define OBJ_FUSED_NW 8          // wavefronts per workgroup
#define OBJ_FUSED_MAXF 32       // in-window frames per object (two rows per keypoint and frame, one row per lane in P2)
#define OBJ_FUSED_KPW 2         // keypoint blocks a wavefront takes in P2 (K <= 16)
struct ObjFusedArgs {
    const ObjEvalArgs* args;    // per object: the track (device pointers), flags, extrinsics
    const ObjArrow* objs;       // per object: row0, rows, K
    int nobj, N, cb0, NA, NAP, NOP, Kmax;
    double* Y;                  // [nobj][NOP][NAP]
    double* Sg;                 // [nobj][N][64] clone tiles [hx | r]^T [hx | r]
    double* rr;                 // |r|^2 of object o at rr[o * rr_stride]
    int rr_stride;
    int* info;                  // [0] += dropped pivots, [1] += objects above OBJ_REFINE_COND (mode 1) / all objects (mode 2)
    int mode;                   // ORCVIO_OPT_OBJECT_REFINE (1 or 2; 0 never comes here)
    unsigned long long* stamps; // diagnostics: wall-clock stamps of object 0's phases
    double tol_rel;             // pivot tolerance relative to the largest pivot (1e-10; ORCVIO_FUSED_TOL for tests of the verification)
    double* dbg;                // diagnostics (ORCVIO_FUSED_DBG): [64] pivots of object 0 (9 border, 3 per keypoint), then the tolerance
};
#define FUSED_STAMP(i) do { if (fa.stamps && o == 0 && threadIdx.x == 0) fa.stamps[i] = wall_clock64(); } while (0)

// dynamic LDS: [factor (astr) | T (astr) | Q~^T r (NOP) | (clone, keypoint) table (N * max(Kmax, 1) int2) | rows (m * OBJ_REFINE_ROW_DOUBLES)]
__host__ __device__ inline size_t obj_fused_lds_doubles(int Kmax, int NOP, int N, int rows) {
    return 2 * (size_t)arrow_stride(Kmax > 0 ? Kmax : 1) + NOP + (size_t)N * (Kmax > 0 ? Kmax : 1) + (size_t)rows * OBJ_REFINE_ROW_DOUBLES;
}

// 1 / x: v_rcp_f64 seed + Newton steps (the seeds of v_rcp_f64 / v_rsq_f64 carry more than 26 bits: one step reaches double precision to
// a few ulp -- which is all a reflector's scalars need: an error of a few ulp in tau / scale is an error of a few ulp in R)
__device__ __forceinline__ double rcp_nr1(double d) {
    const double y = __builtin_amdgcn_rcp(d);
    return y + y * (1.0 - d * y);
}
__device__ __forceinline__ double rsqrt_nr1(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    return y + (0.5 * y) * (1.0 - d * (y * y));
}
// ---- Householder QR of rows held by a wavefront in (row group, column) layout ---------------------------------------------------
// Lane l = 16 g + c holds, in register x[s], entry c of row 4 s + g (NS rows per group: 4 NS rows per wavefront, 16 columns of which
// the callers use 12).  One v_fmac_f64 with a DPP row_newbcast source multiplies a row's pivot-column entry into all its columns, so
// the dot products of reflector j with EVERY column cost one instruction per register row, and so does its application; the four
// row groups are summed by one matrix instruction with an all-ones A (D[i][n] = sum_k B[k][n]: every lane gets the total of its
// column).  Round 5's first version (a row per lane, nine columns in registers, every dot product its own 20-instruction DPP
// reduction) spent 8 800 instructions in the two QR phases; this form about 2 000.
// Reflector j: pivot column JC0 + j, pivot row = register row j of group 0 (rows that do not exist hold zeros: appending zero rows
// changes no R).  LAPACK dlarfg quantities from one reciprocal square root and one reciprocal.
template <int C>
__device__ __forceinline__ void dpp_fmac(double& acc, double a, double b) {   // acc += a[lane C of the 16-lane row] * b
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(C));
}
template <int C>
__device__ __forceinline__ void dpp_fmac_nop(double& acc, double a, double b) {   // ... with the wait states a freshly written `a` needs
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(C));
}
__device__ __forceinline__ double groups_sum(double v) {   // sum over the four row groups, per column, into every lane
    const d4 z = {0, 0, 0, 0};
    return mfma_f64(1.0, v, z)[0];
}
template <int C>
__device__ __forceinline__ double row_bcast_b(double v) {   // lane C of every 16-lane row to all lanes of the row (row_newbcast)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + C, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + C, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// XW > 0: the rows are spread over XW wavefronts of the workgroup (the pivot rows in wavefront 0): the column sums of the wavefronts
// meet in LDS (sRed: 2 x (XW + 1) x 16 doubles, double-buffered by the reflector's parity: ONE workgroup barrier per reflector), summed
// in wave order by every wavefront for itself.
template <int NS, int JC0, int NJ, int J, int XW = 0>
struct RcHouse {
    static __device__ __forceinline__ void run(double (&x)[NS], const int lane, double& pmax, double* __restrict__ sRed = nullptr, const int wave = 0) {
        if constexpr (J < NJ) {
            constexpr int JC = JC0 + J;
            const bool g0 = lane < 16 && (XW == 0 || wave == 0);
            const int c = lane & 15;
            // dot products of column JC (rows below the pivot: every row but the pivots 0..J of group 0) with all columns
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            double xm[J + 1];
#pragma unroll
            for (int s2 = 0; s2 <= J; ++s2) xm[s2] = g0 ? 0.0 : x[s2];
#pragma unroll
            for (int s2 = J + 1; s2 < NS; ++s2) {
                if ((s2 & 3) == 0) dpp_fmac<JC>(a0, x[s2], x[s2]);
                else if ((s2 & 3) == 1) dpp_fmac<JC>(a1, x[s2], x[s2]);
                else if ((s2 & 3) == 2) dpp_fmac<JC>(a2, x[s2], x[s2]);
                else dpp_fmac<JC>(a3, x[s2], x[s2]);
            }
#pragma unroll
            for (int s2 = 0; s2 <= J; ++s2) {
                if (s2 == 0) dpp_fmac_nop<JC>(a0, xm[s2], xm[s2]);
                else if ((s2 & 3) == 1) dpp_fmac<JC>(a1, xm[s2], xm[s2]);
                else if ((s2 & 3) == 2) dpp_fmac<JC>(a2, xm[s2], xm[s2]);
                else if ((s2 & 3) == 3) dpp_fmac<JC>(a3, xm[s2], xm[s2]);
                else dpp_fmac<JC>(a0, xm[s2], xm[s2]);
            }
            const double part = (a0 + a1) + (a2 + a3);
            double g = groups_sum(part);                            // g_c = sum over the rows below of x[row][JC] x[row][c]
            double prow;
            if constexpr (XW == 0) {
                prow = groups_sum(g0 ? x[J] : 0.0);                 // the pivot row, in every group
            } else {
                double* buf = sRed + (J & 1) * (XW + 1) * 16;
                if (lane < 16) {
                    buf[wave * 16 + lane] = g;
                    if (wave == 0) buf[XW * 16 + lane] = x[J];
                }
                __syncthreads();
                const int c16 = lane & 15;
                g = 0.0;
#pragma unroll
                for (int w2 = 0; w2 < XW; ++w2) g += buf[w2 * 16 + c16];   // wave order: deterministic, the same in every wavefront
                prow = buf[XW * 16 + c16];
            }
            const double alpha = row_bcast_b<JC>(prow), gj = row_bcast_b<JC>(g);   // (builtins: the compiler places the wait states behind the matrix instruction)
            double tau = 0.0, beta = alpha, scale = 0.0;
            if (gj > 0.0) {
                const double s2n = alpha * alpha + gj;
                const double ri = rsqrt_nr1(s2n), nrm = s2n * ri, aa = fabs(alpha);
                beta = alpha >= 0.0 ? -nrm : nrm;
                tau = 1.0 + aa * ri;
                const double rc = rcp_nr1(aa + nrm);
                scale = alpha >= 0.0 ? rc : -rc;
            }
            const double w = tau * (prow + scale * g);               // what the pivot row loses in column c (c > JC)
            // rows below: x[row][c] -= w_c * scale * x[row][JC] for c > JC, the pivot column itself to zero (factor 1), earlier columns
            // stay as they are (factor 0)
            double ws = c > JC ? w * scale : (c == JC ? 1.0 : 0.0);
            if (!(gj > 0.0)) ws = 0.0;                               // (nothing below the pivot: H = I)
            const double ws0 = g0 ? 0.0 : ws;                        // (the earlier pivot rows of group 0 are not touched)
            const double pv = c > JC ? x[J] - w : (c == JC ? beta : x[J]);
#pragma unroll
            for (int s2 = J + 1; s2 < NS; ++s2) dpp_fnmac<JC>(x[s2], x[s2], ws);
#pragma unroll
            for (int s2 = 0; s2 <= J; ++s2) {
                if (s2 == 0) asm volatile("s_nop 1");
                dpp_fnmac<JC>(x[s2], x[s2], ws0);
            }
            x[J] = g0 ? pv : x[J];
            pmax = fmax(pmax, fabs(beta));
            RcHouse<NS, JC0, NJ, J + 1, XW>::run(x, lane, pmax, sRed, wave);
        }
    }
};

__global__ __launch_bounds__(512) void k_obj_fused(ObjFusedArgs fa) {
    extern __shared__ double sDyn[];
    __shared__ int2 sRange[36];
    __shared__ int2 sGrp[ORCVIO_MAX_CLONES];
    __shared__ int sOvf;
    __shared__ short sFr[OBJ_FUSED_MAXF], sFrCl[OBJ_FUSED_MAXF];
    __shared__ int sCnt[36];
    __shared__ double sPartR[OBJ_FUSED_NW * 90];
    __shared__ double sMax[OBJ_FUSED_NW];
    __shared__ int sFin;
    typedef unsigned short u16;
    constexpr int QS = 13, NW = OBJ_FUSED_NW;
    const int o = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ObjEvalArgs p = fa.args[o];
    const ObjArrow ob = fa.objs[o];
    const int K = ob.K, m = ob.rows, N = fa.N, Kmax = fa.Kmax, KT = Kmax > 0 ? Kmax : 1, NOP = fa.NOP;
    const int astr = arrow_stride(KT);
    double* sR = sDyn;
    double* sT = sDyn + astr;
    double* sYr = sT + astr;
    int2* sTab = reinterpret_cast<int2*>(sYr + NOP);
    double* rowbuf = sYr + NOP + (size_t)N * KT;
    double* q = rowbuf;                       // [m][QS]: [3 keypoint entries | 9 border entries] of H_f, later of Q~
    double* xr = rowbuf + (size_t)m * QS;     // [m][7]:  hx (6), r
    u16* sList = reinterpret_cast<u16*>(rowbuf + (size_t)m * (QS + 7)) + m;   // (same places as obj_refine_body's staging)
    u16* sK = sList + m;
    u16* sCl = sK + m;
    FUSED_STAMP(0);
    // ---- P0: the in-window frames in frame order, tables cleared ---------------------------------------------------------------
    if (wave == 0) {
        int pos = 0;
        for (int f0 = 0; f0 < p.F; f0 += 64) {
            const int f = f0 + lane;
            const int cl = f < p.F ? p.frame_clone[f] : -1;
            const unsigned long long mk = __ballot(cl >= 0);
            if (cl >= 0) {
                const int at = pos + __popcll(mk & ((1ull << lane) - 1ull));
                if (at < OBJ_FUSED_MAXF) { sFr[at] = (short)f; sFrCl[at] = (short)cl; }
            }
            pos += __popcll(mk);
        }
        if (lane == 0) { sFin = pos < OBJ_FUSED_MAXF ? pos : OBJ_FUSED_MAXF; sOvf = 0; }
    }
    if (tid < ORCVIO_MAX_CLONES) sGrp[tid] = int2{0, 0};
    for (int i = tid; i < N * KT; i += 64 * NW) sTab[i] = int2{-1, -1};
    __syncthreads();
    const int Fin = sFin;
    // ---- P1: the rows.  lpf lanes per frame: the smallest power of two that holds K keypoint lanes and four bbox lanes -----------
    {
        int lpf = 4;
        while (lpf < K + 4) lpf <<= 1;
        const int fpw = 64 / lpf;
        for (int base = 0; base < Fin; base += NW * fpw) {
            const int i = base + wave * fpw + lane / lpf, t = lane & (lpf - 1);
            const int f = i < Fin ? (int)sFr[i] : -1;
            const int cl = i < Fin ? (int)sFrCl[i] : 0;
            const int lp0 = f >= 0 ? p.frame_row0[f] - ob.row0 : 0;   // first row of the frame among the object's rows
            object_rows_lane(p, f, t, lpf, [&](int rif, double r, const double* hx6, const double* hpose, const double* hshape, int kpid,
                                               const double* hkp, int nrows) {
                const int lp = lp0 + rif;
                double* qo = q + lp * QS;
#pragma unroll
                for (int c = 0; c < 3; ++c) qo[c] = hkp ? hkp[c] : 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) qo[3 + c] = hpose[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) qo[9 + c] = hshape ? hshape[c] : 0.0;
                double* xo = xr + lp * 7;
#pragma unroll
                for (int c = 0; c < 6; ++c) xo[c] = hx6[c];
                xo[6] = r;
                sK[lp] = (u16)(kpid >= 0 ? kpid : K);
                sCl[lp] = (u16)cl;
                if (kpid >= 0 && (rif & 1) == 0) sTab[cl * KT + kpid] = int2{lp, lp + 1};   // the two rows of keypoint kpid in this clone
                if (kpid < 0 && rif == nrows - 4) sGrp[cl] = int2{lp0, lp0 + nrows};           // (the first bbox lane knows the frame's rows)
            });
        }
    }
    __syncthreads();
    FUSED_STAMP(1);
    FUSED_STAMP(9);
    // ---- the keypoint lists (positions in frame order): counts, offsets, entries -------------------------------------------------
    for (int k = wave; k < K; k += NW) {
        const int v = lane < Fin ? sTab[(int)sFrCl[lane] * KT + k].x : -1;
        const unsigned long long mk = __ballot(v >= 0);
        if (lane == 0) sCnt[k] = 2 * __popcll(mk);
    }
    __syncthreads();
    if (tid == 0) {
        int at = 0;
        for (int k = 0; k < K; ++k) { sRange[k] = int2{at, at + sCnt[k]}; at += sCnt[k]; }
        sRange[K] = int2{at, at};
    }
    __syncthreads();
    for (int k = wave; k < K; k += NW) {
        const int v = lane < Fin ? sTab[(int)sFrCl[lane] * KT + k].x : -1;
        const unsigned long long mk = __ballot(v >= 0);
        if (v >= 0) {
            const int at = sRange[k].x + 2 * __popcll(mk & ((1ull << lane) - 1ull));
            sList[at] = (u16)v;
            sList[at + 1] = (u16)(v + 1);
        }
    }
    __syncthreads();
    FUSED_STAMP(10);
    // ---- P2: keypoint blocks.  Wavefront w takes keypoints w, w + NW; the rows of a keypoint (list order, <= 64) in (group, column)
    // layout: register row s of group g = list position 4 s + g, lane column c = entry c of the row's 12 numbers [3 keypoint | 9
    // border].  Three reflectors on columns 0..2, applied to all twelve: R_kk | R_kb in the pivot rows, the border of the other rows
    // eliminated in place (registers).
    constexpr int KS = 16, BS = 4, XS = OBJ_FUSED_KPW * KS + BS;   // register rows per keypoint / of bbox rows / per wavefront
    const int grp = lane >> 4, col = lane & 15;
    double x[XS];
#pragma unroll
    for (int u = 0; u < OBJ_FUSED_KPW; ++u) {
        const int k = wave + NW * u;
        double xk[KS];
        const int e0 = k < K ? sRange[k].x : 0, mk = k < K ? sRange[k].y - e0 : 0;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
            const int pos = 4 * s2 + grp;
            const bool in = pos < mk && col < 12;
            const int lp = in ? (int)sList[e0 + pos] : 0;
            const double v = q[lp * QS + (col < 12 ? col : 0)];
            xk[s2] = in ? v : 0.0;
        }
        if (k < K) {   // (wave-uniform)
            double pm = 0.0;
            RcHouse<KS, 0, 3, 0>::run(xk, lane, pm);
            if (lane < 12) {   // group 0: rows 0..2 = [3 of R_kk | 9 of R_kb]
#pragma unroll
                for (int j = 0; j < 3; ++j) sR[36 * k + 12 * j + lane] = (lane < 3 && lane < j) ? 0.0 : xk[j];
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) x[u * KS + s2] = (s2 < 3 && lane < 16) ? 0.0 : xk[s2];   // (the three pivot rows are consumed)
    }
    // ... and this wavefront's share of the border-only rows (four per frame, 16 per wavefront): bbox row t = 16 w + 4 s + g
#pragma unroll
    for (int s2 = 0; s2 < BS; ++s2) {
        const int t = 16 * wave + 4 * s2 + grp;
        const bool in = t < 4 * Fin && col < 12;
        const int lp = in ? sGrp[(int)sFrCl[t >> 2]].y - 4 + (t & 3) : 0;
        const double v = q[lp * QS + (col < 12 ? col : 0)];
        x[OBJ_FUSED_KPW * KS + s2] = in ? v : 0.0;
    }
    FUSED_STAMP(2);
    // ---- P3: the border factor R_b: nine reflectors over ALL rows of the object, which stay in the registers of the wavefronts that
    // hold them (border columns = lane columns 3..11); per reflector one matrix instruction sums a wavefront's row groups, the eight
    // wavefronts' sums meet in LDS behind ONE barrier.  (A two-level TSQR -- every wavefront its own factor, then wavefront 0 the eight
    // triangles -- had no barriers but eighteen reflector latencies in sequence: 10 us against 5.)
    double pmax = 0.0;
    RcHouse<XS, 3, 9, 0, OBJ_FUSED_NW>::run(x, lane, pmax, sPartR, wave);
    if (wave == 0 && lane >= 3 && lane < 12) {
#pragma unroll
        for (int j = 0; j < 9; ++j) sR[36 * Kmax + 9 * j + (lane - 3)] = lane - 3 >= j ? x[j] : 0.0;
    }
    // (unused keypoint blocks of the factor -- objects with fewer keypoints than Kmax -- read as zero)
    for (int i = 36 * K + tid; i < 36 * Kmax; i += 64 * NW) sR[i] = 0.0;
    __syncthreads();
    // pivot tolerance, dropped pivots, reciprocals of the kept ones [9 border | 3 per
    // keypoint]: wavefront 0, one pivot per lane (9 + 3 K <= 57)
    if (wave == 0) {
        const bool has = lane < 9 + 3 * K;
        const double pv = has ? (lane < 9 ? sR[36 * Kmax + 10 * lane] : sR[36 * ((lane - 9) / 3) + 13 * ((lane - 9) % 3)]) : 0.0;
        double mx = fabs(pv);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        // 1e-10 of the largest pivot.  The three-launch pipeline drops at 1e-11 (obj_border_qr_body: real pivots of the reference's own
        // car stay above 4e-9 of the largest, the noise pivot of an exactly dependent border column had reached 1e-13 in its soaks).  Here
        // the soak met a noise pivot of 1.03e-11 (scripts/gpu_soak_objects.py seed 90033: a car seen in two frames, H_f of rank 44 with
        // a next-to-last pivot of 1.3e-7 that amplifies the rounding of the last) -- kept, it is a garbage column of the basis and the
        // update came back 3 % off.  A factor 10 above that noise, 40 below the smallest real pivot met.
        const double tol0 = fa.tol_rel * mx;
        if (fa.dbg && o == 0) { fa.dbg[lane] = has ? pv : 0.0; if (lane == 0) fa.dbg[64] = tol0; }
        if (lane == 0) sR[36 * Kmax + 81] = tol0;
    }
    __syncthreads();
    FUSED_STAMP(3);
    FUSED_STAMP(11);
    // The rank decision above is a threshold on the pivots of an UNPIVOTED factor: a noise pivot that slips through is a garbage column
    // of the basis (a unit vector in a direction H_f does not span) and a wrong update, silently.  So the decision is VERIFIED: with the
    // kept pivots Q~^T Q~ must be the identity to ~cond * eps; an entry further than 1e-3 from it means a dependent column was kept --
    // the rows are restored (h = q R), the tolerance raised a hundredfold, the basis formed again (at most twice).
#pragma unroll 1
    for (int attempt = 0;; ++attempt) {
    // reciprocals of the kept pivots [9 border | 3 per keypoint]
    if (tid < 9 + 3 * K) {
        const double pv = tid < 9 ? sR[36 * Kmax + 10 * tid] : sR[36 * ((tid - 9) / 3) + 13 * ((tid - 9) % 3)];
        sYr[tid] = fabs(pv) > sR[36 * Kmax + 81] ? 1.0 / pv : 0.0;
    }
    if (tid == 0) sOvf = 0;   // (free here: "some entry of T is far from the identity")
    __syncthreads();
    // A: the rows of Q~ in place (q_i R = h_i), a thread per row
    {
        const double* Rb = sR + 36 * Kmax;
        const double* sRi = sYr;
#pragma unroll 1
        for (int lp = tid; lp < m; lp += 64 * NW) {
            const int k = (int)sK[lp];
            double* h = q + lp * QS;
            double hb[9], qk[3] = {0.0, 0.0, 0.0}, qb[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) hb[c] = h[3 + c];
            if (k < K) {
                const double* Rk = sR + 36 * k;
                const double* rik = sRi + 9 + 3 * k;
                double hk[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) hk[j] = h[j];
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
#pragma unroll
            for (int j = 0; j < 3; ++j) h[j] = qk[j];
#pragma unroll
            for (int j = 0; j < 9; ++j) h[3 + j] = qb[j];
        }
    }
    __syncthreads();
    FUSED_STAMP(4);
    // B: T = Q~^T Q~ on the matrix cores.  Per keypoint k one accumulation chain over its rows with A = B = the row's twelve numbers
    // [q_k | q_b]: rows 0..2 of the product are T_kk | T_kb, rows 3..11 the rows' part of T_bb -- which keeps accumulating over the
    // wavefront's keypoints and its share of the bbox rows (obj_refine_B runs a second pass over all rows for T_bb: 390 matrix
    // instructions per object against 210 here).  The wavefronts' parts of T_bb are summed through LDS in wave order.
    {
        const int kk = lane >> 4, cc = lane & 15;
        d4 accB = {0, 0, 0, 0};
#pragma unroll 1
        for (int k = wave; k < K; k += NW) {
            const int e0 = sRange[k].x, e1 = sRange[k].y;
            d4 acc = accB;
            acc[0] = kk < 3 ? 0.0 : accB[0];
            double av[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {   // (<= 64 rows per keypoint) all operands are read before the first instruction issues
                const int ee = e0 + 4 * u + kk;
                const bool ok = ee < e1 && cc < 12;
                const int lp = (int)sList[ee < e1 ? ee : e0];
                const double v = q[lp * QS + (cc < 12 ? cc : 0)];
                av[u] = ok ? v : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (e0 + 4 * u < e1) acc = mfma_f64(av[u], av[u], acc);   // (wave-uniform)
            if (kk < 3 && cc < 12) sT[36 * k + (cc < 3 ? 3 * kk + cc : 9 + 9 * kk + (cc - 3))] = acc[0];
            accB = acc;
        }
        {   // this wavefront's bbox rows (t = 16 w + 4 u + kk, as in P2)
            double av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = 16 * wave + 4 * u + kk;
                const bool ok = t < 4 * Fin && cc >= 3 && cc < 12;
                const int lp = t < 4 * Fin ? sGrp[(int)sFrCl[t >> 2]].y - 4 + (t & 3) : 0;
                const double v = q[lp * QS + (cc < 12 ? cc : 0)];
                av[u] = ok ? v : 0.0;
            }
            d4 acc = accB;
            acc[0] = kk < 3 ? 0.0 : accB[0];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (16 * wave + 4 * u < 4 * Fin) acc = mfma_f64(av[u], av[u], acc);
            accB = acc;
        }
        // D[i = kk + 4 r][j = cc], i, j in 3..11 -> sPartR[wave][9 (i - 3) + (j - 3)]
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = kk + 4 * r;
            if (i >= 3 && cc >= 3 && cc < 12) sPartR[wave * 81 + 9 * (i - 3) + (cc - 3)] = accB[r];
        }
        __syncthreads();
        if (tid < 81) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += sPartR[w * 81 + tid];   // wave order: deterministic
            sT[36 * Kmax + tid] = t;
        }
        __syncthreads();
    }
    {   // every entry of T between two kept columns against the identity
        bool bad = false;
        for (int e = tid; e < 36 * K + 81; e += 64 * NW) {
            int ci, cj;
            double v;
            if (e < 36 * K) {
                const int k = e / 36, w = e - 36 * k;
                if (w < 9) { ci = 9 + 3 * k + w / 3; cj = 9 + 3 * k + w % 3; }
                else { ci = 9 + 3 * k + (w - 9) / 9; cj = (w - 9) % 9; }
                v = sT[e];
            } else {
                const int w = e - 36 * K;
                ci = w / 9; cj = w - 9 * ci;
                v = sT[36 * Kmax + w];
            }
            if (sYr[ci] != 0.0 && sYr[cj] != 0.0 && !(fabs(v - (ci == cj ? 1.0 : 0.0)) < 1e-3)) bad = true;
        }
        if (bad) sOvf = 1;
        __syncthreads();
        const bool again = sOvf != 0 && attempt < 2;
        __syncthreads();
        if (!again) break;
        // restore the rows: h = q R (a dropped column comes back without its residual: noise in a direction that is dependent anyway)
        const double* Rb = sR + 36 * Kmax;
#pragma unroll 1
        for (int lp = tid; lp < m; lp += 64 * NW) {
            const int k = (int)sK[lp];
            double* h = q + lp * QS;
            double qk[3], qb[9], hk[3] = {0.0, 0.0, 0.0}, hb[9];
#pragma unroll
            for (int c = 0; c < 3; ++c) qk[c] = h[c];
#pragma unroll
            for (int c = 0; c < 9; ++c) qb[c] = h[3 + c];
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    if (i <= c) t += qb[i] * Rb[9 * i + c];
                hb[c] = t;
            }
            if (k < K) {
                const double* Rk = sR + 36 * k;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double t = 0.0;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (i <= j) t += qk[i] * Rk[12 * i + j];
                    hk[j] = t;
                }
#pragma unroll
                for (int c = 0; c < 9; ++c) hb[c] += (qk[0] * Rk[3 + c] + qk[1] * Rk[15 + c]) + qk[2] * Rk[27 + c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) h[c] = hk[c];
#pragma unroll
            for (int c = 0; c < 9; ++c) h[3 + c] = hb[c];
        }
        if (tid == 0) sR[36 * Kmax + 81] *= 100.0;
        __syncthreads();
    }
    }   // (attempts)
    if (wave == 0) {   // counters: dropped pivots; every object of this launch is projected through the explicit basis
        const bool has = lane < 9 + 3 * K;
        const int dropped = __popcll(__ballot(has && sYr[has ? lane : 0] == 0.0));
        if (lane == 0 && fa.info) {
            atomicAdd(fa.info + 1, 1);
            if (dropped > 0) atomicAdd(fa.info, dropped);
        }
    }
    FUSED_STAMP(7);
    // Q~ is orthonormal to cond * eps only: T = I + D.  First-order orthonormalisation of the ROWS, Q~ <- Q~ (I - U) with
    // U = strict_upper(D) + diag(D) / 2, so that Q~^T Q~ = I + O(D^2) (U + U^T = D) -- the same first order as round 4's
    // Y'' = (1.5 I - 0.5 T) Y on the columns, but 78 multiply-adds per row of the object instead of 840 per column of the window, and
    // Y = Q~^T X is then a plain product for the matrix cores.  U keeps the arrow shape.
#pragma unroll 1
    for (int lp = tid; lp < m; lp += 64 * NW) {
        const int k = (int)sK[lp];
        double* h = q + lp * QS;
        double qk[3], qb[9], nk[3], nb[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) qk[c] = h[c];
#pragma unroll
        for (int c = 0; c < 9; ++c) { qb[c] = h[3 + c]; nb[c] = qb[c]; }
        const double* Tb = sT + 36 * Kmax;
#pragma unroll
        for (int j = 0; j < 9; ++j) {   // q_b (I - U_bb): column j takes rows i <= j of U_bb
            double t = 0.5 * (Tb[10 * j] - 1.0) * qb[j];
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i < j) t += qb[i] * Tb[9 * i + j];
            nb[j] -= t;
        }
        if (k < K) {
            const double* Tk = sT + 36 * k;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double t = 0.5 * (Tk[4 * j] - 1.0) * qk[j];
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (i < j) t += qk[i] * Tk[3 * i + j];
                nk[j] = qk[j] - t;
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) nb[c] -= (qk[0] * Tk[9 + c] + qk[1] * Tk[18 + c]) + qk[2] * Tk[27 + c];   // - q_k U_kb (U_kb = T_kb: above the diagonal)
#pragma unroll
            for (int c = 0; c < 3; ++c) h[c] = nk[c];
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) h[3 + c] = nb[c];
    }
    __syncthreads();
    FUSED_STAMP(5);
    // ---- C: Y = Q~^T [Hx | r], clone by clone.  Per clone c (its rows g0 .. g1): the nine BORDER rows of Y on the matrix cores --
    // A = the border entries of the clone's rows (9 of 16 live), B = [hx (6) | r] (7 live columns), four rows per instruction -- and the
    // clone's tile of B = X^T X ([hx | r]^T [hx | r]) from the same operands; the KEYPOINT rows of Y have two terms each (the two rows
    // of keypoint k in this clone, from the (clone, keypoint) table): 36 x 7 outputs, four per lane, plain multiply-adds.  The six
    // window columns of the clone go straight to Y; column 6 (Q~_c^T r_c) is summed over the clones of a wavefront in registers and
    // over the wavefronts through LDS (fixed order).  A clone the object is not seen in gets zeros (no rows, empty table).
    {
        const int kk = lane >> 4, cc = lane & 15;
        double* Yo = fa.Y + (size_t)o * NOP * fa.NAP;
        double* So = fa.Sg + (size_t)o * N * 64;
        d4 racc = {0, 0, 0, 0};
        double rr = 0.0;
        const int ncol = 9 + 3 * K;
        // the keypoint outputs of this lane: e = lane + 64 i < 3 K * 7 -> row 9 + e / 7 of Y (keypoint e / 21, entry (e / 7) % 3), column e % 7
        constexpr int KE = (3 * OBJ_FUSED_NW * OBJ_FUSED_KPW * 7 + 63) / 64;   // 6 for <= 16 keypoints
        int ek[KE], ej[KE], en[KE];
        double rk[KE];
#pragma unroll
        for (int i = 0; i < KE; ++i) {
            const int e = lane + 64 * i, kj = e / 7;
            en[i] = e - 7 * kj; ek[i] = kj / 3; ej[i] = kj - 3 * ek[i];
            if (e >= 21 * K) ek[i] = -1;
            rk[i] = 0.0;
        }
#pragma unroll 1
        for (int c = wave; c < N; c += NW) {
            const int2 gr = sGrp[c];
            const int nst = (gr.y - gr.x + 3) >> 2;   // <= (2 * 16 + 4 + 3) / 4 = 9 steps
            d4 acc = {0, 0, 0, 0}, accx = {0, 0, 0, 0};
            const int colb = fa.cb0 + 6 * c;
            // keypoint rows (issued first: their LDS reads are in flight while the matrix instructions run)
            double yk[KE];
#pragma unroll
            for (int i = 0; i < KE; ++i) {
                yk[i] = 0.0;
                if (ek[i] >= 0) {
                    const int2 tb = sTab[c * KT + ek[i]];
                    const int la = tb.x >= 0 ? tb.x : 0, lb = tb.y >= 0 ? tb.y : 0;
                    const double qa = q[la * QS + ej[i]], qbv = q[lb * QS + ej[i]];
                    const double xa = xr[la * 7 + en[i]], xb2 = xr[lb * 7 + en[i]];
                    yk[i] = (tb.x >= 0 ? qa * xa : 0.0) + (tb.y >= 0 ? qbv * xb2 : 0.0);
                }
            }
            {   // all steps' operands (nst <= 9) are read before the first matrix instruction issues
                double bv[9], av[9];
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    const int lp0 = gr.x + 4 * u + kk;
                    const bool ok = u < nst && lp0 < gr.y;
                    const int lp = ok ? lp0 : gr.x;
                    const double xv = xr[lp * 7 + (cc < 7 ? cc : 0)];
                    const double qv = q[lp * QS + 3 + (cc < 9 ? cc : 0)];
                    bv[u] = (ok && cc < 7) ? xv : 0.0;
                    av[u] = (ok && cc < 9) ? qv : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    if (u < nst) {   // (wave-uniform)
                        acc = mfma_f64(av[u], bv[u], acc);
                        accx = mfma_f64(bv[u], bv[u], accx);
                    }
                }
            }
            // D[i = kk + 4 r][n = cc]: rows i < 9 of Y; n < 6 the clone's window columns, n == 6 the residual column
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = kk + 4 * r;
                if (cc < 6 && i < 9) Yo[(size_t)i * fa.NAP + colb + cc] = acc[r];
            }
            if (cc == 6) {
#pragma unroll
                for (int r = 0; r < 4; ++r) racc[r] += acc[r];
            }
#pragma unroll
            for (int i = 0; i < KE; ++i) {
                if (ek[i] >= 0) {
                    const int row = 9 + 3 * ek[i] + ej[i];
                    if (en[i] < 6) Yo[(size_t)row * fa.NAP + colb + en[i]] = yk[i];
                    else rk[i] += yk[i];
                }
            }
            // the clone's 7 x 7 tile [hx | r]^T [hx | r] (8 x 8 stored; the rest of the 16 x 16 product is zero)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = kk + 4 * r;
                if (cc < 8) So[(size_t)c * 64 + i * 8 + cc] = accx[r];
                if (i == 6 && cc == 6) rr += accx[r];
            }
        }
        // rows of Y behind the object's columns (9 + 3 K .. NOP): zero in the clone columns
        for (int idx = tid; idx < (NOP - (9 + 3 * K)) * 6 * N; idx += 64 * NW) {
            const int i = idx / (6 * N), z = idx - i * 6 * N;
            Yo[(size_t)(9 + 3 * K + i) * fa.NAP + fa.cb0 + z] = 0.0;
        }
        FUSED_STAMP(8);
        // Q~^T r and |r|^2: partial sums of the wavefronts through LDS (sPartR is free again: NW * 64 + NW doubles)
        for (int i = lane; i < 64; i += 64) sPartR[wave * 64 + i] = 0.0;
        if (cc == 6) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kk + 4 * r < 9) sPartR[wave * 64 + kk + 4 * r] = racc[r];
        }
#pragma unroll
        for (int i = 0; i < KE; ++i)
            if (ek[i] >= 0 && en[i] == 6) sPartR[wave * 64 + 9 + 3 * ek[i] + ej[i]] = rk[i];
        if (lane == 38) sMax[wave] = rr;   // (kk = 2, cc = 6: the lane that holds D[6][6] in r = 1)
        // zero columns: in front of the clones and the padding behind the residual column (an unseen clone's six columns are written
        // as zeros by the loop above: no rows, zero accumulators)
        {
            const int nz = fa.cb0 + (fa.NAP - fa.NA - 1);
            for (int idx = tid; idx < NOP * nz; idx += 64 * NW) {
                const int i = idx / nz, z = idx - i * nz;
                Yo[(size_t)i * fa.NAP + (z < fa.cb0 ? z : fa.NA + 1 + (z - fa.cb0))] = 0.0;
            }
        }
        __syncthreads();
        if (tid < NOP) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += sPartR[w * 64 + tid];   // fixed order: deterministic
            Yo[(size_t)tid * fa.NAP + fa.NA] = tid < ncol ? t : 0.0;
        }
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < NW; ++w) t += sMax[w];
            fa.rr[(size_t)o * fa.rr_stride] = t;
        }
    }
    FUSED_STAMP(6);
}

// A' = sum_o B_o - Y^T Y with B assembled on the fly from the clone tiles (obj_assemble_B_body's entries): one workgroup of eight
// wavefronts per 16 x 16 tile, split-K over the wavefronts (K = nobj * NOP rows of Y), partial tiles summed through LDS in wave order.
__global__ __launch_bounds__(512) void k_gemm_objA(const double* __restrict__ Y, int NAP, int Krows, const double* __restrict__ Sg, int nobj, int N,
                                                   const double* __restrict__ rr, int rr_stride, int cb0, int NA, double* __restrict__ dst,
                                                   unsigned* __restrict__ done) {
    // done: completion counter (one count per workgroup, behind its stores and an agent-scope release): what the object solve's first
    // launch polls instead of a stream-level join (k_gemm `wait`)
    __shared__ double sPart[7][4][64];
    __shared__ double sBsum[4][64];
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int ntj = NAP >> 4;
    const int bi = blockIdx.x / ntj, bj = blockIdx.x - bi * ntj;
    const int kk = l >> 4, cc = l & 15;
    // entry (i, j) of sum_o B_o (obj_assemble_B_body's): the clone tiles and the corner.  Wavefront 1 + r sums the entries of register r of
    // the output tile over the objects (index order: deterministic), all its loads in flight at once, BEFORE its slice of the product:
    // the latency hides under the matrix instructions (a chain of dependent batches behind the product cost 10 us here).
    if (wave >= 1 && wave <= 4) {
        const int r = wave - 1;
        const int i = 16 * bi + kk + 4 * r, j = 16 * bj + cc;
        const int ci = (i >= cb0 && i < cb0 + 6 * N) ? (i - cb0) / 6 : -1, cj = (j >= cb0 && j < cb0 + 6 * N) ? (j - cb0) / 6 : -1;
        const int ei = ci >= 0 ? i - cb0 - 6 * ci : (i == NA ? 6 : -1), ej = cj >= 0 ? j - cb0 - 6 * cj : (j == NA ? 6 : -1);
        const bool corner = ei == 6 && ej == 6;
        const bool tile = !corner && ei >= 0 && ej >= 0 && (ci == cj || ci < 0 || cj < 0);
        const double* base = corner ? rr : Sg + (size_t)(ci >= 0 ? ci : (cj >= 0 ? cj : 0)) * 64 + (tile ? ei * 8 + ej : 0);
        const size_t st = corner ? (size_t)rr_stride : (size_t)N * 64;
        double sB = 0.0;
        if (corner || tile) {
            for (int ob = 0; ob < nobj; ob += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const double t = base[(size_t)(ob + u < nobj ? ob + u : ob) * st];
                    v[u] = ob + u < nobj ? t : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) sB += v[u];
            }
        }
        sBsum[r][l] = sB;
    }
    const int KS = (((Krows + 7) >> 3) + 3) & ~3;   // k-slice per wavefront, a multiple of the MFMA depth
    const int k0 = wave * KS;
    const int Kw = (Krows - k0 < KS) ? (Krows - k0) : KS;
    // (a slice of 120 rows -- twenty cars -- is ONE batch of loads: three dependent round trips to L2 were most of this launch's 13 us)
    d4 acc = tile_product<32>(Y + (long)k0 * NAP, 1L, (long)NAP, Y + (long)k0 * NAP, (long)NAP, 1L, NAP, NAP, Kw, 16 * bi, 16 * bj, l);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sPart[wave - 1][r][l] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double v = acc[r];
#pragma unroll
        for (int w = 0; w < 7; ++w) v += sPart[w][r][l];
        const int i = 16 * bi + kk + 4 * r, j = 16 * bj + cc;
        dst[(size_t)i * NAP + j] = sBsum[r][l] - v;
    }
    (void)done;
}
// "the compression enqueued as number `value` is complete": one store, in a launch of its own BEHIND k_gemm_objA on the same stream -- the
// kernel boundary makes A' visible device-wide before the word changes.  (Counting the finished workgroups inside k_gemm_objA needed an
// agent-scope release -- an L2 write-back -- per workgroup: 6 of that launch's 13 us.)  The object solve's first product polls the word
// (k_gemm `wait`): no stream-level join.
__global__ __launch_bounds__(64) void k_obj_done(unsigned* __restrict__ done, unsigned value) {
    if (threadIdx.x == 0) __hip_atomic_store(done, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace orcvio_amd

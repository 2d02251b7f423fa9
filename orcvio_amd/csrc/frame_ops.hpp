// frame_ops.hpp -- the small steps of ONE filter frame on the resident covariance, folded into a few launches
// (orcvio_msckf_io_step_frame, capi_step.inc).  What the reference does to state_cov per image (OrcVIO::processFeatures,
// src/orcvio.cpp:567-594) around its two updates is byte and element work on an n x n matrix (n <= 406) and a few KB of tracks: as
// separate launches and copies each of these costs a launch slot (~4.5 us back to back on one stream), which is what bounds a frame
// of 20 clones x 20-200 short tracks -- not the arithmetic.  HBM-bound element kernels, coalesced, one thread per output element.
//   k_pose_step     the window poses of the frame's second update: the first update's, incremented by its dx
//                   (incrementState_IMUCam, src/orcvio.cpp:4468-4567, clone + extrinsic part) or copied; keeps the first
//                   update's status words for the second update's commit
//   k_frame_head    processModel's covariance propagation (:800-816) + stateAugmentation (:962-1010) in one pass, and the pull of
//                   the frame's inputs out of the pinned arena (tracks, derived index arrays, in-state feature records)
//   k_cov_remove_fac  marginalisation (:2935-2951) of the covariance AND of its resident square-root factor in one launch
#pragma once
#include <hip/hip_runtime.h>

#include "io_ops.hpp"

namespace orcvio_amd {

// Sophus v1.0.0 SO3d::exp (unit quaternion from the rotation vector, then the matrix): the arithmetic of the host's
// orcvio_msckf_increment_state (capi_state.inc)
__device__ __forceinline__ void dev_so3_exp(const double* w, double* R) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th = sqrt(th2);
    double imag, real;
    if (th < 1e-10) {
        const double th4 = th2 * th2;
        imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
        real = 1.0 - th2 / 8.0 + th4 / 384.0;
    } else {
        imag = sin(0.5 * th) / th;
        real = cos(0.5 * th);
    }
    const double x = imag * w[0], y = imag * w[1], z = imag * w[2], q = real;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * q); R[2] = 2 * (x * z + y * q);
    R[3] = 2 * (x * y + z * q); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * q);
    R[6] = 2 * (x * z - y * q); R[7] = 2 * (y * z + x * q); R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ void dev_mat3_mul(const double* A, const double* B, double* C) {
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) C[i] = T[i];
}

// One thread per clone: pose record [R_b2w 9 | t_b_w 3 | t_fej 3 | R_b2c 9 | t_c_b 3 | pad] of the second update's window.
// apply == 0: a copy.  apply != 0: incrementState_IMUCam's clone and extrinsic part with dx of the update that has just run
// (left: use_larvio || use_left_perturbation, :4498, :4543; the extrinsic rotation by smallAngleQuaternion, :4512-4516), unless
// discard_large_update discards dx (:4479-4494).  t_fej is the first estimate and stays.
// Thread 0 also copies the status words info[0..15] of the update that has just run into `keep` (the second update's commit
// refuses itself when the first was refused: EpilogueArgs.info_also).
struct PoseStepArgs { const double* src; double* dst; int N, stride; const double* dx; int leg, apply, left, discard_large; const int* info; int* keep; };
__device__ __forceinline__ void pose_step_body(const double* __restrict__ src, double* __restrict__ dst, int N, int stride, const double* __restrict__ dx,
                                               int leg, int apply, int left, int discard_large, const int* __restrict__ info, int* __restrict__ keep, const int c) {
    if (c < 16 && keep) keep[c] = info ? info[c] : 0;   // (info == nullptr: the frame had no first update)
    if (c >= N) return;
    double r[28];
    for (int i = 0; i < 28; ++i) r[i] = src[(size_t)c * stride + i];
    bool app = apply != 0;
    if (app && discard_large) {
        const double nv = sqrt(dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
        const double np = sqrt(dx[6] * dx[6] + dx[7] * dx[7] + dx[8] * dx[8]);
        if (nv > 1.0 || np > 1.5) app = false;
    }
    if (app) {
        double Rt[9];
        const double* da = dx + leg + 6 * c;
        const double w[3] = {da[0], da[1], da[2]};
        dev_so3_exp(w, Rt);
        if (left) dev_mat3_mul(Rt, r, r); else dev_mat3_mul(r, Rt, r);
        r[9] += da[3]; r[10] += da[4]; r[11] += da[5];
        // extrinsic: R_b2c <- R_b2c * R(smallAngleQuaternion(dtheta))^T (math_utils.hpp:104-121), t_c_b += dx[18:21]
        double q0 = 0.5 * dx[15], q1 = 0.5 * dx[16], q2 = 0.5 * dx[17], q3;
        const double n2 = q0 * q0 + q1 * q1 + q2 * q2;
        if (n2 <= 1.0) q3 = sqrt(1.0 - n2);
        else {
            const double s = 1.0 / sqrt(1.0 + n2);
            q0 *= s; q1 *= s; q2 *= s; q3 = s;
        }
        const double x = q0, y = q1, z = q2, ww = q3;
        const double RqT[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + z * ww), 2 * (x * z - y * ww),
                               2 * (x * y - z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z + x * ww),
                               2 * (x * z + y * ww), 2 * (y * z - x * ww), 1 - 2 * (x * x + y * y)};
        dev_mat3_mul(r + 15, RqT, r + 15);
        r[24] += dx[18]; r[25] += dx[19]; r[26] += dx[20];
    }
    for (int i = 0; i < 28; ++i) dst[(size_t)c * stride + i] = r[i];
}
__global__ __launch_bounds__(64) void k_pose_step(PoseStepArgs p) {
    pose_step_body(p.src, p.dst, p.N, p.stride, p.dx, p.leg, p.apply, p.left, p.discard_large, p.info, p.keep, threadIdx.x);
}


// ---- k_frame_head --------------------------------------------------------------------------------------------------------------
// Byte segments pulled out of the pinned arena (host-coherent, device-visible) into HBM: the tracks and derived index arrays of the
// frame's first update in one piece, the records of the in-state features in their nine arrays.  Every segment is cut into pieces of
// FH_PIECE bytes; piece p goes to ingest workgroup p mod nb_ing (PCIe reads want many requests in flight, and the small segments
// must not queue behind the large one).  16 bytes per lane where source, destination and length allow, else 4.
#define FH_SEGS 12
#define FH_PIECE 4096
struct FrameHeadArgs {
    // covariance: out (m x m) from P (n x n); Phi / Q: pinned, nullptr = no propagation; pose < 0 = no augmentation (m == n)
    const double* P; int n; double* out; int m; const double* Phi; const double* Q; int leg; int pose;
    int nb_cov;                 // workgroups of the covariance part (0: none)
    int nb_ing;                 // ingest workgroups behind them
    int nseg;
    const char* src[FH_SEGS]; char* dst[FH_SEGS]; unsigned bytes[FH_SEGS];
    int pose_block;             // ingest workgroup that also runs the pose step of the frame's second update (-1: none)
    PoseStepArgs ps;
};

// The arithmetic of k_cov_propagate_rows / k_cov_propagate_finish / k_cov_augment (cov_ops.hpp), element by element in the same order --
// the separate calls and this launch give the same bits (tests/test_gpu_stream.py):
//   P' = [[sym(Phi P_LL Phi^T + Q), Phi P_LC], [(Phi P_LC)^T, P_CC]]   (src/orcvio.cpp:800-816)
//   out = P' with the new clone's rows / columns copied from the IMU's (theta, p) and (X + X^T) / 2 (:962-1010)
// One thread per output element, enumerated so that the elements whose source lies in the IMU rows come first: L_out = the output
// indices whose source index is < leg (the IMU block and, when augmenting, the new clone), C_out the others.
//   region 1  (a, b) in L_out x L_out   needs T_LL = Phi P_LL (leg x leg, formed in LDS by every workgroup of the region: 2 products per thread)
//   region 2  (a, b) in L_out x C_out   T(sa, sb) = sum_l Phi(sa, l) P(l, sb), written to (a, b) and (b, a)
//   region 3  (a, b) in C_out x C_out   P(sa, sb), symmetrised as the augmentation does
// Only regions 1 and 2 (a fifth of the matrix) read Phi from the pinned arena.
template <int LEG>   // 22 or 46 (leg_dim): the products over the IMU block unroll completely -- their loads in flight together, the sums in the same order
__global__ __launch_bounds__(256) void k_frame_head(FrameHeadArgs a) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (b >= a.nb_cov) {
        const int w = b - a.nb_cov;
        if (w == a.pose_block && t < 64) pose_step_body(a.ps.src, a.ps.dst, a.ps.N, a.ps.stride, a.ps.dx, a.ps.leg, a.ps.apply, a.ps.left, a.ps.discard_large, a.ps.info, a.ps.keep, t);
        int p0 = 0;   // pieces of the segments before this one
        for (int sgi = 0; sgi < a.nseg; ++sgi) {
            const unsigned bytes = a.bytes[sgi];
            const int np = (int)((bytes + FH_PIECE - 1) / FH_PIECE);
            for (int q = 0; q < np; ++q) {
                if ((p0 + q) % a.nb_ing != w) continue;
                const unsigned off = (unsigned)q * FH_PIECE;
                const unsigned len = bytes - off < FH_PIECE ? bytes - off : FH_PIECE;
                const char* sp = a.src[sgi] + off;
                char* dp = a.dst[sgi] + off;
                if ((((size_t)sp | (size_t)dp | (size_t)len) & 15) == 0) {
                    const u32x4* s16 = reinterpret_cast<const u32x4*>(sp);
                    u32x4* d16 = reinterpret_cast<u32x4*>(dp);
                    for (unsigned i = t; i < len / 16; i += 256) d16[i] = __builtin_nontemporal_load(s16 + i);
                } else {
                    const unsigned* s4 = reinterpret_cast<const unsigned*>(sp);
                    unsigned* d4 = reinterpret_cast<unsigned*>(dp);
                    for (unsigned i = t; i < len / 4; i += 256) d4[i] = __builtin_nontemporal_load(s4 + i);
                }
            }
            p0 += np;
        }
        return;
    }
    __shared__ double sPhi[LEG * LEG];
    __shared__ double sT[LEG * LEG];
    const int n = a.n, m = a.m, pose = a.pose;
    constexpr int leg = LEG;
    const int nL = a.Phi ? leg + (pose >= 0 ? 6 : 0) : 0, nC = m - nL;
    const int E1 = nL * nL, E2 = nL * nC, E3 = nC * nC;
    const int e0 = b * 256;
    if (e0 < E1 + E2) {   // (this workgroup has elements of regions 1 / 2)
        for (int i = t; i < leg * leg; i += 256) sPhi[i] = a.Phi[i];
        __syncthreads();
        if (e0 < E1) {
            for (int idx = t; idx < leg * leg; idx += 256) {
                const int i = idx / leg, j = idx - i * leg;
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < leg; ++k) s += sPhi[i * leg + k] * a.P[(size_t)k * n + j];
                sT[idx] = s;
            }
            __syncthreads();
        }
    }
    const int e = e0 + t;
    if (e >= E1 + E2 + E3) return;
    // output indices (oa, ob)
    auto Lout = [&](int k) { return k < leg ? k : pose + (k - leg); };
    auto Cout = [&](int k) { int q = leg + k; if (pose >= 0 && q >= pose) q += 6; return q; };
    auto src = [&](int o, bool& isnew) {
        isnew = pose >= 0 && o >= pose && o < pose + 6;
        if (pose < 0) return o;
        if (isnew) { const int r = o - pose; return r < 3 ? r : r + 3; }
        return o < pose ? o : o - 6;
    };
    int oa, ob, region;
    if (e < E1) { region = 1; oa = Lout(e / nL); ob = Lout(e % nL); }
    else if (e < E1 + E2) { region = 2; const int q = e - E1; oa = Lout(q / nC); ob = Cout(q % nC); }
    else { region = 3; const int q = e - E1 - E2; if (nL == 0) { oa = q / m; ob = q - oa * m; } else { oa = Cout(q / nC); ob = Cout(q % nC); } }
    bool na, nb;
    const int sa = src(oa, na), sb = src(ob, nb);
    if (region == 1) {
        double x = a.Q[sa * leg + sb], y = a.Q[sb * leg + sa];
#pragma unroll
        for (int k = 0; k < leg; ++k) { x += sT[sa * leg + k] * sPhi[sb * leg + k]; y += sT[sb * leg + k] * sPhi[sa * leg + k]; }
        a.out[(size_t)oa * m + ob] = 0.5 * (x + y);
    } else if (region == 2) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < leg; ++k) s += sPhi[sa * leg + k] * a.P[(size_t)k * n + sb];
        a.out[(size_t)oa * m + ob] = s;
        a.out[(size_t)ob * m + oa] = s;
    } else {
        const double p1 = a.P[(size_t)sa * n + sb], p2 = a.P[(size_t)sb * n + sa];
        double v;
        if (pose < 0) v = p1;   // (no augmentation: the propagation copies P_CC as it stands)
        else if (na == nb) v = 0.5 * (p1 + p2);
        else if (na) v = p1;
        else v = p2;
        a.out[(size_t)oa * m + ob] = v;
    }
}

// ---- marginalisation of the covariance and of its resident factor in one launch (k_cov_remove + k_fac_remove, cov_ops.hpp) ----------
// The removed clones come by value (ascending window indices): new index -> old index by walking the list.  Workgroups [0, nb_P):
// out (mm x mm) = P without the removed rows / columns; the others: the factor's rows (k x mm, ld ldo) likewise (fac_k = 0: none).
struct RemoveArgs { int leg, count, clone[8]; };
__device__ __forceinline__ int remove_map(const RemoveArgs& r, int i) {
    int o = i;
    for (int q = 0; q < r.count; ++q)
        if (o >= r.leg + 6 * r.clone[q]) o += 6;
    return o;
}
__global__ __launch_bounds__(256) void k_cov_remove_fac(const double* __restrict__ P, int n, int mm, RemoveArgs r, double* __restrict__ out, int nb_P,
                                                        const double* __restrict__ F, int ld, int k, double* __restrict__ Fout, int ldo) {
    const int b = blockIdx.x;
    if (b < nb_P) {
        const int idx = b * 256 + threadIdx.x;
        if (idx >= mm * mm) return;
        const int i = idx / mm, j = idx - i * mm;
        out[idx] = P[(size_t)remove_map(r, i) * n + remove_map(r, j)];
        return;
    }
    const int idx = (b - nb_P) * 256 + threadIdx.x;
    if (idx >= k * mm) return;
    const int i = idx / mm, j = idx - i * mm;
    Fout[(size_t)i * ldo + j] = F[(size_t)i * ld + remove_map(r, j)];
}


// ---- one wavefront that waits for a word: the launches enqueued behind it on ITS stream start once another stream has reached the
// point the word stands for (k_front's first instruction stores it: the frame head in front of k_front is complete).  The kernel
// boundary behind this launch is the acquire; bounded like every wait on the device (the update is flagged and run again).
__global__ __launch_bounds__(64) void k_wait_word(const unsigned* __restrict__ word, unsigned expect, int spin_limit, int* __restrict__ lost) {
    if (threadIdx.x != 0) return;
    int it = 0;
    while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expect) < 0 && it < spin_limit) { __builtin_amdgcn_s_sleep(8); ++it; }
    if (it >= spin_limit) atomicAdd(lost, 1);
}

// ---- the rows of the in-state features: k_ekf_eval + (fill) + k_ekf_gate of ekf_rows.hpp in ONE launch ---------------------------------
// A wavefront per feature: lane 0 evaluates the four blocks (measurementJacobian_ekf_{3,1}didp), the wavefront clears the feature's two
// dense rows, gates the row pair against the prior (2 degrees of freedom, src/orcvio.cpp:2457) and writes them if accepted.  The
// bodies are the separate kernels': the same bits.  k_gram of the dense rows follows as before.
// bail: optional word (the update's lost-hand-off counter) -- behind a wait that gave up the records may not be there yet: nothing is touched.
__global__ __launch_bounds__(64) void k_ekf_evalgate(EkfEvalArgs e, int do_eval, EkfGateArgs g, const int* __restrict__ bail) {
    const int f = blockIdx.x, l = threadIdx.x;
    if (bail && __hip_atomic_load(bail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    for (int i = l; i < 2 * g.NAP; i += 64) g.E[(size_t)2 * f * g.NAP + i] = 0.0;
    if (do_eval && l == 0) ekf_eval_body(e, f);
    __syncthreads();   // (one wavefront: the stores of lane 0 and of the fill are out before the gate reads / writes)
    ekf_gate_body(g, f, l);
}

// ---- k_finish_pub: k_finish_sqrt + k_epilogue in ONE launch (a feature update's last two launches) -----------------------------------
// The tiles of P+ = s2 Zn^T Zn and dx = Zn^T z exactly as k_finish_sqrt forms them (one workgroup per lower tile, the same split of K
// over four wavefronts, the same order: the same bits).  What k_epilogue did behind it rides along:
//   commit     P+ is written straight into the SPARE covariance buffer -- the host swaps the two buffers instead of a copy over the
//              resident one -- and every workgroup writes its share of S+ = sigma Z^T (or of the prior's own factor) into the spare
//              factor buffer.  Refusals known at launch (the pivot counters of chol(M), a lost in-launch hand-off, the frame's first
//              update refused: info_also) are decided by every workgroup for itself: the spare buffers then receive the PRIOR and its
//              factor, so that the swap is right either way.  A non-finite dx -- which only shows once the tiles are done -- is caught
//              by the workgroup that arrives last: it rewrites both buffers with the prior (never on sane input).
//   publish    the workgroup that arrives last (a device counter behind an agent-scope fence) copies the small result block
//              [info | dx | gamma | accept] into host-coherent memory and raises the flag word the calling thread spins on.
// Not with P+ to the host (want_P): the tiles written one by one into the pinned block measured SLOWER than k_epilogue's forty copy
// workgroups behind k_finish_sqrt (13.3 against 5.2 + 6.3 us, profiles/r6a): that form keeps the two launches.
struct FinishPubArgs {
    const double* Z; int ldz, n, kdim; double s2;
    const double* P;        // the prior
    double* P_dst;          // commit: the spare covariance buffer; no commit: the outputs arena's P_out
    double* dx;
    int commit;             // 0: none (k_finish_sqrt's semantics), 1: P+ only, 2: P+ and the factor
    double sigma; const double* prior; long sLi, sLj; double* Sout; int ldo;
    const int* info; const int* info_also;
    const u32x4* small_src; u32x4* small_dst; size_t small16;
    int* counter; unsigned long long* seq; unsigned long long* flag;
};
__global__ __launch_bounds__(256) void k_finish_pub(FinishPubArgs a) {
    __shared__ double sPart[3][4][64];
    __shared__ int sLast, sBad;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, t = threadIdx.x;
    const int n = a.n, kdim = a.kdim, ldz = a.ldz;
    int bi, bj;
    tile_from_linear(blockIdx.x, bi, bj);
    const bool failed = a.info[2] != 0 || a.info[3] != 0;
    const bool also = a.info_also && (a.info_also[2] != 0 || a.info_also[3] != 0 || a.info_also[8] != 0);
    const bool refuse_up = failed || a.info[8] != 0 || also;
    // what the host reads besides dx -- info, gamma, accept: older launches' results -- leaves with the first workgroup, under the products
    const size_t dx0 = 256 / 4, dx1 = dx0 + (((size_t)n * 8 + 255) & ~(size_t)255) / 4;   // (words of the small block that hold dx: layout_outputs)
    if (blockIdx.x == 0) {
        const unsigned* s4 = reinterpret_cast<const unsigned*>(a.small_src);
        unsigned* d4p = reinterpret_cast<unsigned*>(a.small_dst);
        for (size_t i = t; i < a.small16 * 4; i += 256)
            if (i < dx0 || i >= dx1) d4p[i] = s4[i];
    }
    const int KS = ((kdim + 15) >> 4) << 2;
    const int k0 = wave * KS;
    const int Kw = (kdim - k0 < KS) ? (kdim - k0) : KS;
    d4 acc = tile_product(a.Z + (size_t)k0 * ldz, 1, ldz, a.Z + (size_t)k0 * ldz, ldz, 1, n + 1, n + 1, Kw, 16 * bi, 16 * bj, l);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sPart[wave - 1][r][l] = acc[r];
    }
    __syncthreads();
    const int kk = l >> 4, cc = l & 15;
    double* dx_host = reinterpret_cast<double*>(reinterpret_cast<char*>(a.small_dst) + 256);
    if (wave == 0) {
        const bool app = !failed;   // (dx as k_finish_sqrt decides it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = ((acc[r] + sPart[0][r][l]) + sPart[1][r][l]) + sPart[2][r][l];
            const int i = 16 * bi + kk + 4 * r, jj = 16 * bj + cc;
            double pv = 0.0, pm = 0.0;   // (i, jj) and its mirror
            if (i < n && jj < n && jj <= i) {
                if (a.commit) {
                    if (refuse_up) { pv = a.P[(size_t)i * n + jj]; pm = a.P[(size_t)jj * n + i]; }
                    else pv = pm = a.s2 * v;
                } else pv = pm = app ? a.s2 * v : 0.5 * (a.P[(size_t)i * n + jj] + a.P[(size_t)jj * n + i]);
                a.P_dst[(size_t)i * n + jj] = pv;
                a.P_dst[(size_t)jj * n + i] = pm;
            } else if (i == n && jj < n) {
                const double d = app ? v : 0.0;
                st_pub<true>(a.dx + jj, d);   // (read back by the workgroup that arrives last: past the caches)
                dx_host[jj] = d;
            }
        }
    }
    if (a.commit == 2) {   // this workgroup's share of the factor
        const size_t kn = (size_t)a.kdim * n, stride = (size_t)gridDim.x * 256;
        for (size_t idx = (size_t)blockIdx.x * 256 + t; idx < kn; idx += stride) {
            const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
            a.Sout[(size_t)i * a.ldo + j] = refuse_up ? a.prior[(long)j * a.sLi + (long)i * a.sLj] : a.sigma * a.Z[(size_t)i * ldz + j];
        }
    }
    // arrive: this workgroup's stores have been acknowledged (the host-coherent ones too) before it counts itself in
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        // (a workgroup that stored to host memory -- dx, the first workgroup's block -- makes those stores visible at system
        //  scope before it counts itself in, as k_epilogue's workgroups do: the flag must not overtake them on another path of the fabric)
        if (blockIdx.x == 0 || 16 * bi + 15 >= n) __threadfence_system();
        const int old = __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sLast = (old == (int)gridDim.x - 1) ? 1 : 0;
        sBad = 0;
    }
    __syncthreads();
    if (!sLast) return;
    int bad = 0;
    for (int i = t; i < n; i += 256) { const double v = ld_pub(a.dx + i); bad |= !(v - v == 0.0); }
    if (bad) sBad = 1;
    __syncthreads();
    if (a.commit && sBad && !refuse_up) {   // non-finite result: the spare buffers get the prior and its factor after all
        for (size_t i = t; i < (size_t)n * n; i += 256) a.P_dst[i] = a.P[i];
        if (a.commit == 2) {
            const size_t kn = (size_t)a.kdim * n;
            for (size_t idx = t; idx < kn; idx += 256) {
                const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
                a.Sout[(size_t)i * a.ldo + j] = a.prior[(long)j * a.sLi + (long)i * a.sLj];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (t == 0) {
        __threadfence_system();
        __hip_atomic_store(a.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long v = atomicAdd(a.seq, 1ull) + 1ull;
        __hip_atomic_store(a.flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}


// ---- the thin update: a stack of a few projected rows, applied in the reference's own direct form ------------------------------------
// pruneImuStateBuffer's update (src/orcvio.cpp:2803-2851) stacks ONE row per feature seen in both clones that leave: a handful of rows
// against ~150 states.  measurementUpdate_msckf skips its QR for such a stack (H.rows() <= H.cols(), :1664-1681) and applies
//     S = H P H^T + s2 I,  K = P H^T S^-1,  dx = K r,  P+ = (I - K H) P, symmetrised                       (:1684-1753)
// directly.  The square-root form of the general path (chol P, M = s2 I + L^T A L of dimension n, chol M, triangular solves: ~70 us
// of latency-bound launches) is the wrong tool for m <= 16 rows; here, from the materialised projected rows H' (k_feature):
//   k_thin_gain   ONE workgroup: W = P_a H'^T (n x m), S = H' W_a + s2 I (m x m) = L L^T, V = W L^-T (n x m), u = L^-1 r
//   k_thin_apply  P+ = sym(P) - V V^T, dx = V u, elementwise over the n x n matrix; the commit into the spare covariance buffer and
//                 the publication ride along exactly as in k_finish_pub (the last workgroup raises the flag)
// S is s2 I plus a positive semi-definite matrix: well conditioned whatever the prior's rank; a non-positive or non-finite pivot
// (non-finite input) refuses the update (info[2]: P+ = P, dx = 0), as the general path's chol(M) does.
#define THIN_MAX_ROWS 16
#define THIN_CHUNK 32
__host__ __device__ inline size_t thin_gain_lds_doubles(int m, int NA, int n) { return (size_t)m * (NA + 1) + (size_t)n * m + (size_t)m * m + (size_t)n * (THIN_CHUNK + 1); }
struct ThinGainArgs { const double* Hs; int m, NAP, NA, n; const double* P; double s2; double* V; double* u; int* info; };
__global__ __launch_bounds__(1024) void k_thin_gain(ThinGainArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sThin[];
    const int m = a.m, n = a.n, NA = a.NA, t = threadIdx.x;
    double* sH = sThin;                       // [m][NA + 1]  (the residual in column NA)
    double* sW = sH + (size_t)m * (NA + 1);   // [n][m]
    double* sS = sW + (size_t)n * m;          // [m][m] -> its lower Cholesky factor
    double* sP = sS + (size_t)m * m;          // [n][THIN_CHUNK + 1]  the entries of P of the current chunk of columns
    __shared__ int sFail;
    // the columns ANY row touches, ascending: a projected row has non-zeros in the extrinsic columns and the clones of its track, and
    // pruneImuStateBuffer's tracks all sit on the two clones that leave -- ~20 of NA columns.  The products below run over this list
    // only (the same terms in the same ascending order as the dense loops, the zero terms left out).
    __shared__ unsigned char sUse[512];
    __shared__ short sU[512];
    __shared__ int sNU;
    for (int idx = t; idx < m * (NA + 1); idx += 1024) { const int k = idx / (NA + 1), c = idx - k * (NA + 1); sH[idx] = a.Hs[(size_t)k * a.NAP + c]; }
    if (t == 0) sFail = 0;
    __syncthreads();
    if (t < NA) {
        bool nz = false;
        for (int k = 0; k < m; ++k) nz = nz || sH[(size_t)k * (NA + 1) + t] != 0.0;
        sUse[t] = nz ? 1 : 0;
    }
    __syncthreads();
    if (t < 64) {   // ascending compaction by one wavefront: ballot + prefix count, 64 columns per step
        int cnt = 0;
        for (int base = 0; base < NA; base += 64) {
            const int c = base + t;
            const bool use = c < NA && sUse[c] != 0;
            const unsigned long long mask = __ballot(use);
            if (use) sU[cnt + __popcll(mask & ((1ull << t) - 1ull))] = (short)c;
            cnt += __popcll(mask);
        }
        if (t == 0) sNU = cnt;
    }
    __syncthreads();
    const int nU = sNU;
    // W = P(:, active) H'^T, row t by thread t.  The entries of P it needs are brought into LDS by the WHOLE workgroup first, 32 columns
    // of the list at a time: n x 32 independent loads in flight together = one round trip to memory per chunk (a thread walking its own
    // row eight loads at a time paid four round trips of ~2 us each: P was written by the launch in front, on other compute units).
    {
        double acc[THIN_MAX_ROWS];
#pragma unroll
        for (int k = 0; k < THIN_MAX_ROWS; ++k) acc[k] = 0.0;
        for (int q0 = 0; q0 < nU; q0 += THIN_CHUNK) {
            const int nc = nU - q0 < THIN_CHUNK ? nU - q0 : THIN_CHUNK;
            if (q0 > 0) __syncthreads();
            for (int idx = t; idx < n * nc; idx += 1024) {
                const int i = idx / nc, q = idx - i * nc;
                sP[i * (THIN_CHUNK + 1) + q] = a.P[(size_t)i * n + 15 + sU[q0 + q]];
            }
            __syncthreads();
            if (t < n) {
                for (int q = 0; q < nc; ++q) {
                    const int c = sU[q0 + q];
                    const double pv = sP[t * (THIN_CHUNK + 1) + q];
#pragma unroll
                    for (int k = 0; k < THIN_MAX_ROWS; ++k) if (k < m) acc[k] += pv * sH[(size_t)k * (NA + 1) + c];
                }
            }
        }
        if (t < n) {
#pragma unroll
            for (int k = 0; k < THIN_MAX_ROWS; ++k) if (k < m) sW[(size_t)t * m + k] = acc[k];
        }
    }
    __syncthreads();
    for (int idx = t; idx < m * m; idx += 1024) {   // S = H' W(active rows) + s2 I
        const int k = idx / m, l = idx - k * m;
        const double* Hk = sH + (size_t)k * (NA + 1);
        double s = (k == l) ? a.s2 : 0.0;
        for (int q = 0; q < nU; ++q) { const int c = sU[q]; s += Hk[c] * sW[(size_t)(15 + c) * m + l]; }
        sS[idx] = s;
    }
    __syncthreads();
    // Cholesky of the m x m matrix, in place (lower), right-looking, the whole workgroup: per pivot the column scaled by m threads,
    // the trailing block updated by m x m (thread 0 alone -- FP64 sqrt and divide in one dependent chain -- took 4.5 us at m = 5)
    for (int k = 0; k < m; ++k) {
        const double d = sS[k * m + k];
        const bool badp = !(d > 0.0) || !(d - d == 0.0);
        const double lk = sqrt(badp ? 1.0 : d);
        double lik = 0.0;
        if (t > k && t < m) lik = sS[t * m + k] / lk;
        __syncthreads();   // (every thread has read the pivot and its column entry)
        if (t == k) { sS[k * m + k] = lk; if (badp) sFail = 1; }
        if (t > k && t < m) sS[t * m + k] = lik;
        __syncthreads();
        if (t < m * m) {
            const int i = t / m, j = t - i * m;
            if (i > k && j > k && j <= i) sS[i * m + j] -= sS[i * m + k] * sS[j * m + k];
        }
        __syncthreads();
    }
    if (t == 0) {
        a.info[2] = sFail; a.info[3] = 0;
        a.info[0] = 0; a.info[1] = 0;   // (no factorisation of the prior on this path)
    }
    if (t < n) {   // row t of V = W L^-T: v L^T = w, forward over the columns, in place in LDS (no private arrays: they would live in scratch)
        double* w = sW + (size_t)t * m;
        for (int k = 0; k < m; ++k) {
            double x = w[k];
            for (int q = 0; q < k; ++q) x -= w[q] * sS[k * m + q];
            x /= sS[k * m + k];
            w[k] = x;
            a.V[(size_t)t * THIN_MAX_ROWS + k] = x;
        }
    } else if (t == n) {   // u = L^-1 r, in place in the residual column of sH
        for (int k = 0; k < m; ++k) {
            double x = sH[(size_t)k * (NA + 1) + NA];
            for (int q = 0; q < k; ++q) x -= sS[k * m + q] * sH[(size_t)q * (NA + 1) + NA];
            x /= sS[k * m + k];
            sH[(size_t)k * (NA + 1) + NA] = x;
            a.u[k] = x;
        }
    }
}

struct ThinApplyArgs {
    const double* V; const double* u; int m, n;
    const double* P; double* P_dst; double* dx; int commit;
    const int* info; const int* info_also;
    const u32x4* small_src; u32x4* small_dst; size_t small16;
    int* counter; unsigned long long* seq; unsigned long long* flag;
};
__global__ __launch_bounds__(256) void k_thin_apply(ThinApplyArgs a) {
    __shared__ int sLast, sBad;
    const int t = threadIdx.x, n = a.n, m = a.m;
    const bool failed = a.info[2] != 0 || a.info[3] != 0;
    const bool also = a.info_also && (a.info_also[2] != 0 || a.info_also[3] != 0 || a.info_also[8] != 0);
    const bool refuse_up = failed || also;
    const size_t dx0 = 256 / 4, dx1 = dx0 + (((size_t)n * 8 + 255) & ~(size_t)255) / 4;
    if (blockIdx.x == 0) {   // what the host reads besides dx: info, gamma, accept (written by the launches in front)
        const unsigned* s4 = reinterpret_cast<const unsigned*>(a.small_src);
        unsigned* d4p = reinterpret_cast<unsigned*>(a.small_dst);
        for (size_t i = t; i < a.small16 * 4; i += 256)
            if (i < dx0 || i >= dx1) d4p[i] = s4[i];
    }
    double* dx_host = reinterpret_cast<double*>(reinterpret_cast<char*>(a.small_dst) + 256);
    const int idx = blockIdx.x * 256 + t;
    bool wrote_dx = false;
    if (idx < n * n) {
        const int i = idx / n, j = idx - i * n;
        double pv;
        if (a.commit && refuse_up) pv = a.P[idx];   // (refused: the spare buffer receives the prior as it stands)
        else {
            pv = 0.5 * (a.P[idx] + a.P[(size_t)j * n + i]);
            if (!failed) {
                double s = 0.0;
                for (int k = 0; k < m; ++k) s += a.V[(size_t)i * THIN_MAX_ROWS + k] * a.V[(size_t)j * THIN_MAX_ROWS + k];
                pv -= s;
            }
        }
        a.P_dst[idx] = pv;
    } else if (idx < n * n + n) {
        const int i = idx - n * n;
        double s = 0.0;
        if (!failed)
            for (int k = 0; k < m; ++k) s += a.V[(size_t)i * THIN_MAX_ROWS + k] * a.u[k];
        st_pub<true>(a.dx + i, s);
        dx_host[i] = s;
        wrote_dx = true;
    }
    const int any_dx = __syncthreads_or(wrote_dx ? 1 : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        if (blockIdx.x == 0 || any_dx) __threadfence_system();   // (host-coherent stores out before this workgroup counts itself in)
        const int old = __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sLast = (old == (int)gridDim.x - 1) ? 1 : 0;
        sBad = 0;
    }
    __syncthreads();
    if (!sLast) return;
    int bad = 0;
    for (int i = t; i < n; i += 256) { const double v = ld_pub(a.dx + i); bad |= !(v - v == 0.0); }
    if (bad) sBad = 1;
    __syncthreads();
    if (a.commit && sBad && !refuse_up) {   // non-finite result: the spare buffer gets the prior after all
        for (size_t i = t; i < (size_t)n * n; i += 256) a.P_dst[i] = a.P[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (t == 0) {
        __threadfence_system();
        __hip_atomic_store(a.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long v = atomicAdd(a.seq, 1ull) + 1ull;
        __hip_atomic_store(a.flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace orcvio_amd

// object_rows.hpp -- evaluation of the object residual rows at a fixed state:
// keypoint reprojection and bounding-box/quadric residuals with their Jacobians w.r.t. the camera pose
// (CameraLM, reference src/obj/ObjectResJacCam.cpp:153-519) and w.r.t. the object state (ObjectLM,
// src/obj/ObjectLM.cpp:250-632), followed by the re-indexing into the sliding window of
// OrcVIO::constructObjectResidualJacobians (src/orcvio.cpp:2017-2151): Hx6 = J_cam * D with
// D = get_cam_wrt_imu_se3_jacobian (include/orcvio/utils/se3_ops.hpp:531-552), rows interleaved per frame
// [keypoint rows ; 4 bbox rows].  One wavefront per in-window frame: lane i < K is keypoint i, lanes
// K..K+3 are the four bbox lines.  The *new* bbox residual is restated literally (its Jacobian uses the
// world-frame plane, SURVEY.md note N8) with new_bbox == 1, and with corrected Jacobians with new_bbox == 2.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace orcvio_amd {

struct ObjEvalArgs {
    const double* wTo;        // [16]
    const double* shape;      // [3]
    const double* kps;        // [K][3]
    const double* frame_wTc;  // [F][16]
    const double* frame_zs;   // [F][K][2]
    const double* frame_bbox; // [F][4]
    const int* frame_clone;   // [F] (-1: not in the window)
    const int* frame_row0;    // [F] first output row of the frame (only for in-window frames)
    int K, F, ncol;           // ncol = 9 + 3K
    int ldhf;                 // leading dimension of Hf (>= ncol; the columns beyond ncol are written as 0)
    int rcol;                 // >= ncol: the residual is ALSO written to column rcol of the Hf row (compact [Hf | r] rows of the
                              // object compression, msckf_kernels.hpp k_obj_cross); -1: not
    int* row_cols;            // optional: ncol of the object is written per row
    int obj_left, new_bbox, vio_left, fix_D;
    double R_b2c[9], t_c_b[3];
    int* row_clone;
    double* Hx6;
    double* Hf;
    double* res;
};

__device__ __forceinline__ void skew3d(const double* w, double* S) {
    S[0] = 0; S[1] = -w[2]; S[2] = w[1]; S[3] = w[2]; S[4] = 0; S[5] = -w[0]; S[6] = -w[1]; S[7] = w[0]; S[8] = 0;
}
// out(1x6) = w(1x4) * circledCirc(x)^T  = [ w4 * x123 , w123^T skew(x123) ]   (se3_ops.hpp:229-240)
__device__ __forceinline__ void row_times_ccT(const double* w, const double* x, double* out) {
    out[0] = w[3] * x[0]; out[1] = w[3] * x[1]; out[2] = w[3] * x[2];
    out[3] = w[1] * x[2] - w[2] * x[1];
    out[4] = w[2] * x[0] - w[0] * x[2];
    out[5] = w[0] * x[1] - w[1] * x[0];
}
// y = T(4x4 rigid: R 3x3 row-major in rows of a 16-array, t) * x(4)
__device__ __forceinline__ void mat4_vec(const double* T, const double* x, double* y) {
    for (int i = 0; i < 4; ++i) y[i] = T[i * 4] * x[0] + T[i * 4 + 1] * x[1] + T[i * 4 + 2] * x[2] + T[i * 4 + 3] * x[3];
}

// The rows of ONE in-window frame, one lane per row pair: lane role t < K is keypoint t (two rows if it was detected), roles
// K..K+3 are the four bbox lines; `lpf` lanes of the wavefront (a power of two >= K + 4, or 64) belong to one frame, so that a
// wavefront evaluates 64 / lpf frames side by side (k_obj_fused) or one (lpf = 64: k_object_rows).  f < 0: an idle lane group.
// Every lane of the wavefront must call it (it holds a ballot).  emit(row_in_frame, r, hx6, hpose, hshape | nullptr, kp | -1,
// hkp | nullptr, nrows_of_frame) is called once per row by the lane that owns it: hx6 = J_cam D (the row's six window columns),
// hpose / hshape / hkp the structural non-zeros of its H_f row.
template <class Emit>
__device__ __forceinline__ void object_rows_lane(const ObjEvalArgs& p, const int f, const int t, const int lpf, Emit&& emit) {
    const int clone = f >= 0 ? p.frame_clone[f] : -1;
    const bool live = clone >= 0;
    const int K = p.K;
    const int fq = live ? f : 0;
    // frame transforms
    double wTc[16], cTw[16], wTo[16];
    for (int i = 0; i < 16; ++i) { wTc[i] = p.frame_wTc[(size_t)fq * 16 + i]; wTo[i] = p.wTo[i]; }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) cTw[i * 4 + j] = wTc[j * 4 + i];
        cTw[i * 4 + 3] = -(wTc[0 * 4 + i] * wTc[3] + wTc[1 * 4 + i] * wTc[7] + wTc[2 * 4 + i] * wTc[11]);
    }
    cTw[12] = cTw[13] = cTw[14] = 0.0; cTw[15] = 1.0;
    // D = d(camera se3) / d(IMU [theta, p]) (get_cam_wrt_imu_se3_jacobian, se3_ops.hpp:531-552) in its 3 x 3 blocks
    //     D = [ D00  D01 ]      right perturbation: D00 = -R_b2c skew(t_c_b), D01 = R_w2c, D10 = R_b2c
    //         [ D10   0  ]      left perturbation:  D00 = skew(t_b_w),        D01 = I,     D10 = I
    // Only D00 is kept per lane (D10 is uniform or the identity, D01 is the rotation of cTw or the identity): the 36-entry matrix,
    // more than half of it zeros, cost 72 registers that the one-launch compression (k_obj_fused: 256 per lane) did not have.
    double D00[9];
    if (p.fix_D) {
        for (int i = 0; i < 9; ++i) D00[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else if (p.vio_left) {
        double v[3], tbw[3];
        for (int i = 0; i < 3; ++i) v[i] = -(p.R_b2c[i * 3] * p.t_c_b[0] + p.R_b2c[i * 3 + 1] * p.t_c_b[1] + p.R_b2c[i * 3 + 2] * p.t_c_b[2]);
        for (int i = 0; i < 3; ++i) tbw[i] = wTc[i * 4] * v[0] + wTc[i * 4 + 1] * v[1] + wTc[i * 4 + 2] * v[2] + wTc[i * 4 + 3];
        skew3d(tbw, D00);
    } else {
        double S[9];
        skew3d(p.t_c_b, S);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += p.R_b2c[i * 3 + k] * S[k * 3 + j];
                D00[i * 3 + j] = -s;
            }
    }
    // valid keypoints of this frame and their rank (row position)
    bool valid = false;
    double z0 = 0, z1 = 0;
    if (live && t < K) {
        z0 = p.frame_zs[((size_t)f * K + t) * 2];
        z1 = p.frame_zs[((size_t)f * K + t) * 2 + 1];
        valid = isfinite(z0) && isfinite(z1);
    }
    const unsigned long long mask_all = __ballot(valid);
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned long long mask = lpf >= 64 ? mask_all : (mask_all >> ((lane / lpf) * lpf)) & ((1ull << lpf) - 1ull);
    const int nvalid = __popcll(mask);
    const int rank = __popcll(mask & ((1ull << t) - 1ull));
    const int nrows = 2 * nvalid + 4;
    auto out = [&](int row_in_frame, double r, const double* jc, const double* hpose, const double* hshape, int kpid, const double* hkp) {
        double hx6[6];   // = jc D, block by block (the terms that multiply D's zero block are left out: they add exact zeros)
        for (int c = 0; c < 3; ++c) {
            const double a = (jc[0] * D00[c] + jc[1] * D00[3 + c]) + jc[2] * D00[6 + c];
            if (p.fix_D) { hx6[c] = a; hx6[3 + c] = jc[3 + c]; }
            else if (p.vio_left) { hx6[c] = a + jc[3 + c]; hx6[3 + c] = jc[c]; }
            else {
                hx6[c] = a + ((jc[3] * p.R_b2c[c] + jc[4] * p.R_b2c[3 + c]) + jc[5] * p.R_b2c[6 + c]);
                hx6[3 + c] = (jc[0] * cTw[c] + jc[1] * cTw[4 + c]) + jc[2] * cTw[8 + c];   // D01 = R_w2c
            }
        }
        emit(row_in_frame, r, hx6, hpose, hshape, kpid, hkp, nrows);
    };

    if (valid) {
        // ---- keypoint rows (ObjectResJacCam.cpp:153-282, ObjectLM.cpp:272-346) --------------------------
        double Xo[4] = {p.kps[3 * t], p.kps[3 * t + 1], p.kps[3 * t + 2], 1.0}, Xw[4], Xc[4];
        mat4_vec(wTo, Xo, Xw);
        mat4_vec(cTw, Xw, Xc);
        const double iz = 1.0 / Xc[2], zsq = Xc[2] * Xc[2];
        const double dpi[6] = {iz, 0, -Xc[0] / zsq, 0, iz, -Xc[1] / zsq};
        // A = [R | -R skew(x)] (3x6) with (R, x) = (R_cw, Xw) left / (I, Xc) right for the camera; pose of the object:
        // left [R_cw | -R_cw skew(Xw)], right R_co [I | -skew(Xo)]
        double Rco[9], S[9], M1[18], M2[18];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rco[i * 3 + j] = cTw[i * 4] * wTo[j] + cTw[i * 4 + 1] * wTo[4 + j] + cTw[i * 4 + 2] * wTo[8 + j];
        if (p.obj_left) {
            skew3d(Xw, S);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    M1[i * 6 + j] = cTw[i * 4 + j];
                    M1[i * 6 + 3 + j] = -(cTw[i * 4] * S[j] + cTw[i * 4 + 1] * S[3 + j] + cTw[i * 4 + 2] * S[6 + j]);
                }
            for (int i = 0; i < 18; ++i) M2[i] = M1[i];   // object pose (left): dpi * P * odot(wTo X)  -- same matrix
        } else {
            skew3d(Xc, S);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) { M1[i * 6 + j] = (i == j) ? 1.0 : 0.0; M1[i * 6 + 3 + j] = -S[i * 3 + j]; }
            skew3d(Xo, S);
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    M2[i * 6 + j] = Rco[i * 3 + j];
                    M2[i * 6 + 3 + j] = -(Rco[i * 3] * S[j] + Rco[i * 3 + 1] * S[3 + j] + Rco[i * 3 + 2] * S[6 + j]);
                }
        }
        for (int s = 0; s < 2; ++s) {
            double jc[6], hp[6], hk[3];
            for (int c = 0; c < 6; ++c) {
                const double a = dpi[s * 3] * M1[c] + dpi[s * 3 + 1] * M1[6 + c] + dpi[s * 3 + 2] * M1[12 + c];
                jc[c] = -a;
                hp[c] = dpi[s * 3] * M2[c] + dpi[s * 3 + 1] * M2[6 + c] + dpi[s * 3 + 2] * M2[12 + c];
            }
            for (int c = 0; c < 3; ++c) hk[c] = dpi[s * 3] * Rco[c] + dpi[s * 3 + 1] * Rco[3 + c] + dpi[s * 3 + 2] * Rco[6 + c];
            const double r = (s == 0 ? Xc[0] * iz - z0 : Xc[1] * iz - z1);
            out(2 * rank + s, r, jc, hp, nullptr, t, hk);
        }
    }
    if (live && t >= K && t < K + 4) {
        // ---- bbox rows (ObjectResJacCam.cpp:308-494, ObjectLM.cpp:441-616) --------------------------------
        const int j = t - K;
        const double* bb = p.frame_bbox + (size_t)f * 4;
        const double px[4] = {bb[0], bb[2], bb[2], bb[0]}, py[4] = {bb[1], bb[1], bb[3], bb[3]};
        const int j1 = (j + 1) & 3;
        const double ln[3] = {py[j] - py[j1], px[j1] - px[j], px[j] * py[j1] - py[j] * px[j1]};   // cross((x,y,1),(x',y',1))
        const double v2[3] = {p.shape[0] * p.shape[0], p.shape[1] * p.shape[1], p.shape[2] * p.shape[2]};
        double yyw[4], yyo[4];
        for (int c = 0; c < 4; ++c) yyw[c] = ln[0] * cTw[c] + ln[1] * cTw[4 + c] + ln[2] * cTw[8 + c];
        for (int c = 0; c < 4; ++c) yyo[c] = yyw[0] * wTo[c] + yyw[1] * wTo[4 + c] + yyw[2] * wTo[8 + c] + yyw[3] * wTo[12 + c];
        const double lprime[4] = {ln[0], ln[1], ln[2], 0.0};
        double r, jc[6], hp[6], hs[3];
        if (!p.new_bbox) {
            r = v2[0] * yyo[0] * yyo[0] + v2[1] * yyo[1] * yyo[1] + v2[2] * yyo[2] * yyo[2] - yyo[3] * yyo[3];
            const double u[4] = {2 * yyo[0] * v2[0], 2 * yyo[1] * v2[1], 2 * yyo[2] * v2[2], -2 * yyo[3]};   // 2 yyo Qi
            double w[4];
            mat4_vec(wTo, u, w);               // (u wTo^T)^T
            if (p.obj_left) {
                row_times_ccT(w, yyw, hp);
                for (int c = 0; c < 6; ++c) jc[c] = -hp[c];
            } else {
                double wc[4];
                mat4_vec(cTw, w, wc);          // (u wTo^T cTw^T)^T
                row_times_ccT(wc, lprime, jc);
                for (int c = 0; c < 6; ++c) jc[c] = -jc[c];
                row_times_ccT(u, yyo, hp);
            }
            for (int c = 0; c < 3; ++c) hs[c] = 2 * p.shape[c] * yyo[c] * yyo[c];
        } else {
            {   // residual: plane in the OBJECT frame
                const double sq = sqrt(v2[0] * yyo[0] * yyo[0] + v2[1] * yyo[1] * yyo[1] + v2[2] * yyo[2] * yyo[2]);
                const double bn = sqrt(yyo[0] * yyo[0] + yyo[1] * yyo[1] + yyo[2] * yyo[2]);
                r = (yyo[3] - (yyo[3] > 0 ? 1.0 : -1.0) * sq) / bn;
            }
            // Jacobians.  new_bbox == 1: plane in the WORLD frame, as the reference computes them (note N8: inconsistent with
            // its own residual, restated literally).  new_bbox == 2 (opt-in, "corrected"): plane in the OBJECT frame and the
            // shape derivative with its -sign(b4); these agree with central differences of the residual to 1e-9
            const bool corrected = p.new_bbox == 2;
            const double* ub = corrected ? yyo : yyw;
            const double sq = sqrt(v2[0] * ub[0] * ub[0] + v2[1] * ub[1] * ub[1] + v2[2] * ub[2] * ub[2]);
            const double bn = sqrt(ub[0] * ub[0] + ub[1] * ub[1] + ub[2] * ub[2]);
            const double sg = ub[3] > 0 ? 1.0 : -1.0;
            const double pa[4] = {-sg * v2[0] * ub[0] / sq, -sg * v2[1] * ub[1] / sq, -sg * v2[2] * ub[2] / sq, 1.0};
            const double pu = pa[0] * ub[0] + pa[1] * ub[1] + pa[2] * ub[2] + pa[3] * ub[3];
            double g[4];
            for (int c = 0; c < 4; ++c) g[c] = pa[c] / bn - ((c < 3) ? pu * ub[c] / (bn * bn * bn) : 0.0);
            double w[4];
            mat4_vec(wTo, g, w);
            if (p.obj_left) {
                row_times_ccT(w, yyw, hp);
                for (int c = 0; c < 6; ++c) jc[c] = -hp[c];
            } else {
                double wc[4];
                mat4_vec(cTw, w, wc);
                row_times_ccT(wc, lprime, jc);
                for (int c = 0; c < 6; ++c) jc[c] = -jc[c];
                row_times_ccT(g, yyo, hp);
            }
            for (int c = 0; c < 3; ++c) hs[c] = (corrected ? -sg : 1.0) * p.shape[c] * ub[c] * ub[c] / (bn * sq);
        }
        out(2 * nvalid + j, r, jc, hp, hs, -1, nullptr);
    }
}

// One wavefront per in-window frame, rows to the compact row arrays in device memory (k_object_rows / k_object_rows_batch).
__device__ __forceinline__ void object_rows_body(const ObjEvalArgs& p, const int f, const int t) {
    const int clone = p.frame_clone[f];
    if (clone < 0) return;   // (wave-uniform)
    const int row0 = p.frame_row0[f];
    const int ncol = p.ncol;
    // The rows of one frame are consecutive (keypoint rows, then the four bbox rows) and mostly zero: a row written by its lane
    // alone is ldhf scalar stores 8 * ldhf bytes apart from the next lane's (13.6 us for the 20 x 30 frames of config 3).  Up to
    // 28 rows x 48 columns (12 keypoints) the block is put together in LDS -- zeroed by the wavefront, the ten structural
    // non-zeros of a row written by its lane -- and streamed out with full-width stores.
    constexpr int LDS_ROWS = 28, LDS_LD = 48;
    __shared__ double sHf[LDS_ROWS * LDS_LD];
    const bool small = p.ldhf <= LDS_LD && 2 * p.K + 4 <= LDS_ROWS;   // (wave-uniform; the frame's own row count is at most 2 K + 4)
    if (small)
        for (int i = t; i < (2 * p.K + 4) * p.ldhf; i += 64) sHf[i] = 0.0;   // (the LDS executes one wavefront's operations in order)
    int nrows_frame = 0;
    object_rows_lane(p, f, t, 64, [&](int rif, double r, const double* hx6, const double* hpose, const double* hshape, int kpid, const double* hkp, int nrows) {
        const int row = row0 + rif;
        nrows_frame = nrows;
        p.res[row] = r;
        p.row_clone[row] = clone;
        for (int c = 0; c < 6; ++c) p.Hx6[(size_t)row * 6 + c] = hx6[c];
        double* hf = small ? sHf + (size_t)rif * p.ldhf : p.Hf + (size_t)row * p.ldhf;
        if (!small)
            for (int c = 0; c < p.ldhf; ++c) hf[c] = 0.0;
        if (p.rcol >= 0) hf[p.rcol] = r;
        if (p.row_cols) p.row_cols[row] = ncol;
        for (int c = 0; c < 6; ++c) hf[c] = hpose[c];
        if (hshape) for (int c = 0; c < 3; ++c) hf[6 + c] = hshape[c];
        if (hkp) for (int c = 0; c < 3; ++c) hf[9 + 3 * kpid + c] = hkp[c];
    });
    if (small) {
        // the frame's row count: the bbox lanes know it (they always emit); lane K broadcasts
        const int nrows = __shfl(nrows_frame, p.K);
        double* dst = p.Hf + (size_t)row0 * p.ldhf;
        for (int i = t; i < nrows * p.ldhf; i += 64) dst[i] = sHf[i];
    }
}

// one object per launch (grid = its frames) ...
__global__ __launch_bounds__(64) void k_object_rows(ObjEvalArgs p) { object_rows_body(p, blockIdx.x, threadIdx.x); }
// ... or every object of an update in one launch: grid (most frames of any object, objects), the arguments of object
// blockIdx.y from a device array (wave-uniform: scalar loads)
// The blocks behind the objects (blockIdx.y >= nobj) zero the scratch the compression accumulates into (`zero`, nzero doubles)
// the two pivot counters and the shard status words -- no fill launches in front of k_obj_front or of the solve.
__global__ __launch_bounds__(64) void k_object_rows_batch(const ObjEvalArgs* __restrict__ args, int nobj, double* __restrict__ zero,
                                                          size_t nzero, int* __restrict__ counters) {
    if ((int)blockIdx.y >= nobj) {
        const size_t nblk = (size_t)gridDim.x * (gridDim.y - nobj), b = (size_t)(blockIdx.y - nobj) * gridDim.x + blockIdx.x;
        double2* z2 = reinterpret_cast<double2*>(zero);   // (the scratch is 16-byte aligned and nzero is even)
        for (size_t i = b * 64 + threadIdx.x; i < nzero / 2; i += nblk * 64) z2[i] = double2{0.0, 0.0};
        if (b == 0 && (threadIdx.x < 2 || (threadIdx.x >= 5 && threadIdx.x < 9))) counters[threadIdx.x] = 0;   // info[4,5]: pivots; info[9..12]: shard status words
        return;
    }
    const ObjEvalArgs p = args[blockIdx.y];
    if ((int)blockIdx.x >= p.F) return;
    object_rows_body(p, blockIdx.x, threadIdx.x);
}

}  // namespace orcvio_amd

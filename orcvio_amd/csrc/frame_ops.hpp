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
__global__ __launch_bounds__(64) void k_pose_step(const double* __restrict__ src, double* __restrict__ dst, int N, int stride, const double* __restrict__ dx,
                                                  int leg, int apply, int left, int discard_large, const int* __restrict__ info, int* __restrict__ keep) {
    const int c = threadIdx.x;
    if (c < 16 && keep) keep[c] = info[c];
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

}  // namespace orcvio_amd

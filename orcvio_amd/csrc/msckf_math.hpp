// msckf_math.hpp -- per-observation MSCKF measurement Jacobian, shared by the HIP
// kernels (device) and by a host-compiled unit test of the same inline function.
//
// Follows reference src/orcvio.cpp:1071-1168 (measurementJacobian_msckf) using the
// closed forms of SURVEY.md 8a row 3:
//   R_w2c = R_b2c R_b2w^T, t_c_w = t_b_w + R_b2w t_c_b, p_c = R_w2c (p_w - t_c_w)
//   dz    = [[1/z, 0, -x/z^2], [0, 1/z, -y/z^2]]
//   H_f   = dz R_w2c                                   (:1161)
//   H_x   = dz [R_w2c skew(p_bf), -R_w2c]              LARVIO (:1145-1149); OrcVIO-left is
//           the same expression with p_bf = p_w - t_b_w (no FEJ) (:1116-1137)
//   H_x   = dz [R_b2c skew(t_c_b) + skew(p_c) R_b2c, -R_w2c]   OrcVIO-right (:1138-1143)
//   H_e   = dz [R_w2c skew(p_bf) R_b2w - R_b2c skew(t_c_b), -R_b2c]   (:1152-1160)
//   r     = z - (x/z, y/z)                             (:1165)
#pragma once

#if defined(__HIPCC__)
#define ORC_HD __host__ __device__ __forceinline__
#else
#define ORC_HD inline
#endif

namespace orcvio_amd {

// pose record per clone: 28 doubles (27 used) so that records stay 16-byte aligned
enum { POSE_STRIDE = 28, POSE_R_B2W = 0, POSE_T_B_W = 9, POSE_T_FEJ = 12, POSE_R_B2C = 15, POSE_T_C_B = 24 };

struct ObsFlags {
    int use_larvio, use_left, if_fej;
};

// out: Hx[2][6], He[2][6], Hf[2][3], r[2]
ORC_HD void obs_jacobian(const double* pose, const double* p_w, const double* z, ObsFlags f,
                         double Hx[12], double He[12], double Hf[6], double r[2]) {
    const double* Rbw = pose + POSE_R_B2W;
    const double* tbw = pose + POSE_T_B_W;
    const double* tfj = pose + POSE_T_FEJ;
    const double* Rbc = pose + POSE_R_B2C;
    const double* tcb = pose + POSE_T_C_B;
    double Rwc[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            Rwc[a * 3 + b] = Rbc[a * 3 + 0] * Rbw[b * 3 + 0] + Rbc[a * 3 + 1] * Rbw[b * 3 + 1] + Rbc[a * 3 + 2] * Rbw[b * 3 + 2];
    double tcw[3], d[3], pc[3], pbf[3];
    for (int a = 0; a < 3; ++a) tcw[a] = tbw[a] + Rbw[a * 3 + 0] * tcb[0] + Rbw[a * 3 + 1] * tcb[1] + Rbw[a * 3 + 2] * tcb[2];
    for (int a = 0; a < 3; ++a) d[a] = p_w[a] - tcw[a];
    for (int a = 0; a < 3; ++a) pc[a] = Rwc[a * 3 + 0] * d[0] + Rwc[a * 3 + 1] * d[1] + Rwc[a * 3 + 2] * d[2];
    for (int a = 0; a < 3; ++a) pbf[a] = f.if_fej ? (p_w[a] - tfj[a]) : (p_w[a] - tbw[a]);
    const double iz = 1.0 / pc[2];
    const double dz02 = -pc[0] / (pc[2] * pc[2]);
    const double dz12 = -pc[1] / (pc[2] * pc[2]);
    // rows of dz*X for a 3xK matrix X: row0 = iz*X[0] + dz02*X[2]; row1 = iz*X[1] + dz12*X[2]
#define ORC_DZ(dst, ld, col, X0, X1, X2)           \
    dst[0 * ld + col] = iz * (X0) + dz02 * (X2);   \
    dst[1 * ld + col] = iz * (X1) + dz12 * (X2);
    for (int b = 0; b < 3; ++b) { ORC_DZ(Hf, 3, b, Rwc[0 * 3 + b], Rwc[1 * 3 + b], Rwc[2 * 3 + b]) }
    // A = R_w2c * skew(p)   (skew(p) columns: [0,p2,-p1], [-p2,0,p0], [p1,-p0,0])
    double A[9];
    for (int a = 0; a < 3; ++a) {
        A[a * 3 + 0] = Rwc[a * 3 + 1] * pbf[2] - Rwc[a * 3 + 2] * pbf[1];
        A[a * 3 + 1] = -Rwc[a * 3 + 0] * pbf[2] + Rwc[a * 3 + 2] * pbf[0];
        A[a * 3 + 2] = Rwc[a * 3 + 0] * pbf[1] - Rwc[a * 3 + 1] * pbf[0];
    }
    // B = R_b2c * skew(t_c_b)
    double B[9];
    for (int a = 0; a < 3; ++a) {
        B[a * 3 + 0] = Rbc[a * 3 + 1] * tcb[2] - Rbc[a * 3 + 2] * tcb[1];
        B[a * 3 + 1] = -Rbc[a * 3 + 0] * tcb[2] + Rbc[a * 3 + 2] * tcb[0];
        B[a * 3 + 2] = Rbc[a * 3 + 0] * tcb[1] - Rbc[a * 3 + 1] * tcb[0];
    }
    if (f.use_larvio || f.use_left) {
        // OrcVIO-left equals the LARVIO expression evaluated WITHOUT FEJ
        double Al[9];
        if (!f.use_larvio && f.if_fej) {
            double q[3] = {p_w[0] - tbw[0], p_w[1] - tbw[1], p_w[2] - tbw[2]};
            for (int a = 0; a < 3; ++a) {
                Al[a * 3 + 0] = Rwc[a * 3 + 1] * q[2] - Rwc[a * 3 + 2] * q[1];
                Al[a * 3 + 1] = -Rwc[a * 3 + 0] * q[2] + Rwc[a * 3 + 2] * q[0];
                Al[a * 3 + 2] = Rwc[a * 3 + 0] * q[1] - Rwc[a * 3 + 1] * q[0];
            }
        } else {
            for (int a = 0; a < 9; ++a) Al[a] = A[a];
        }
        for (int b = 0; b < 3; ++b) {
            ORC_DZ(Hx, 6, b, Al[0 * 3 + b], Al[1 * 3 + b], Al[2 * 3 + b])
            ORC_DZ(Hx, 6, 3 + b, -Rwc[0 * 3 + b], -Rwc[1 * 3 + b], -Rwc[2 * 3 + b])
        }
    } else {
        // C = B + skew(p_c) * R_b2c ; skew(p) rows: [0,-p2,p1], [p2,0,-p0], [-p1,p0,0]
        double Cm[9];
        for (int b = 0; b < 3; ++b) {
            Cm[0 * 3 + b] = B[0 * 3 + b] + (-pc[2] * Rbc[1 * 3 + b] + pc[1] * Rbc[2 * 3 + b]);
            Cm[1 * 3 + b] = B[1 * 3 + b] + (pc[2] * Rbc[0 * 3 + b] - pc[0] * Rbc[2 * 3 + b]);
            Cm[2 * 3 + b] = B[2 * 3 + b] + (-pc[1] * Rbc[0 * 3 + b] + pc[0] * Rbc[1 * 3 + b]);
        }
        for (int b = 0; b < 3; ++b) {
            ORC_DZ(Hx, 6, b, Cm[0 * 3 + b], Cm[1 * 3 + b], Cm[2 * 3 + b])
            ORC_DZ(Hx, 6, 3 + b, -Rwc[0 * 3 + b], -Rwc[1 * 3 + b], -Rwc[2 * 3 + b])
        }
    }
    // H_e: [A*R_b2w - B, -R_b2c]
    for (int b = 0; b < 3; ++b) {
        double e0 = A[0] * Rbw[0 * 3 + b] + A[1] * Rbw[1 * 3 + b] + A[2] * Rbw[2 * 3 + b] - B[0 * 3 + b];
        double e1 = A[3] * Rbw[0 * 3 + b] + A[4] * Rbw[1 * 3 + b] + A[5] * Rbw[2 * 3 + b] - B[1 * 3 + b];
        double e2 = A[6] * Rbw[0 * 3 + b] + A[7] * Rbw[1 * 3 + b] + A[8] * Rbw[2 * 3 + b] - B[2 * 3 + b];
        ORC_DZ(He, 6, b, e0, e1, e2)
        ORC_DZ(He, 6, 3 + b, -Rbc[0 * 3 + b], -Rbc[1 * 3 + b], -Rbc[2 * 3 + b])
    }
#undef ORC_DZ
    r[0] = z[0] - pc[0] / pc[2];
    r[1] = z[1] - pc[1] / pc[2];
}

}  // namespace orcvio_amd

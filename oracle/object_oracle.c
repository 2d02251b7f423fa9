/*
 * object_oracle.c -- TEST / BENCH INFRASTRUCTURE ONLY: plain-C restatement of the OBJECT half of the update path
 * (SURVEY.md 8a rows 12-18), the C twin of oracle/mirror_objects.py and bench.py's CPU baseline for the object update.
 *
 *   orc_oracle_object_rows     rows 12-16: CameraLM / ObjectLM residual rows at a fixed state
 *                              (src/obj/ObjectResJacCam.cpp:153-519, src/obj/ObjectLM.cpp:250-632,
 *                              include/orcvio/utils/se3_ops.hpp:229-240,325-453) and their re-indexing into the window by
 *                              OrcVIO::constructObjectResidualJacobians (src/orcvio.cpp:2017-2151); output rows interleaved
 *                              per in-window frame [keypoint rows ; 4 bbox rows], Hx as its 6 non-zeros per row
 *   orc_oracle_objects_update  rows 17-18: per object the left-nullspace projection against Hf (math_utils.hpp:287-312: the
 *                              reference takes the last rows - cols columns of a full-U SVD; any orthonormal basis of that
 *                              nullspace gives the same update: Householder QR of Hf, Q^T applied to the DENSE zero-filled
 *                              rows x n Hx as the reference's A^T H_x product does), the blocks stacked
 *                              (System.cpp:684-702 with per-object projection, SURVEY note N3), gate with dof = rows
 *                              (src/orcvio.cpp:2172-2176) and measurementUpdate_msckf (:1654-1763).
 *                              The gate is evaluated through the identity of SURVEY Appendix A on the QR-compressed stack
 *                              (the literal H P H^T of 15 000 stacked rows is a 1.8 GB matrix).
 * PARITY STATUS: the row functions are pinned by the reference's HDF5 goldens (tests/test_oracle_objects.py holds this file to
 * them and to mirror_objects.py); the update half is unpinned like msckf_oracle.c.  Nothing under orcvio_amd/ may link this.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_oracle_house_qr(double* A, int m, int n, int lda, double* beta);
void orc_oracle_house_apply_qt(const double* A, int m, int n, int lda, const double* beta, double* C, int nc, int ldc);
int orc_oracle_chol_lower(double* S, int n);
void orc_oracle_cam_wrt_imu(const double* R_b2c, const double* t_c_b, const double* R_w2c, const double* t_b_w, int left, double J[36]);
int orc_oracle_measurement_update(double* H, double* r, int m, int n, const double* P, double sigma2, double* dx, double* P_out,
                                  double* H_thin_out, double* r_thin_out, double* K_out, double* G_out);
double orc_oracle_chi2_quantile(int dof, double p);

static void m44(const double* A, const double* B, double* C) {
    double T[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) T[i * 4 + j] = A[i * 4] * B[j] + A[i * 4 + 1] * B[4 + j] + A[i * 4 + 2] * B[8 + j] + A[i * 4 + 3] * B[12 + j];
    memcpy(C, T, sizeof(T));
}
static void m4v(const double* A, const double* x, double* y) {
    for (int i = 0; i < 4; ++i) y[i] = A[i * 4] * x[0] + A[i * 4 + 1] * x[1] + A[i * 4 + 2] * x[2] + A[i * 4 + 3] * x[3];
}
static void rigid_inverse(const double* T, double* Ti) {   /* Sophus SE3::inverse: (R^T, -R^T t) */
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Ti[i * 4 + j] = T[j * 4 + i];
        Ti[i * 4 + 3] = -(T[0 * 4 + i] * T[3] + T[1 * 4 + i] * T[7] + T[2 * 4 + i] * T[11]);
    }
    Ti[12] = Ti[13] = Ti[14] = 0.0; Ti[15] = 1.0;
}
/* se3_ops.hpp:510-519 odotOperator: [x4 I3, -skew(x123); 0] (4 x 6) */
static void odot46(const double* x, double* T) {
    memset(T, 0, 24 * sizeof(double));
    for (int i = 0; i < 3; ++i) T[i * 6 + i] = x[3];
    T[0 * 6 + 4] = x[2];  T[0 * 6 + 5] = -x[1];
    T[1 * 6 + 3] = -x[2]; T[1 * 6 + 5] = x[0];
    T[2 * 6 + 3] = x[1];  T[2 * 6 + 4] = -x[0];
}
/* out (1 x 6) = w (1 x 4) * circledCirc(x)^T, circledCirc (se3_ops.hpp:229-240) 6 x 4: rows 0:3 col 3 = x123, rows 3:6 cols 0:3 = -skew(x123) */
static void row_times_ccT(const double* w, const double* x, double* out) {
    out[0] = w[3] * x[0]; out[1] = w[3] * x[1]; out[2] = w[3] * x[2];
    out[3] = w[1] * x[2] - w[2] * x[1];
    out[4] = w[2] * x[0] - w[0] * x[2];
    out[5] = w[0] * x[1] - w[1] * x[0];
}

/* Rows of ONE object in window coordinates.  frame_clone[f] < 0: the frame is not in the window (dropped, :2073).
 * Outputs sized for 2 K + 4 rows per frame: row_clone, Hx6 [rows][6], Hf [rows][9 + 3K], res.  Returns the row count. */
int orc_oracle_object_rows(int K, int F, const double* wTo, const double* shape, const double* kps, const double* frame_wTc,
                           const double* frame_zs, const double* frame_bbox, const int* frame_clone, int obj_left, int new_bbox,
                           int vio_left, int fix_D, const double* R_b2c, const double* t_c_b,
                           int* row_clone, double* Hx6, double* Hf, double* res) {
    const int ncol = 9 + 3 * K;
    int row = 0;
    for (int f = 0; f < F; ++f) {
        if (frame_clone[f] < 0) continue;
        const double* wTc = frame_wTc + 16 * f;
        double cTw[16], cTo[16], D[36];
        rigid_inverse(wTc, cTw);
        m44(cTw, wTo, cTo);
        if (fix_D) {
            memset(D, 0, sizeof(D));
            for (int i = 0; i < 6; ++i) D[i * 6 + i] = 1.0;
        } else {   /* src/orcvio.cpp:2079-2093 with the current extrinsics */
            double Rw2c[9], v[3], tbw[3];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rw2c[i * 3 + j] = cTw[i * 4 + j];
            for (int i = 0; i < 3; ++i) v[i] = -(R_b2c[i * 3] * t_c_b[0] + R_b2c[i * 3 + 1] * t_c_b[1] + R_b2c[i * 3 + 2] * t_c_b[2]);
            for (int i = 0; i < 3; ++i) tbw[i] = wTc[i * 4] * v[0] + wTc[i * 4 + 1] * v[1] + wTc[i * 4 + 2] * v[2] + wTc[i * 4 + 3];
            orc_oracle_cam_wrt_imu(R_b2c, t_c_b, Rw2c, tbw, vio_left, D);
        }
#define EMIT(rr, jc, hpose, hshape, kpid, hkp)                                                            \
        do {                                                                                              \
            res[row] = (rr);                                                                              \
            row_clone[row] = frame_clone[f];                                                              \
            for (int c = 0; c < 6; ++c) {                                                                 \
                double s_ = 0;                                                                            \
                for (int k = 0; k < 6; ++k) s_ += (jc)[k] * D[k * 6 + c];                                 \
                Hx6[(size_t)row * 6 + c] = s_;                                                            \
            }                                                                                             \
            double* hf_ = Hf + (size_t)row * ncol;                                                        \
            memset(hf_, 0, (size_t)ncol * sizeof(double));                                                \
            for (int c = 0; c < 6; ++c) hf_[c] = (hpose)[c];                                              \
            if (hshape) for (int c = 0; c < 3; ++c) hf_[6 + c] = ((const double*)(hshape))[c];            \
            if (hkp) for (int c = 0; c < 3; ++c) hf_[9 + 3 * (kpid) + c] = ((const double*)(hkp))[c];     \
            ++row;                                                                                        \
        } while (0)
        /* ---- keypoint rows (ObjectResJacCam.cpp:153-282, ObjectLM.cpp:272-346) ------------------------------------ */
        for (int k = 0; k < K; ++k) {
            const double z0 = frame_zs[((size_t)f * K + k) * 2], z1 = frame_zs[((size_t)f * K + k) * 2 + 1];
            if (!(isfinite(z0) && isfinite(z1))) continue;                      /* ObjectLM.cpp:171-198 */
            const double Xo[4] = {kps[3 * k], kps[3 * k + 1], kps[3 * k + 2], 1.0};
            double Xw[4], Xc[4];
            m4v(wTo, Xo, Xw);
            m4v(cTw, Xw, Xc);
            const double iz = 1.0 / Xc[2], z2 = Xc[2] * Xc[2];
            const double dpi[6] = {iz, 0, -Xc[0] / z2, 0, iz, -Xc[1] / z2};      /* project_image_df */
            double O[24], M1[18], M2[18];
            if (obj_left) {   /* -dpi [I 0] cTw odot(wTo X)  /  dpi P odot(wTo X) */
                odot46(Xw, O);
                for (int i = 0; i < 3; ++i)
                    for (int c = 0; c < 6; ++c) {
                        double s = 0;
                        for (int q = 0; q < 4; ++q) s += cTw[i * 4 + q] * O[q * 6 + c];
                        M1[i * 6 + c] = s; M2[i * 6 + c] = s;
                    }
            } else {          /* -dpi [I 0] odot(cTw wTo X)  /  dpi P wTo odot(X) */
                odot46(Xc, O);
                for (int i = 0; i < 3; ++i) for (int c = 0; c < 6; ++c) M1[i * 6 + c] = O[i * 6 + c];
                odot46(Xo, O);
                for (int i = 0; i < 3; ++i)
                    for (int c = 0; c < 6; ++c) {
                        double s = 0;
                        for (int q = 0; q < 4; ++q) s += cTo[i * 4 + q] * O[q * 6 + c];
                        M2[i * 6 + c] = s;
                    }
            }
            for (int a = 0; a < 2; ++a) {
                double jc[6], hp[6], hk[3];
                for (int c = 0; c < 6; ++c) {
                    jc[c] = -(dpi[a * 3] * M1[c] + dpi[a * 3 + 1] * M1[6 + c] + dpi[a * 3 + 2] * M1[12 + c]);
                    hp[c] = dpi[a * 3] * M2[c] + dpi[a * 3 + 1] * M2[6 + c] + dpi[a * 3 + 2] * M2[12 + c];
                }
                for (int c = 0; c < 3; ++c) hk[c] = dpi[a * 3] * cTo[c] + dpi[a * 3 + 1] * cTo[4 + c] + dpi[a * 3 + 2] * cTo[8 + c];   /* ObjectLM.cpp:336-341 */
                const double r = (a == 0 ? Xc[0] * iz - z0 : Xc[1] * iz - z1);
                EMIT(r, jc, hp, (const double*)0, k, hk);
            }
        }
        /* ---- bbox rows (ObjectResJacCam.cpp:308-494, ObjectLM.cpp:441-616) --------------------------------------- */
        const double* bb = frame_bbox + 4 * f;
        const double px[4] = {bb[0], bb[2], bb[2], bb[0]}, py[4] = {bb[1], bb[1], bb[3], bb[3]};   /* bbox2poly */
        const double v2[3] = {shape[0] * shape[0], shape[1] * shape[1], shape[2] * shape[2]};
        for (int j = 0; j < 4; ++j) {
            const int j1 = (j + 1) & 3;
            const double ln[3] = {py[j] - py[j1], px[j1] - px[j], px[j] * py[j1] - py[j] * px[j1]};   /* poly2lineh: cross product */
            double yyw[4], yyo[4];
            for (int c = 0; c < 4; ++c) yyw[c] = ln[0] * cTw[c] + ln[1] * cTw[4 + c] + ln[2] * cTw[8 + c];
            for (int c = 0; c < 4; ++c) yyo[c] = yyw[0] * wTo[c] + yyw[1] * wTo[4 + c] + yyw[2] * wTo[8 + c] + yyw[3] * wTo[12 + c];
            const double lprime[4] = {ln[0], ln[1], ln[2], 0.0};
            double r, jc[6], hp[6], hs[3];
            if (!new_bbox) {
                r = v2[0] * yyo[0] * yyo[0] + v2[1] * yyo[1] * yyo[1] + v2[2] * yyo[2] * yyo[2] - yyo[3] * yyo[3];
                const double u[4] = {2 * yyo[0] * v2[0], 2 * yyo[1] * v2[1], 2 * yyo[2] * v2[2], -2 * yyo[3]};   /* 2 yyo Qi */
                double w[4];
                m4v(wTo, u, w);                       /* (u wTo^T)^T */
                if (obj_left) {
                    row_times_ccT(w, yyw, hp);        /* ObjectLM.cpp:527-533 */
                    for (int c = 0; c < 6; ++c) jc[c] = -hp[c];   /* ResJacCam.cpp:425-430 */
                } else {
                    double wc[4], t[4];
                    m4v(cTw, w, wc);                  /* (u wTo^T wTc^-T)^T */
                    row_times_ccT(wc, lprime, jc);    /* :436 */
                    for (int c = 0; c < 6; ++c) jc[c] = -jc[c];
                    for (int c = 0; c < 4; ++c) t[c] = yyo[c];   /* circledCirc(wTo^T yyw) = circledCirc(yyo) */
                    row_times_ccT(u, t, hp);          /* ObjectLM.cpp:540-541 */
                }
                for (int c = 0; c < 3; ++c) hs[c] = 2 * shape[c] * yyo[c] * yyo[c];   /* ObjectLM.cpp:546-548 */
            } else {
                {   /* residual: the plane in the OBJECT frame (ResJacCam.cpp:314,334) */
                    const double sq = sqrt(v2[0] * yyo[0] * yyo[0] + v2[1] * yyo[1] * yyo[1] + v2[2] * yyo[2] * yyo[2]);
                    const double bn = sqrt(yyo[0] * yyo[0] + yyo[1] * yyo[1] + yyo[2] * yyo[2]);
                    r = (yyo[3] - (yyo[3] > 0 ? 1.0 : -1.0) * sq) / bn;
                }
                /* Jacobians: the plane in the WORLD frame, as the reference computes them (SURVEY note N8) */
                const double* ub = yyw;
                const double sq = sqrt(v2[0] * ub[0] * ub[0] + v2[1] * ub[1] * ub[1] + v2[2] * ub[2] * ub[2]);
                const double bn = sqrt(ub[0] * ub[0] + ub[1] * ub[1] + ub[2] * ub[2]);
                const double sg = ub[3] > 0 ? 1.0 : -1.0;
                const double pa[4] = {-sg * v2[0] * ub[0] / sq, -sg * v2[1] * ub[1] / sq, -sg * v2[2] * ub[2] / sq, 1.0};
                const double pu = pa[0] * ub[0] + pa[1] * ub[1] + pa[2] * ub[2] + pa[3] * ub[3];
                double g[4], w[4];
                for (int c = 0; c < 4; ++c) g[c] = pa[c] / bn - ((c < 3) ? pu * ub[c] / (bn * bn * bn) : 0.0);
                m4v(wTo, g, w);
                if (obj_left) {
                    row_times_ccT(w, yyw, hp);        /* ObjectLM.cpp:598-599 */
                    for (int c = 0; c < 6; ++c) jc[c] = -hp[c];   /* ResJacCam.cpp:487 */
                } else {
                    double wc[4];
                    m4v(cTw, w, wc);
                    row_times_ccT(wc, lprime, jc);
                    for (int c = 0; c < 6; ++c) jc[c] = -jc[c];
                    row_times_ccT(g, yyo, hp);
                }
                for (int c = 0; c < 3; ++c) hs[c] = shape[c] * ub[c] * ub[c] / (bn * sq);   /* ObjectLM.cpp:602-603 */
            }
            EMIT(r, jc, hp, hs, 0, (const double*)0);
        }
#undef EMIT
    }
    return row;
}

/* The object update from pre-evaluated rows of several objects (CSR over objects: row_ptr[nobj + 1]; Hf with ld ncol_max,
 * ncol[o] live columns).  Outputs: accept, gamma, dof, dx [n], P_out [n x n].  Returns 0, or -1 if S is not SPD. */
int orc_oracle_objects_update(int n_clones, int leg, int nobj, const int* row_ptr, const int* ncol, int ncol_max, const int* row_clone,
                              const double* Hx6, const double* Hf, const double* res, const double* P, double sigma, double chi2_prob,
                              int* accept, double* gamma, int* dof_out, double* dx, double* P_out) {
    const int n = leg + 6 * n_clones;
    const double s2 = sigma * sigma;
    int tot = 0;
    for (int o = 0; o < nobj; ++o) {
        const int m = row_ptr[o + 1] - row_ptr[o];
        if (m > ncol[o]) tot += m - ncol[o];
    }
    *accept = 0; *gamma = NAN; *dof_out = tot;
    memset(dx, 0, (size_t)n * sizeof(double));
    memcpy(P_out, P, (size_t)n * n * sizeof(double));
    if (tot == 0) return 0;
    double* H = (double*)calloc((size_t)tot * n, sizeof(double));
    double* r = (double*)calloc((size_t)tot, sizeof(double));
    int at = 0;
    for (int o = 0; o < nobj; ++o) {
        const int r0 = row_ptr[o], m = row_ptr[o + 1] - r0, nc = ncol[o];
        if (m <= nc) continue;                                            /* nullspace_project_inplace_svd returns false */
        double* A = (double*)malloc((size_t)m * nc * sizeof(double));
        double* X = (double*)calloc((size_t)m * (n + 1), sizeof(double));   /* dense zero-filled [Hx | r], as the reference holds it */
        double* beta = (double*)calloc((size_t)nc, sizeof(double));
        for (int i = 0; i < m; ++i) {
            memcpy(A + (size_t)i * nc, Hf + (size_t)(r0 + i) * ncol_max, (size_t)nc * sizeof(double));
            for (int c = 0; c < 6; ++c) X[(size_t)i * (n + 1) + leg + 6 * row_clone[r0 + i] + c] = Hx6[(size_t)(r0 + i) * 6 + c];
            X[(size_t)i * (n + 1) + n] = res[r0 + i];
        }
        orc_oracle_house_qr(A, m, nc, nc, beta);
        orc_oracle_house_apply_qt(A, m, nc, nc, beta, X, n + 1, n + 1);
        for (int i = nc; i < m; ++i) {
            memcpy(H + (size_t)at * n, X + (size_t)i * (n + 1), (size_t)n * sizeof(double));
            r[at++] = X[(size_t)i * (n + 1) + n];
        }
        free(A); free(X); free(beta);
    }
    /* gate through the compressed form: Q^T [H | r] = [R, r1; 0, r2], gamma = r1^T (R P R^T + s2 I)^-1 r1 + |r2|^2 / s2 */
    double rr = 0.0;
    for (int i = 0; i < tot; ++i) rr += r[i] * r[i];
    int mt = tot;
    double* Hc = (double*)malloc((size_t)tot * n * sizeof(double));
    double* rc = (double*)malloc((size_t)tot * sizeof(double));
    memcpy(Hc, H, (size_t)tot * n * sizeof(double));
    memcpy(rc, r, (size_t)tot * sizeof(double));
    if (tot > n) {
        double* beta = (double*)calloc((size_t)n, sizeof(double));
        orc_oracle_house_qr(Hc, tot, n, n, beta);
        orc_oracle_house_apply_qt(Hc, tot, n, n, beta, rc, 1, 1);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j) Hc[(size_t)i * n + j] = 0.0;
        free(beta);
        mt = n;
    }
    double r1r1 = 0.0;
    for (int i = 0; i < mt; ++i) r1r1 += rc[i] * rc[i];
    double* HP = (double*)calloc((size_t)mt * n, sizeof(double));
    double* S = (double*)calloc((size_t)mt * mt, sizeof(double));
    for (int i = 0; i < mt; ++i)
        for (int k = 0; k < n; ++k) {
            const double h = Hc[(size_t)i * n + k];
            if (h == 0.0) continue;
            for (int j = 0; j < n; ++j) HP[(size_t)i * n + j] += h * P[(size_t)k * n + j];
        }
    for (int i = 0; i < mt; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = (i == j) ? s2 : 0.0;
            for (int k = 0; k < n; ++k) s += HP[(size_t)i * n + k] * Hc[(size_t)j * n + k];
            S[(size_t)i * mt + j] = s; S[(size_t)j * mt + i] = s;
        }
    int rcode = 0;
    if (orc_oracle_chol_lower(S, mt) != 0) rcode = -1;
    else {
        double g = 0.0;
        double* y = (double*)malloc((size_t)mt * sizeof(double));
        for (int i = 0; i < mt; ++i) {
            double s = rc[i];
            for (int k = 0; k < i; ++k) s -= S[(size_t)i * mt + k] * y[k];
            y[i] = s / S[(size_t)i * mt + i];
            g += y[i] * y[i];
        }
        free(y);
        g += (rr - r1r1) / s2;
        *gamma = g;
        const double thr = orc_oracle_chi2_quantile(tot, chi2_prob);      /* dof = rows (:2172), on the fly above the table */
        int has_nan = !(g == g);
        if (!has_nan && g < thr) {
            *accept = 1;
            if (orc_oracle_measurement_update(H, r, tot, n, P, s2, dx, P_out, 0, 0, 0, 0) < 0) rcode = -1;
        }
    }
    free(H); free(r); free(Hc); free(rc); free(HP); free(S);
    return rcode;
}

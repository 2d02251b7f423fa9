/*
 * msckf_oracle.c -- CPU restatement of the reference MSCKF measurement update.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the
 * `cpu_baseline` of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or call it; nothing under orcvio_amd/
 * does.  Plain C99, FP64, single thread, no dependencies.
 *
 * PARITY STATUS: the reference (shanmo/OrcVIO) cannot be compiled in this
 * image (needs Eigen, SuiteSparse/SPQR, Sophus, Boost.Math, OpenCV -- none
 * present) and its test-suite holds no golden vector for the
 * feature-side arithmetic, so for the functions in this file **parity is
 * unpinned**: the anchors are (i) operation-for-operation agreement with the
 * numpy mirror oracle/mirror.py, (ii) central-difference checks of every
 * Jacobian variant, (iii) the committed vectors tests/golden/feat_*.npz that
 * were generated from the mirror by scripts/make_golden.py.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference).  Third-party arithmetic that is not vendored there is
 * restated by its standard algorithm:
 *   Eigen JacobiSVD(ComputeFullU) left-nullspace  -> Householder QR, Q2 = last
 *       rows-3 columns of the full Q (any orthonormal basis of the left
 *       nullspace gives the same gamma, dx, K*H, P+; the reference's own
 *       test-only twin nullspace_project_inplace_qr does exactly this,
 *       include/orcvio/utils/math_utils.hpp:315-344);
 *   Eigen LDLT::solve on an SPD matrix            -> Cholesky solve;
 *   SuiteSparse SPQR (natural ordering) Q^T*H     -> dense Householder QR, top
 *       LEG+6N rows kept;
 *   Boost.Math quantile(chi_squared(dof), p)      -> Newton/bisection on the
 *       regularised lower incomplete gamma function.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX(r, c, ld) ((size_t)(r) * (size_t)(ld) + (size_t)(c)) /* row-major */

/* ------------------------------------------------------------------------ */
/* chi-square quantile (Boost.Math quantile, src/orcvio.cpp:486-494,1962-1968) */
/* ------------------------------------------------------------------------ */
static double gammainc_lower_reg(double a, double x) {
    /* regularised P(a,x): series for x < a+1, Lentz continued fraction else */
    if (x <= 0.0) return 0.0;
    double gln = lgamma(a);
    if (x < a + 1.0) {
        double ap = a, sum = 1.0 / a, del = sum;
        for (int it = 0; it < 100000; ++it) {
            ap += 1.0;
            del *= x / ap;
            sum += del;
            if (fabs(del) < fabs(sum) * 1e-17) break;
        }
        return sum * exp(-x + a * log(x) - gln);
    }
    double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, h = d;
    for (int i = 1; i < 100000; ++i) {
        double an = -i * (i - a);
        b += 2.0;
        d = an * d + b;
        if (fabs(d) < 1e-300) d = 1e-300;
        c = b + an / c;
        if (fabs(c) < 1e-300) c = 1e-300;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return 1.0 - exp(-x + a * log(x) - gln) * h;
}

double orc_oracle_chi2_quantile(int dof, double p) {
    double a = 0.5 * dof;
    /* Wilson-Hilferty start, then bracketed Newton on P(a, x/2) = p */
    double z = 1.6448536269514722; /* only a start value; refined below */
    if (p != 0.95) {
        /* crude normal quantile start by bisection on erf */
        double lo = -10, hi = 10;
        for (int i = 0; i < 200; ++i) {
            double m = 0.5 * (lo + hi);
            if (0.5 * (1.0 + erf(m / sqrt(2.0))) < p) lo = m; else hi = m;
        }
        z = 0.5 * (lo + hi);
    }
    double t = 1.0 - 2.0 / (9.0 * dof) + z * sqrt(2.0 / (9.0 * dof));
    double x = dof * t * t * t;
    if (x <= 0) x = 1e-3;
    double lo = 0.0, hi = x * 4.0 + 50.0;
    for (int it = 0; it < 200; ++it) {
        double f = gammainc_lower_reg(a, 0.5 * x) - p;
        if (f > 0) hi = x; else lo = x;
        /* pdf of chi2 */
        double lpdf = (a - 1.0) * log(0.5 * x) - 0.5 * x - lgamma(a) - log(2.0);
        double step = f / exp(lpdf);
        double xn = x - step;
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (fabs(xn - x) <= 1e-15 * fabs(x)) { x = xn; break; }
        x = xn;
    }
    return x;
}

/* ------------------------------------------------------------------------ */
/* small fixed-size helpers                                                  */
/* ------------------------------------------------------------------------ */
/* include/orcvio/utils/math_utils.hpp:27-39 skewSymmetric */
static void skew3(const double w[3], double S[9]) {
    S[0] = 0;     S[1] = -w[2]; S[2] = w[1];
    S[3] = w[2];  S[4] = 0;     S[5] = -w[0];
    S[6] = -w[1]; S[7] = w[0];  S[8] = 0;
}
static void mm(const double* A, const double* B, double* C, int m, int k, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static void tr3(const double* A, double* T) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[j * 3 + i] = A[i * 3 + j];
}

/* include/orcvio/utils/se3_ops.hpp:510-519 odotOperator (4x6 row-major) */
static void odot(const double x[4], double T[24]) {
    memset(T, 0, 24 * sizeof(double));
    double S[9];
    skew3(x, S);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 6 + 3 + j] = -S[i * 3 + j];
    T[0] = x[3]; T[7] = x[3]; T[14] = x[3];
}

/* include/orcvio/utils/se3_ops.hpp:531-552 get_cam_wrt_imu_se3_jacobian (6x6) */
static void cam_wrt_imu(const double* R_b2c, const double* t_c_b, const double* R_w2c,
                        const double* t_b_w, int left, double J[36]) {
    memset(J, 0, 36 * sizeof(double));
    double S[9];
    if (left) {
        skew3(t_b_w, S);
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) J[i * 6 + j] = S[i * 3 + j];
            J[(3 + i) * 6 + i] = 1.0;
            J[i * 6 + 3 + i] = 1.0;
        }
    } else {
        double RS[9];
        skew3(t_c_b, S);
        mm(R_b2c, S, RS, 3, 3, 3);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                J[i * 6 + j] = -RS[i * 3 + j];
                J[(3 + i) * 6 + j] = R_b2c[i * 3 + j];
                J[i * 6 + 3 + j] = R_w2c[i * 3 + j];
            }
    }
}

/* ------------------------------------------------------------------------ */
/* src/orcvio.cpp:1071-1168 measurementJacobian_msckf                         */
/*   flags: [0]=leg_dim [1]=use_larvio [2]=use_left [3]=if_FEJ [4]=estimate_td */
/* ------------------------------------------------------------------------ */
void orc_oracle_measurement_jacobian(const int* flags, const double* R_b2w, const double* t_b_w,
                                     const double* t_fej, const double* R_b2c, const double* t_c_b,
                                     const double* p_w, const double* z,
                                     double* H_x /*2x6*/, double* H_e /*2x6*/, double* H_f /*2x3*/,
                                     double* r /*2*/) {
    double R_w2b[9], R_w2c[9], t_c_w[3], d[3], p_c[3], p_bf[3];
    tr3(R_b2w, R_w2b);
    mm(R_b2c, R_w2b, R_w2c, 3, 3, 3);                                  /* :1090 */
    for (int i = 0; i < 3; ++i)                                         /* :1091 */
        t_c_w[i] = t_b_w[i] + R_b2w[i * 3] * t_c_b[0] + R_b2w[i * 3 + 1] * t_c_b[1] + R_b2w[i * 3 + 2] * t_c_b[2];
    for (int i = 0; i < 3; ++i) d[i] = p_w[i] - t_c_w[i];               /* :1099 */
    for (int i = 0; i < 3; ++i) p_c[i] = R_w2c[i * 3] * d[0] + R_w2c[i * 3 + 1] * d[1] + R_w2c[i * 3 + 2] * d[2];
    for (int i = 0; i < 3; ++i) p_bf[i] = flags[3] ? p_w[i] - t_fej[i] : p_w[i] - t_b_w[i];   /* :1104 */
    double dz[6] = {1 / p_c[2], 0, -p_c[0] / (p_c[2] * p_c[2]),         /* :1107-1111 */
                    0, 1 / p_c[2], -p_c[1] / (p_c[2] * p_c[2])};
    double dpc[18];
    if (!flags[1]) {                                                    /* :1115-1143 */
        double Tinv[16] = {0}, x4[4], od[24], D[36], M[24], t46[24];
        /* wTc.inverse() = [R_w2c, -R_w2c*t_c_w; 0 1] */
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) Tinv[i * 4 + j] = R_w2c[i * 3 + j];
            Tinv[i * 4 + 3] = -(R_w2c[i * 3] * t_c_w[0] + R_w2c[i * 3 + 1] * t_c_w[1] + R_w2c[i * 3 + 2] * t_c_w[2]);
        }
        Tinv[15] = 1.0;
        cam_wrt_imu(R_b2c, t_c_b, R_w2c, t_b_w, flags[2], D);
        if (flags[2]) {
            x4[0] = p_w[0]; x4[1] = p_w[1]; x4[2] = p_w[2]; x4[3] = 1.0;
            odot(x4, od);
            mm(Tinv, od, t46, 4, 4, 6);
            mm(t46, D, M, 4, 6, 6);
        } else {
            double ul[4] = {p_w[0], p_w[1], p_w[2], 1.0};
            for (int i = 0; i < 4; ++i)
                x4[i] = Tinv[i * 4] * ul[0] + Tinv[i * 4 + 1] * ul[1] + Tinv[i * 4 + 2] * ul[2] + Tinv[i * 4 + 3] * ul[3];
            odot(x4, od);
            mm(od, D, M, 4, 6, 6);
        }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) dpc[i * 6 + j] = M[i * 6 + j]; /* temp_mat * . */
        double Hx[12];
        mm(dz, dpc, Hx, 2, 3, 6);
        for (int i = 0; i < 12; ++i) H_x[i] = -Hx[i];                   /* :1143 */
    } else {                                                            /* :1145-1149 */
        double S[9], RS[9];
        skew3(p_bf, S);
        mm(R_w2c, S, RS, 3, 3, 3);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            dpc[i * 6 + j] = RS[i * 3 + j];
            dpc[i * 6 + 3 + j] = -R_w2c[i * 3 + j];
        }
        mm(dz, dpc, H_x, 2, 3, 6);
    }
    {                                                                   /* :1152-1160 */
        double S[9], RS[9], RSR[9], St[9], RcSt[9], dpe[18];
        skew3(p_bf, S);
        mm(R_w2c, S, RS, 3, 3, 3);
        mm(RS, R_b2w, RSR, 3, 3, 3);
        skew3(t_c_b, St);
        mm(R_b2c, St, RcSt, 3, 3, 3);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            dpe[i * 6 + j] = RSR[i * 3 + j] - RcSt[i * 3 + j];
            dpe[i * 6 + 3 + j] = -R_b2c[i * 3 + j];
        }
        mm(dz, dpe, H_e, 2, 3, 6);
    }
    mm(dz, R_w2c, H_f, 2, 3, 3);                                        /* :1161 */
    r[0] = z[0] - p_c[0] / p_c[2];                                      /* :1165 */
    r[1] = z[1] - p_c[1] / p_c[2];
}

/* ------------------------------------------------------------------------ */
/* dense linear algebra (row-major)                                          */
/* ------------------------------------------------------------------------ */
/* Householder QR of A (m x n, m >= n) in place; reflectors v_k stored below
 * the diagonal (v_k[k] = 1 implicit), beta[k].  LAPACK dgeqr2 convention. */
static void house_qr(double* A, int m, int n, int lda, double* beta) {
    for (int k = 0; k < n && k < m; ++k) {
        double nrm = 0;
        for (int i = k + 1; i < m; ++i) nrm += A[IDX(i, k, lda)] * A[IDX(i, k, lda)];
        double alpha = A[IDX(k, k, lda)];
        if (nrm == 0.0) { beta[k] = 0; continue; }
        double nu = sqrt(alpha * alpha + nrm);
        double bk = (alpha >= 0) ? -nu : nu;
        beta[k] = (bk - alpha) / bk;
        double sc = 1.0 / (alpha - bk);
        for (int i = k + 1; i < m; ++i) A[IDX(i, k, lda)] *= sc;
        A[IDX(k, k, lda)] = bk;
        /* apply to trailing columns */
        for (int j = k + 1; j < n; ++j) {
            double s = A[IDX(k, j, lda)];
            for (int i = k + 1; i < m; ++i) s += A[IDX(i, k, lda)] * A[IDX(i, j, lda)];
            s *= beta[k];
            A[IDX(k, j, lda)] -= s;
            for (int i = k + 1; i < m; ++i) A[IDX(i, j, lda)] -= s * A[IDX(i, k, lda)];
        }
    }
}
/* C <- Q^T C for the reflectors of house_qr (A: m x n factored), C: m x nc */
static void house_apply_qt(const double* A, int m, int n, int lda, const double* beta,
                           double* C, int nc, int ldc) {
    for (int k = 0; k < n && k < m; ++k) {
        if (beta[k] == 0.0) continue;
        for (int j = 0; j < nc; ++j) {
            double s = C[IDX(k, j, ldc)];
            for (int i = k + 1; i < m; ++i) s += A[IDX(i, k, lda)] * C[IDX(i, j, ldc)];
            s *= beta[k];
            C[IDX(k, j, ldc)] -= s;
            for (int i = k + 1; i < m; ++i) C[IDX(i, j, ldc)] -= s * A[IDX(i, k, lda)];
        }
    }
}
/* in-place Cholesky (lower) of SPD S (n x n); returns 0 on success */
static int chol_lower(double* S, int n) {
    for (int j = 0; j < n; ++j) {
        double d = S[IDX(j, j, n)];
        for (int k = 0; k < j; ++k) d -= S[IDX(j, k, n)] * S[IDX(j, k, n)];
        if (!(d > 0)) return 1;
        d = sqrt(d);
        S[IDX(j, j, n)] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = S[IDX(i, j, n)];
            for (int k = 0; k < j; ++k) s -= S[IDX(i, k, n)] * S[IDX(j, k, n)];
            S[IDX(i, j, n)] = s / d;
        }
    }
    return 0;
}
/* solve L L^T X = B in place, B: n x nb row-major */
static void chol_solve(const double* L, int n, double* B, int nb) {
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < i; ++k) {
            double l = L[IDX(i, k, n)];
            if (l != 0.0) for (int j = 0; j < nb; ++j) B[IDX(i, j, nb)] -= l * B[IDX(k, j, nb)];
        }
        double inv = 1.0 / L[IDX(i, i, n)];
        for (int j = 0; j < nb; ++j) B[IDX(i, j, nb)] *= inv;
    }
    for (int i = n - 1; i >= 0; --i) {
        for (int k = i + 1; k < n; ++k) {
            double l = L[IDX(k, i, n)];
            if (l != 0.0) for (int j = 0; j < nb; ++j) B[IDX(i, j, nb)] -= l * B[IDX(k, j, nb)];
        }
        double inv = 1.0 / L[IDX(i, i, n)];
        for (int j = 0; j < nb; ++j) B[IDX(i, j, nb)] *= inv;
    }
}

/* ------------------------------------------------------------------------ */
/* include/orcvio/utils/math_utils.hpp:287-312 nullspace_project_inplace_svd  */
/* H_f: m x nf, H_x: m x n, r: m.  On success rows shrink to m-nf (returned). */
/* Literal cost structure: explicit A (m x (m-nf)), dense A^T*H_x.            */
/* ------------------------------------------------------------------------ */
int orc_oracle_nullspace_project(double* H_f, int m, int nf, double* H_x, int n, double* r) {
    if (m <= nf) return -1;                           /* :292-296 */
    double* beta = (double*)calloc(nf, sizeof(double));
    house_qr(H_f, m, nf, nf, beta);
    /* explicit Q = H_1...H_nf applied to I, then A = Q[:, nf:] */
    double* Q = (double*)calloc((size_t)m * m, sizeof(double));
    for (int i = 0; i < m; ++i) Q[IDX(i, i, m)] = 1.0;
    for (int k = nf - 1; k >= 0; --k) {               /* Q = H_0 (H_1 (... I)) */
        if (beta[k] == 0.0) continue;
        for (int j = 0; j < m; ++j) {
            double s = Q[IDX(k, j, m)];
            for (int i = k + 1; i < m; ++i) s += H_f[IDX(i, k, nf)] * Q[IDX(i, j, m)];
            s *= beta[k];
            Q[IDX(k, j, m)] -= s;
            for (int i = k + 1; i < m; ++i) Q[IDX(i, j, m)] -= s * H_f[IDX(i, k, nf)];
        }
    }
    int mo = m - nf;
    double* Hn = (double*)calloc((size_t)mo * n, sizeof(double));
    double* rn = (double*)calloc(mo, sizeof(double));
    for (int a = 0; a < mo; ++a) {                    /* H_x = A^T H_x ; res = A^T res (:306-307) */
        for (int i = 0; i < m; ++i) {
            double q = Q[IDX(i, nf + a, m)];
            if (q == 0.0) continue;
            const double* hx = &H_x[IDX(i, 0, n)];
            double* ho = &Hn[IDX(a, 0, n)];
            for (int c = 0; c < n; ++c) ho[c] += q * hx[c];
            rn[a] += q * r[i];
        }
    }
    memcpy(H_x, Hn, (size_t)mo * n * sizeof(double));
    memcpy(r, rn, mo * sizeof(double));
    free(beta); free(Q); free(Hn); free(rn);
    return mo;
}

/* ------------------------------------------------------------------------ */
/* src/orcvio.cpp:1953-1976 gatingTestFeature: gamma = r^T (H P H^T + s2 I)^-1 r */
/* ------------------------------------------------------------------------ */
double orc_oracle_gating_gamma(const double* H, const double* r, int m, int n,
                               const double* P, double sigma2) {
    double* HP = (double*)calloc((size_t)m * n, sizeof(double));
    double* S = (double*)calloc((size_t)m * m, sizeof(double));
    double* x = (double*)malloc(m * sizeof(double));
    for (int i = 0; i < m; ++i)
        for (int k = 0; k < n; ++k) {
            double h = H[IDX(i, k, n)];
            if (h == 0.0) continue;
            const double* p = &P[IDX(k, 0, n)];
            double* o = &HP[IDX(i, 0, n)];
            for (int j = 0; j < n; ++j) o[j] += h * p[j];
        }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0;
            const double* a = &HP[IDX(i, 0, n)];
            const double* b = &H[IDX(j, 0, n)];
            for (int k = 0; k < n; ++k) s += a[k] * b[k];
            if (i == j) s += sigma2;
            S[IDX(i, j, m)] = s; S[IDX(j, i, m)] = s;
        }
    double g = NAN;
    if (chol_lower(S, m) == 0) {
        memcpy(x, r, m * sizeof(double));
        chol_solve(S, m, x, 1);
        g = 0;
        for (int i = 0; i < m; ++i) g += r[i] * x[i];
    }
    free(HP); free(S); free(x);
    return g;
}

/* ------------------------------------------------------------------------ */
/* src/orcvio.cpp:1654-1763 measurementUpdate_msckf (== pure-MSCKF case of    */
/* measurementUpdate_hybrid :1766-1950): optional QR compression :1664-1679,  */
/* S, K, dx, P <- (I-KH)P, symmetrise.  H: m x n row-major (destroyed).       */
/* Outputs (each may be NULL): H_thin (mt x n), r_thin, K (n x mt), G (n x n) */
/* Returns mt = rows used in the update.                                      */
/* ------------------------------------------------------------------------ */
int orc_oracle_measurement_update(double* H, double* r, int m, int n, const double* P, double sigma2,
                                  double* dx, double* P_out,
                                  double* H_thin_out, double* r_thin_out, double* K_out, double* G_out) {
    if (m == 0) {
        memset(dx, 0, n * sizeof(double));
        memcpy(P_out, P, (size_t)n * n * sizeof(double));
        return 0;
    }
    int mt = m;
    if (m > n) {                                      /* :1664-1679 / :2532-2552 */
        double* beta = (double*)calloc(n, sizeof(double));
        double* Hc = (double*)malloc((size_t)m * n * sizeof(double));
        memcpy(Hc, H, (size_t)m * n * sizeof(double));
        house_qr(Hc, m, n, n, beta);
        house_apply_qt(Hc, m, n, n, beta, r, 1, 1);   /* r_temp = Q^T r */
        /* H_temp = Q^T H: upper-triangular R in the top n rows */
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) H[IDX(i, j, n)] = (j >= i) ? Hc[IDX(i, j, n)] : 0.0;
        free(beta); free(Hc);
        mt = n;                                       /* top LEG+6N rows */
    }
    /* S = H P H^T + sigma2 I ; K^T = S^-1 (H P) */
    double* HP = (double*)calloc((size_t)mt * n, sizeof(double));
    double* S = (double*)calloc((size_t)mt * mt, sizeof(double));
    for (int i = 0; i < mt; ++i)
        for (int k = 0; k < n; ++k) {
            double h = H[IDX(i, k, n)];
            if (h == 0.0) continue;
            const double* p = &P[IDX(k, 0, n)];
            double* o = &HP[IDX(i, 0, n)];
            for (int j = 0; j < n; ++j) o[j] += h * p[j];
        }
    for (int i = 0; i < mt; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += HP[IDX(i, k, n)] * H[IDX(j, k, n)];
            if (i == j) s += sigma2;
            S[IDX(i, j, mt)] = s; S[IDX(j, i, mt)] = s;
        }
    double* KT = (double*)malloc((size_t)mt * n * sizeof(double));
    memcpy(KT, HP, (size_t)mt * n * sizeof(double));
    int bad = chol_lower(S, mt);
    if (bad) { free(HP); free(S); free(KT); return -1; }
    chol_solve(S, mt, KT, n);                         /* K_transpose = S.ldlt().solve(H*P) */
    for (int a = 0; a < n; ++a) {                     /* delta_x = K * r_thin */
        double s = 0;
        for (int i = 0; i < mt; ++i) s += KT[IDX(i, a, n)] * r[i];
        dx[a] = s;
    }
    /* G = K H ; P <- (I - G) P ; symmetrise */
    double* G = (double*)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < mt; ++i)
        for (int a = 0; a < n; ++a) {
            double k = KT[IDX(i, a, n)];
            if (k == 0.0) continue;
            const double* h = &H[IDX(i, 0, n)];
            double* g = &G[IDX(a, 0, n)];
            for (int c = 0; c < n; ++c) g[c] += k * h[c];
        }
    double* Pn = (double*)malloc((size_t)n * n * sizeof(double));
    memcpy(Pn, P, (size_t)n * n * sizeof(double));
    for (int a = 0; a < n; ++a)
        for (int k = 0; k < n; ++k) {
            double g = G[IDX(a, k, n)];
            if (g == 0.0) continue;
            const double* p = &P[IDX(k, 0, n)];
            double* o = &Pn[IDX(a, 0, n)];
            for (int c = 0; c < n; ++c) o[c] -= g * p[c];
        }
    for (int a = 0; a < n; ++a)
        for (int c = 0; c <= a; ++c) {
            double s = 0.5 * (Pn[IDX(a, c, n)] + Pn[IDX(c, a, n)]);
            P_out[IDX(a, c, n)] = s; P_out[IDX(c, a, n)] = s;
        }
    if (H_thin_out) memcpy(H_thin_out, H, (size_t)mt * n * sizeof(double));
    if (r_thin_out) memcpy(r_thin_out, r, mt * sizeof(double));
    if (K_out) for (int a = 0; a < n; ++a) for (int i = 0; i < mt; ++i) K_out[IDX(a, i, mt)] = KT[IDX(i, a, n)];
    if (G_out) memcpy(G_out, G, (size_t)n * n * sizeof(double));
    free(HP); free(S); free(KT); free(G); free(Pn);
    return mt;
}

/* ------------------------------------------------------------------------ */
/* src/orcvio.cpp:1171-1226 featureJacobian_msckf + the stacking loops        */
/* :2497-2527 (removeLostFeatures) / :2803-2848 (pruneImuStateBuffer when     */
/* clone_mask != NULL) + the update.                                          */
/* Optional outputs: H_stack (sum rho_j x n, row-major, only accepted blocks  */
/* in order), r_stack, block_ptr[F+1] (row offsets of every feature's block   */
/* in H_all/r_all which hold ALL projected blocks, accepted or not).          */
/* ------------------------------------------------------------------------ */
int orc_oracle_msckf_update(
    int N, int F, const int* flags, double sigma, double chi2_prob, const double* chi2_table, int chi2_table_len,
    const double* R_b2w, const double* t_b_w, const double* t_fej, const double* R_b2c, const double* t_c_b,
    const double* p_w, const int* obs_ptr, const int* obs_clone, const double* obs_z, const double* obs_zvel,
    const int* clone_mask, const double* P,
    double* dx, double* P_out, int* accept, double* gamma,
    double* H_all, double* r_all, int* block_ptr,
    double* H_thin, double* r_thin, double* K, double* G, int* info /* [0]=stacked rows [1]=thin rows */) {
    const int leg = flags[0];
    const int n = leg + 6 * N;
    const double sigma2 = sigma * sigma;
    /* upper bound on stacked rows */
    size_t max_rows = 0;
    int maxM = 0;
    for (int j = 0; j < F; ++j) {
        int M = obs_ptr[j + 1] - obs_ptr[j];
        max_rows += (size_t)(2 * M);
        if (M > maxM) maxM = M;
    }
    double* Hs = (double*)calloc((max_rows + 1) * (size_t)n, sizeof(double));
    double* rs = (double*)calloc(max_rows + 1, sizeof(double));
    double* Hxj = (double*)malloc((size_t)(2 * maxM + 1) * n * sizeof(double));
    double* Hfj = (double*)malloc((size_t)(2 * maxM + 1) * 3 * sizeof(double));
    double* rj = (double*)malloc((size_t)(2 * maxM + 1) * sizeof(double));
    int stack = 0, all_rows = 0;
    if (block_ptr) block_ptr[0] = 0;
    for (int j = 0; j < F; ++j) {
        /* valid_state_ids (:1179-1186) */
        int M = 0;
        for (int k = obs_ptr[j]; k < obs_ptr[j + 1]; ++k)
            if (!clone_mask || clone_mask[obs_clone[k]]) ++M;
        accept[j] = 0;
        gamma[j] = NAN;
        if (M < 2) { if (block_ptr) block_ptr[j + 1] = all_rows; continue; }   /* callers guarantee M>=2 (:2777) */
        int rows = 2 * M;
        memset(Hxj, 0, (size_t)rows * n * sizeof(double));
        int c = 0;
        for (int k = obs_ptr[j]; k < obs_ptr[j + 1]; ++k) {
            int i = obs_clone[k];
            if (clone_mask && !clone_mask[i]) continue;
            double Hx[12], He[12], Hf[6], r2[2];
            orc_oracle_measurement_jacobian(flags, &R_b2w[9 * i], &t_b_w[3 * i], &t_fej[3 * i],
                                            &R_b2c[9 * i], &t_c_b[3 * i], &p_w[3 * j], &obs_z[2 * k],
                                            Hx, He, Hf, r2);
            for (int a = 0; a < 2; ++a) {
                for (int b = 0; b < 6; ++b) {
                    Hxj[IDX(c + a, leg + 6 * i + b, n)] = Hx[a * 6 + b];      /* :1209 */
                    Hxj[IDX(c + a, 15 + b, n)] = He[a * 6 + b];               /* :1210 */
                }
                if (flags[4]) Hxj[IDX(c + a, 21, n)] = obs_zvel[2 * k + a];   /* :1211-1212 */
                for (int b = 0; b < 3; ++b) Hfj[IDX(c + a, b, 3)] = Hf[a * 3 + b];
                rj[c + a] = r2[a];
            }
            c += 2;
        }
        int mo = orc_oracle_nullspace_project(Hfj, rows, 3, Hxj, n, rj);      /* :1220 */
        if (mo < 0) mo = rows;   /* rows<=3: inputs untouched (math_utils.hpp:292) */
        int dof = 2 * M - 3;                                                  /* :2514 / :2835 */
        double g = orc_oracle_gating_gamma(Hxj, rj, mo, n, P, sigma2);
        double thr = (dof < chi2_table_len) ? chi2_table[dof] : orc_oracle_chi2_quantile(dof, chi2_prob);
        gamma[j] = g;
        if (H_all) memcpy(&H_all[IDX(all_rows, 0, n)], Hxj, (size_t)mo * n * sizeof(double));
        if (r_all) memcpy(&r_all[all_rows], rj, mo * sizeof(double));
        all_rows += mo;
        if (block_ptr) block_ptr[j + 1] = all_rows;
        if (g < thr) {                                                        /* :1970 */
            accept[j] = 1;
            memcpy(&Hs[IDX(stack, 0, n)], Hxj, (size_t)mo * n * sizeof(double));
            memcpy(&rs[stack], rj, mo * sizeof(double));
            stack += mo;
        }
    }
    int mt = orc_oracle_measurement_update(Hs, rs, stack, n, P, sigma2, dx, P_out, H_thin, r_thin, K, G);
    if (info) { info[0] = stack; info[1] = mt; }
    free(Hs); free(rs); free(Hxj); free(Hfj); free(rj);
    return mt < 0 ? -1 : 0;
}


/* ---- helpers shared with object_oracle.c (exported under the oracle's prefix) ------------------------------------ */
void orc_oracle_house_qr(double* A, int m, int n, int lda, double* beta) { house_qr(A, m, n, lda, beta); }
void orc_oracle_house_apply_qt(const double* A, int m, int n, int lda, const double* beta, double* C, int nc, int ldc) {
    house_apply_qt(A, m, n, lda, beta, C, nc, ldc);
}
int orc_oracle_chol_lower(double* S, int n) { return chol_lower(S, n); }
void orc_oracle_cam_wrt_imu(const double* R_b2c, const double* t_c_b, const double* R_w2c, const double* t_b_w, int left, double J[36]) {
    cam_wrt_imu(R_b2c, t_c_b, R_w2c, t_b_w, left, J);
}

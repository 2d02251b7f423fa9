/*
 * msckf_fast.c -- TEST / BENCH INFRASTRUCTURE ONLY: a best-effort CPU implementation of the same MSCKF update, used
 * (a) by bench.py as the "all host cores" CPU baseline beside the literal single-threaded port (msckf_oracle.c), and
 * (b) by tests/test_oracle.py as a third, independently written evaluation of the update (different nullspace
 * construction, different compression, different solve) that must agree with the literal restatement.
 *
 * Same results as the reference's update (src/orcvio.cpp:1171-1226, 1953-1976, 2497-2560, 1654-1763), minimum work:
 *   per track (OpenMP over tracks, one thread each)
 *     - the 2M x (7 + 6M) block over the columns the track touches (extrinsics 6, td 1, its M clones) -- not n wide
 *     - three Householder reflectors of H_f (2M x 3) applied to [block | r]; rows 3.. are the projected block
 *       (the reference forms the full 2M x 2M U of an SVD and a dense A^T H_x product)
 *     - the gate on those columns only: S = H' P_sub H'^T + s2 I, Cholesky, gamma  (the reference multiplies n-wide)
 *     - accepted: the block's Gram is added into a per-thread (NA+1)^2 accumulator [H'^T H', H'^T r'; ., r'^T r']
 *   then, once: the sum of the accumulators, and the Kalman solve in square-root form
 *       P = L L^T, M = s2 I + L_a^T A L_a, delta_x = L M^-1 L_a^T b, P+ = s2 L M^-1 L^T
 *     (= K r and (I - K H) P; no rank decision on the singular Gram block), dense loops parallelised over rows.
 * Nothing under orcvio_amd/ may link or load this file.  Parity unpinned like msckf_oracle.c (same header applies).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_oracle_measurement_jacobian(const int* flags, const double* R_b2w, const double* t_b_w, const double* t_fej,
                                     const double* R_b2c, const double* t_c_b, const double* p_w, const double* z,
                                     double* H_x, double* H_e, double* H_f, double* r);
double orc_oracle_chi2_quantile(int dof, double p);

#define MAXM 64

/* in-place Cholesky (lower) of an SPD matrix with leading dimension ld; zero pivots (<= tol) drop their column */
static int chol_psd(double* S, int n, int ld, double tol) {
    int dropped = 0;
    for (int j = 0; j < n; ++j) {
        double d = S[j * ld + j];
        for (int k = 0; k < j; ++k) d -= S[j * ld + k] * S[j * ld + k];
        if (!(d > tol)) {
            if (d != d) return -1;
            for (int i = j; i < n; ++i) S[i * ld + j] = 0.0;
            ++dropped;
            continue;
        }
        const double l = sqrt(d);
        S[j * ld + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double s = S[i * ld + j];
            for (int k = 0; k < j; ++k) s -= S[i * ld + k] * S[j * ld + k];
            S[i * ld + j] = s / l;
        }
    }
    return dropped;
}

int orc_fast_msckf_update(int N, int F, const int* flags, double sigma, double chi2_prob, const double* chi2_table, int chi2_table_len,
                          const double* R_b2w, const double* t_b_w, const double* t_fej, const double* R_b2c, const double* t_c_b,
                          const double* p_w, const int* obs_ptr, const int* obs_clone, const double* obs_z, const double* obs_zvel,
                          const double* P, double* dx, double* P_out, int* accept, double* gamma, int* threads_used) {
    const int leg = flags[0], n = leg + 6 * N, NA = n - 15, W = NA + 1;
    const double s2 = sigma * sigma;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    if (threads_used) *threads_used = nthreads;
    double* Aacc = (double*)calloc((size_t)nthreads * W * W, sizeof(double));
    int bad = 0;
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double* A = Aacc + (size_t)tid * W * W;
        const int CM = 7 + 6 * MAXM;
        double* X = (double*)malloc((size_t)2 * MAXM * (CM + 1) * sizeof(double));     /* [block | r] */
        double* HP = (double*)malloc((size_t)2 * MAXM * CM * sizeof(double));
        double* S = (double*)malloc((size_t)4 * MAXM * MAXM * sizeof(double));
        int* col = (int*)malloc((size_t)CM * sizeof(int));
#pragma omp for schedule(dynamic, 4)
        for (int j = 0; j < F; ++j) {
            const int o0 = obs_ptr[j], M = obs_ptr[j + 1] - o0;
            accept[j] = 0;
            gamma[j] = NAN;
            if (M < 2 || M > MAXM) continue;
            const int rows = 2 * M, nc = 7 + 6 * M, ld = nc + 1;
            double Hf[2 * MAXM * 3];
            memset(X, 0, (size_t)rows * ld * sizeof(double));
            for (int c = 0; c < 7; ++c) col[c] = 15 + c;                                /* extrinsics 15..20, td 21 */
            for (int k = 0; k < M; ++k) {
                const int i = obs_clone[o0 + k];
                double Hx[12], He[12], hf[6], r2[2];
                orc_oracle_measurement_jacobian(flags, &R_b2w[9 * i], &t_b_w[3 * i], &t_fej[3 * i], &R_b2c[9 * i], &t_c_b[3 * i],
                                                &p_w[3 * j], &obs_z[2 * (o0 + k)], Hx, He, hf, r2);
                for (int e = 0; e < 6; ++e) col[7 + 6 * k + e] = leg + 6 * i + e;
                for (int a = 0; a < 2; ++a) {
                    double* row = X + (size_t)(2 * k + a) * ld;
                    for (int e = 0; e < 6; ++e) { row[e] = He[a * 6 + e]; row[7 + 6 * k + e] = Hx[a * 6 + e]; }
                    if (flags[4]) row[6] = obs_zvel[2 * (o0 + k) + a];
                    row[nc] = r2[a];
                    for (int e = 0; e < 3; ++e) Hf[(2 * k + a) * 3 + e] = hf[a * 3 + e];
                }
            }
            /* three reflectors of H_f, applied to [block | r] */
            for (int q = 0; q < 3 && q < rows; ++q) {
                double nrm2 = 0.0;
                for (int i = q + 1; i < rows; ++i) nrm2 += Hf[i * 3 + q] * Hf[i * 3 + q];
                if (nrm2 == 0.0) continue;
                const double alpha = Hf[q * 3 + q], nu = sqrt(alpha * alpha + nrm2), beta = alpha >= 0.0 ? -nu : nu;
                const double tau = (beta - alpha) / beta, sc = 1.0 / (alpha - beta);
                double v[2 * MAXM];
                v[q] = 1.0;
                for (int i = q + 1; i < rows; ++i) v[i] = Hf[i * 3 + q] * sc;
                for (int c = q + 1; c < 3; ++c) {
                    double w = 0.0;
                    for (int i = q; i < rows; ++i) w += v[i] * Hf[i * 3 + c];
                    w *= tau;
                    for (int i = q; i < rows; ++i) Hf[i * 3 + c] -= w * v[i];
                }
                for (int c = 0; c <= nc; ++c) {
                    double w = 0.0;
                    for (int i = q; i < rows; ++i) w += v[i] * X[(size_t)i * ld + c];
                    w *= tau;
                    if (w != 0.0)
                        for (int i = q; i < rows; ++i) X[(size_t)i * ld + c] -= w * v[i];
                }
            }
            const int m = rows - 3;                         /* projected rows 3 .. rows-1 (rows <= 3: nothing to project) */
            if (m <= 0) continue;
            const double* Hp = X + (size_t)3 * ld;
            /* gate: S = H' P_sub H'^T + s2 I on the nc touched columns */
            for (int i = 0; i < m; ++i)
                for (int c = 0; c < nc; ++c) {
                    double s = 0.0;
                    const double* prow = P + (size_t)col[c] * n;      /* P symmetric: row col[c] */
                    for (int k = 0; k < nc; ++k) s += Hp[(size_t)i * ld + k] * prow[col[k]];
                    HP[(size_t)i * nc + c] = s;
                }
            for (int i = 0; i < m; ++i)
                for (int k = 0; k <= i; ++k) {
                    double s = (i == k) ? s2 : 0.0;
                    for (int c = 0; c < nc; ++c) s += HP[(size_t)i * nc + c] * Hp[(size_t)k * ld + c];
                    S[i * m + k] = s;
                }
            if (chol_psd(S, m, m, 0.0) != 0) {
#pragma omp atomic write
                bad = 1;
                continue;
            }
            double g = 0.0, y[2 * MAXM];
            for (int i = 0; i < m; ++i) {
                double s = Hp[(size_t)i * ld + nc];
                for (int k = 0; k < i; ++k) s -= S[i * m + k] * y[k];
                y[i] = s / S[i * m + i];
                g += y[i] * y[i];
            }
            gamma[j] = g;
            const int dof = 2 * M - 3;
            const double thr = dof < chi2_table_len ? chi2_table[dof] : orc_oracle_chi2_quantile(dof, chi2_prob);
            if (!(g < thr)) continue;
            accept[j] = 1;
            /* Gram of the accepted block into this thread's accumulator (lower triangle; index NA = the residual) */
            for (int a = 0; a <= nc; ++a) {
                const int ia = a < nc ? col[a] - 15 : NA;
                for (int b = 0; b <= a; ++b) {
                    const int ib = b < nc ? col[b] - 15 : NA;
                    double s = 0.0;
                    for (int i = 0; i < m; ++i) s += Hp[(size_t)i * ld + a] * Hp[(size_t)i * ld + b];
                    if (ia >= ib) A[(size_t)ia * W + ib] += s; else A[(size_t)ib * W + ia] += s;
                }
            }
        }
        free(X); free(HP); free(S); free(col);
    }
    if (bad) { free(Aacc); return -1; }
    /* A = sum of the accumulators (thread order: deterministic for a fixed thread count), mirrored */
    double* A = (double*)calloc((size_t)W * W, sizeof(double));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < W; ++i)
        for (int k = 0; k <= i; ++k) {
            double s = 0.0;
            for (int t = 0; t < nthreads; ++t) s += Aacc[(size_t)t * W * W + (size_t)i * W + k];
            A[(size_t)i * W + k] = s;
        }
    for (int i = 0; i < W; ++i)
        for (int k = i + 1; k < W; ++k) A[(size_t)i * W + k] = A[(size_t)k * W + i];
    free(Aacc);
    /* ---- square-root Kalman solve ------------------------------------------------------------------------------- */
    double* L = (double*)malloc((size_t)n * n * sizeof(double));
    memcpy(L, P, (size_t)n * n * sizeof(double));
    double mx = 0.0;
    for (int i = 0; i < n; ++i) if (P[(size_t)i * n + i] > mx) mx = P[(size_t)i * n + i];
    if (chol_psd(L, n, n, 8.0 * 2.220446049250313e-16 * mx) < 0) { free(A); free(L); return -1; }
    for (int i = 0; i < n; ++i)
        for (int k = i + 1; k < n; ++k) L[(size_t)i * n + k] = 0.0;
    /* U = A[0:NA,0:NA] L_a  (NA x n),  L_a = L[15:, :] ;  g = L_a^T b */
    double* U = (double*)calloc((size_t)NA * n, sizeof(double));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < NA; ++i)
        for (int k = 0; k < NA; ++k) {
            const double a = A[(size_t)i * W + k];
            if (a == 0.0) continue;
            const double* lrow = L + (size_t)(15 + k) * n;
            double* urow = U + (size_t)i * n;
            for (int c = 0; c <= 15 + k && c < n; ++c) urow[c] += a * lrow[c];
        }
    double* Mm = (double*)calloc((size_t)n * n, sizeof(double));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < NA; ++k) {
            const double l = L[(size_t)(15 + k) * n + i];
            if (l == 0.0) continue;
            const double* urow = U + (size_t)k * n;
            double* mrow = Mm + (size_t)i * n;
            for (int c = 0; c <= i; ++c) mrow[c] += l * urow[c];
        }
    for (int i = 0; i < n; ++i) Mm[(size_t)i * n + i] += s2;
    if (chol_psd(Mm, n, n, 0.0) != 0) { free(A); free(L); free(U); free(Mm); return -1; }
    /* Z = L_M^-1 [L^T | g]  (n x (n+1)) */
    double* Z = (double*)malloc((size_t)n * (n + 1) * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int c = 0; c <= n; ++c) {
        for (int i = 0; i < n; ++i) {
            double s;
            if (c < n) s = L[(size_t)c * n + i];
            else {
                s = 0.0;
                for (int k = 0; k < NA; ++k) s += L[(size_t)(15 + k) * n + i] * A[(size_t)k * W + NA];
            }
            for (int k = 0; k < i; ++k) s -= Mm[(size_t)i * n + k] * Z[(size_t)k * (n + 1) + c];
            Z[(size_t)i * (n + 1) + c] = s / Mm[(size_t)i * n + i];
        }
    }
#pragma omp parallel for schedule(static)
    for (int a = 0; a < n; ++a) {
        for (int b = 0; b <= a; ++b) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += Z[(size_t)i * (n + 1) + a] * Z[(size_t)i * (n + 1) + b];
            P_out[(size_t)a * n + b] = s2 * s;
            P_out[(size_t)b * n + a] = s2 * s;
        }
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += Z[(size_t)i * (n + 1) + a] * Z[(size_t)i * (n + 1) + n];
        dx[a] = s;
    }
    free(A); free(L); free(U); free(Mm); free(Z);
    return 0;
}

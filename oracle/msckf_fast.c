/*
 * msckf_fast.c -- TEST / BENCH INFRASTRUCTURE ONLY: a best-effort CPU implementation of the same MSCKF update, used
 * (a) by bench.py as the "all host cores" CPU baseline beside the literal single-threaded port (msckf_oracle.c), and
 * (b) by tests/test_oracle.py as a third, independently written evaluation of the update (different nullspace
 * construction, different compression, different solve) that must agree with the literal restatement.
 *
 * Same results as the reference's update (src/orcvio.cpp:1171-1226, 1953-1976, 2497-2560, 1654-1763), minimum work --
 * the sparsity the device path exploits, so that the CPU figure beside it is not a strawman:
 *   per track (OpenMP over tracks, one thread each)
 *     - the 2M un-projected rows X = [H_e | zvel | H_x(clone) | r]: 13 Jacobian non-zeros each
 *     - E = X P X^T (2M x 2M) from those non-zeros: u_l = X_l P restricted to the columns the track touches (7 + 6 M), then
 *       13 products per entry -- 0.35 MFLOP at M = 30 where the dense H' P H'^T of the reference is 5 MFLOP
 *     - the three Householder reflectors of H_f (2M x 3) applied to E + s2 I from both sides and to r; the trailing (2M-3)^2
 *       block is the gate's S, rows 3.. of Q^T r its residual; Cholesky, gamma
 *     - accepted: [H'^T H', H'^T r'] = X^T X - T3^T T3 with T3 = the first three rows of Q^T [X | r] (Q orthogonal): a sparse
 *       outer-product pass (14 non-zeros per row) and a three-row dense Gram into a per-thread (NA+1)^2 accumulator
 *   then, once: the sum of the accumulators, and the Kalman solve in square-root form
 *       P = L L^T, M = s2 I + L_a^T A L_a, delta_x = L M^-1 L_a^T b, P+ = s2 L M^-1 L^T
 *     (= K r and (I - K H) P; no rank decision on the singular Gram block), dense loops parallelised over rows / right-hand sides.
 * orc_fast_set_threads(t) bounds the OpenMP team (0 = the runtime's default): bench.py sweeps it and reports the best.
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

/* The Kalman solve in square-root form on a compressed block A ((NA+1)^2, symmetric, index NA = the residual column):
 * P = L L^T, M = s2 I + L_a^T A L_a, delta_x = L M^-1 L_a^T b, P+ = s2 L M^-1 L^T; *zg2 = |L_M^-1 L_a^T b|^2 (what the joint gate of
 * the object update needs: gamma = (|r'|^2 - zg2) / s2).  Dense loops parallelised over rows / right-hand sides (nt2 threads).
 * Shared by orc_fast_msckf_update and orc_fast_objects_update (oracle/object_fast.c). */
int orc_fast_sqrt_solve(int n, int NA, const double* A, const double* P, double s2, double* dx, double* P_out, double* zg2, int nt2) {
    const int W = NA + 1;
    /* ---- square-root Kalman solve ------------------------------------------------------------------------------- */
    double* L = (double*)malloc((size_t)n * n * sizeof(double));
    memcpy(L, P, (size_t)n * n * sizeof(double));
    double mx = 0.0;
    for (int i = 0; i < n; ++i) if (P[(size_t)i * n + i] > mx) mx = P[(size_t)i * n + i];
    if (chol_psd(L, n, n, 8.0 * 2.220446049250313e-16 * mx) < 0) { free(L); return -1; }
    for (int i = 0; i < n; ++i)
        for (int k = i + 1; k < n; ++k) L[(size_t)i * n + k] = 0.0;
    /* U = A[0:NA,0:NA] L_a  (NA x n),  L_a = L[15:, :] ;  g = L_a^T b */
    double* U = (double*)calloc((size_t)NA * n, sizeof(double));
#pragma omp parallel for schedule(static) num_threads(nt2)
    for (int i = 0; i < NA; ++i)
        for (int k = 0; k < NA; ++k) {
            const double a = A[(size_t)i * W + k];
            if (a == 0.0) continue;
            const double* lrow = L + (size_t)(15 + k) * n;
            double* urow = U + (size_t)i * n;
            for (int c = 0; c <= 15 + k && c < n; ++c) urow[c] += a * lrow[c];
        }
    double* Mm = (double*)calloc((size_t)n * n, sizeof(double));
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt2)
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < NA; ++k) {
            const double l = L[(size_t)(15 + k) * n + i];
            if (l == 0.0) continue;
            const double* urow = U + (size_t)k * n;
            double* mrow = Mm + (size_t)i * n;
            for (int c = 0; c <= i; ++c) mrow[c] += l * urow[c];
        }
    for (int i = 0; i < n; ++i) Mm[(size_t)i * n + i] += s2;
    if (chol_psd(Mm, n, n, 0.0) != 0) { free(L); free(U); free(Mm); return -1; }
    /* Z^T = (L_M^-1 [L^T | g])^T, one right-hand side per row ((n+1) x n: both operands of the substitution contiguous) */
    double* Z = (double*)malloc((size_t)(n + 1) * n * sizeof(double));
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt2)
    for (int c = 0; c <= n; ++c) {
        double* z = Z + (size_t)c * n;
        for (int i = 0; i < n; ++i) {
            double s;
            if (c < n) s = L[(size_t)c * n + i];
            else {
                s = 0.0;
                for (int k = 0; k < NA; ++k) s += L[(size_t)(15 + k) * n + i] * A[(size_t)k * W + NA];
            }
            const double* mrow = Mm + (size_t)i * n;
            for (int k = 0; k < i; ++k) s -= mrow[k] * z[k];
            z[i] = s / mrow[i];
        }
    }
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt2)
    for (int a = 0; a < n; ++a) {
        const double* za = Z + (size_t)a * n;
        for (int b = 0; b <= a; ++b) {
            const double* zb = Z + (size_t)b * n;
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += za[i] * zb[i];
            P_out[(size_t)a * n + b] = s2 * s;
            P_out[(size_t)b * n + a] = s2 * s;
        }
        const double* zg = Z + (size_t)n * n;
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += za[i] * zg[i];
        dx[a] = s;
    }
    if (zg2) {
        const double* zg = Z + (size_t)n * n;
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += zg[i] * zg[i];
        *zg2 = s;
    }
    free(L); free(U); free(Mm); free(Z);
    return 0;
}

static int g_fast_threads = 0;
void orc_fast_set_threads(int t) { g_fast_threads = t > 0 ? t : 0; }

int orc_fast_msckf_update(int N, int F, const int* flags, double sigma, double chi2_prob, const double* chi2_table, int chi2_table_len,
                          const double* R_b2w, const double* t_b_w, const double* t_fej, const double* R_b2c, const double* t_c_b,
                          const double* p_w, const int* obs_ptr, const int* obs_clone, const double* obs_z, const double* obs_zvel,
                          const double* P, double* dx, double* P_out, int* accept, double* gamma, int* threads_used) {
    const int leg = flags[0], n = leg + 6 * N, NA = n - 15, W = NA + 1;
    const double s2 = sigma * sigma;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = g_fast_threads > 0 ? g_fast_threads : omp_get_max_threads();
    if (nthreads > F && F > 0) nthreads = F;
#endif
    if (threads_used) *threads_used = nthreads;
    double* Aacc = (double*)calloc((size_t)nthreads * W * W, sizeof(double));
    int bad = 0;
#pragma omp parallel num_threads(nthreads)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double* A = Aacc + (size_t)tid * W * W;
        const int CM = 7 + 6 * MAXM;                    /* track-local columns: [ext 6 | td | clone block of observation 0, 1, ...] */
        double* Xe = (double*)malloc((size_t)2 * MAXM * 7 * sizeof(double));      /* rows x [H_e | zvel]          */
        double* Xc = (double*)malloc((size_t)2 * MAXM * 6 * sizeof(double));      /* rows x H_x of the row's clone */
        double* U = (double*)malloc((size_t)2 * CM * sizeof(double));             /* u of the two rows of one observation */
        double* S = (double*)malloc((size_t)4 * MAXM * MAXM * sizeof(double));
        double* T3 = (double*)malloc((size_t)3 * W * sizeof(double));
        int* ci = (int*)malloc((size_t)MAXM * sizeof(int));
#pragma omp for schedule(dynamic, 2)
        for (int j = 0; j < F; ++j) {
            const int o0 = obs_ptr[j], M = obs_ptr[j + 1] - o0;
            accept[j] = 0;
            gamma[j] = NAN;
            if (M < 2 || M > MAXM) continue;
            const int rows = 2 * M;
            double Hf[2 * MAXM * 3], rv[2 * MAXM];
            for (int k = 0; k < M; ++k) {
                const int i = obs_clone[o0 + k];
                double Hx[12], He[12], hf[6], r2[2];
                orc_oracle_measurement_jacobian(flags, &R_b2w[9 * i], &t_b_w[3 * i], &t_fej[3 * i], &R_b2c[9 * i], &t_c_b[3 * i],
                                                &p_w[3 * j], &obs_z[2 * (o0 + k)], Hx, He, hf, r2);
                ci[k] = i;
                for (int a = 0; a < 2; ++a) {
                    const int row = 2 * k + a;
                    for (int e = 0; e < 6; ++e) { Xe[row * 7 + e] = He[a * 6 + e]; Xc[row * 6 + e] = Hx[a * 6 + e]; }
                    Xe[row * 7 + 6] = flags[4] ? obs_zvel[2 * (o0 + k) + a] : 0.0;
                    rv[row] = r2[a];
                    for (int e = 0; e < 3; ++e) Hf[row * 3 + e] = hf[a * 3 + e];
                }
            }
            /* E + s2 I, lower triangle by observation pairs: u_l = X_l P on the track's columns, E(k, l) = X_k u_l^T (k <= l) */
            for (int l = 0; l < M; ++l) {
                for (int a = 0; a < 2; ++a) {
                    const int row = 2 * l + a;
                    double* u = U + (size_t)a * CM;
                    const double* pr[13];
                    double cf[13];
                    for (int e = 0; e < 7; ++e) { pr[e] = P + (size_t)(15 + e) * n; cf[e] = Xe[row * 7 + e]; }
                    for (int e = 0; e < 6; ++e) { pr[7 + e] = P + (size_t)(leg + 6 * ci[l] + e) * n; cf[7 + e] = Xc[row * 6 + e]; }
                    for (int c = 0; c < 7; ++c) {
                        double s = 0.0;
                        for (int e = 0; e < 13; ++e) s += cf[e] * pr[e][15 + c];
                        u[c] = s;
                    }
                    for (int k = 0; k <= l; ++k) {      /* (columns of the clones of observations k <= l are all E's lower triangle needs) */
                        const int c0 = leg + 6 * ci[k];
                        double acc[6] = {0, 0, 0, 0, 0, 0};
                        for (int e = 0; e < 13; ++e) {
                            const double f = cf[e];
                            const double* q = pr[e] + c0;
                            for (int c = 0; c < 6; ++c) acc[c] += f * q[c];
                        }
                        for (int c = 0; c < 6; ++c) u[7 + 6 * k + c] = acc[c];
                    }
                }
                for (int k = 0; k <= l; ++k)
                    for (int b = 0; b < 2; ++b) {
                        const int rk = 2 * k + b;
                        for (int a = 0; a < 2; ++a) {
                            const double* u = U + (size_t)a * CM;
                            double s = 0.0;
                            for (int e = 0; e < 7; ++e) s += Xe[rk * 7 + e] * u[e];
                            for (int e = 0; e < 6; ++e) s += Xc[rk * 6 + e] * u[7 + 6 * k + e];
                            S[(size_t)(2 * l + a) * rows + rk] = s;     /* entry (row of l, row of k): k <= l */
                        }
                    }
            }
            for (int i = 0; i < rows; ++i) {
                for (int k = i + 1; k < rows; ++k) S[(size_t)i * rows + k] = S[(size_t)k * rows + i];
                S[(size_t)i * rows + i] += s2;
            }
            /* three reflectors of H_f: S <- H S H, r <- H r; the vectors are kept for Q1 */
            double V[3][2 * MAXM], tau[3];
            int nref = 0;
            for (int q = 0; q < 3 && q < rows; ++q) {
                double nrm2 = 0.0;
                tau[q] = 0.0;
                for (int i = 0; i < rows; ++i) V[q][i] = 0.0;
                ++nref;
                for (int i = q + 1; i < rows; ++i) nrm2 += Hf[i * 3 + q] * Hf[i * 3 + q];
                if (nrm2 == 0.0) continue;
                const double alpha = Hf[q * 3 + q], nu = sqrt(alpha * alpha + nrm2), beta = alpha >= 0.0 ? -nu : nu;
                const double sc = 1.0 / (alpha - beta);
                double* v = V[q];
                tau[q] = (beta - alpha) / beta;
                v[q] = 1.0;
                for (int i = q + 1; i < rows; ++i) v[i] = Hf[i * 3 + q] * sc;
                for (int c = q + 1; c < 3; ++c) {
                    double w = 0.0;
                    for (int i = q; i < rows; ++i) w += v[i] * Hf[i * 3 + c];
                    w *= tau[q];
                    for (int i = q; i < rows; ++i) Hf[i * 3 + c] -= w * v[i];
                }
                double wv[2 * MAXM];
                for (int c = 0; c < rows; ++c) {           /* left: S -= tau v (v^T S) */
                    double w = 0.0;
                    for (int i = q; i < rows; ++i) w += v[i] * S[(size_t)i * rows + c];
                    wv[c] = tau[q] * w;
                }
                for (int i = q; i < rows; ++i)
                    for (int c = 0; c < rows; ++c) S[(size_t)i * rows + c] -= v[i] * wv[c];
                for (int i = 0; i < rows; ++i) {           /* right: S -= tau (S v) v^T */
                    double w = 0.0;
                    for (int c = q; c < rows; ++c) w += S[(size_t)i * rows + c] * v[c];
                    w *= tau[q];
                    for (int c = q; c < rows; ++c) S[(size_t)i * rows + c] -= w * v[c];
                }
                double w = 0.0;
                for (int i = q; i < rows; ++i) w += v[i] * rv[i];
                w *= tau[q];
                for (int i = q; i < rows; ++i) rv[i] -= w * v[i];
            }
            const int m = rows - 3;                         /* projected rows 3 .. rows-1 (rows <= 3: nothing to project) */
            if (m <= 0) continue;
            double* Sp = S + (size_t)3 * rows + 3;          /* trailing block, leading dimension rows */
            for (int i = 0; i < m; ++i)                     /* (symmetrise what rounding left of the two-sided products) */
                for (int k = 0; k < i; ++k) Sp[(size_t)i * rows + k] = 0.5 * (Sp[(size_t)i * rows + k] + Sp[(size_t)k * rows + i]);
            if (chol_psd(Sp, m, rows, 0.0) != 0) {
#pragma omp atomic write
                bad = 1;
                continue;
            }
            double g = 0.0, y[2 * MAXM];
            for (int i = 0; i < m; ++i) {
                double s = rv[3 + i];
                for (int k = 0; k < i; ++k) s -= Sp[(size_t)i * rows + k] * y[k];
                y[i] = s / Sp[(size_t)i * rows + i];
                g += y[i] * y[i];
            }
            gamma[j] = g;
            const int dof = 2 * M - 3;
            const double thr = dof < chi2_table_len ? chi2_table[dof] : orc_oracle_chi2_quantile(dof, chi2_prob);
            if (!(g < thr)) continue;
            accept[j] = 1;
            /* Gram of the accepted block: X^T X - T3^T T3 (lower triangle of the accumulator; index NA = the residual).  The
             * residual entering X is the UN-projected one: recover it (rv was reflected) from the Jacobian call's values */
            double Q1[2 * MAXM][3];                         /* Q e_c = H_1 H_2 H_3 e_c */
            for (int c = 0; c < 3; ++c) {
                double e[2 * MAXM];
                for (int i = 0; i < rows; ++i) e[i] = i == c ? 1.0 : 0.0;
                for (int q = nref - 1; q >= 0; --q) {
                    if (tau[q] == 0.0) continue;
                    double w = 0.0;
                    for (int i = q; i < rows; ++i) w += V[q][i] * e[i];
                    w *= tau[q];
                    for (int i = q; i < rows; ++i) e[i] -= w * V[q][i];
                }
                for (int i = 0; i < rows; ++i) Q1[i][c] = e[i];
            }
            /* r = Q (Q^T r): the un-projected residual from the reflected one */
            double r0[2 * MAXM];
            for (int i = 0; i < rows; ++i) r0[i] = rv[i];
            for (int q = nref - 1; q >= 0; --q) {
                if (tau[q] == 0.0) continue;
                double w = 0.0;
                for (int i = q; i < rows; ++i) w += V[q][i] * r0[i];
                w *= tau[q];
                for (int i = q; i < rows; ++i) r0[i] -= w * V[q][i];
            }
            const int ne = leg - 15;                         /* ext columns of the active block: 6 extrinsics (+ td) */
            int lo = W, hi = 0;                              /* T3 is non-zero on the ext columns, the track's clones and NA */
            for (int k = 0; k < M; ++k) { const int c0 = leg - 15 + 6 * ci[k]; if (c0 < lo) lo = c0; if (c0 + 6 > hi) hi = c0 + 6; }
            for (int c = 0; c < 3; ++c) {
                double* t3 = T3 + (size_t)c * W;
                for (int a = 0; a < ne; ++a) t3[a] = 0.0;
                for (int a = lo; a < hi; ++a) t3[a] = 0.0;
                t3[NA] = 0.0;
            }
            for (int row = 0; row < rows; ++row) {
                int idx[14];
                double val[14];
                const int c0 = leg - 15 + 6 * ci[row >> 1];
                for (int e = 0; e < 7; ++e) { idx[e] = e; val[e] = Xe[row * 7 + e]; }
                for (int e = 0; e < 6; ++e) { idx[7 + e] = c0 + e; val[7 + e] = Xc[row * 6 + e]; }
                idx[13] = NA; val[13] = r0[row];
                for (int a = 0; a < 14; ++a) {
                    const double va = val[a];
                    if (va == 0.0) continue;
                    for (int b = 0; b < 14; ++b) {
                        if (idx[b] > idx[a]) continue;     /* lower triangle: (idx[a], idx[b]) with idx[b] <= idx[a] */
                        A[(size_t)idx[a] * W + idx[b]] += va * val[b];
                    }
                    for (int c = 0; c < 3; ++c) T3[(size_t)c * W + idx[a]] += Q1[row][c] * va;
                }
            }
            for (int c = 0; c < 3; ++c) {
                const double* t3 = T3 + (size_t)c * W;
                for (int a = 0; a < ne; ++a) {
                    double* arow = A + (size_t)a * W;
                    for (int b = 0; b <= a; ++b) arow[b] -= t3[a] * t3[b];
                }
                for (int a = lo; a < hi; ++a) {
                    double* arow = A + (size_t)a * W;
                    const double ta = t3[a];
                    for (int b = 0; b < ne; ++b) arow[b] -= ta * t3[b];
                    for (int b = lo; b <= a; ++b) arow[b] -= ta * t3[b];
                }
                {
                    double* arow = A + (size_t)NA * W;
                    const double ta = t3[NA];
                    for (int b = 0; b < ne; ++b) arow[b] -= ta * t3[b];
                    for (int b = lo; b < hi; ++b) arow[b] -= ta * t3[b];
                    arow[NA] -= ta * ta;
                }
            }
        }
        free(Xe); free(Xc); free(U); free(S); free(T3); free(ci);
    }
    if (bad) { free(Aacc); return -1; }
    int nt2 = 1;
#ifdef _OPENMP
    nt2 = g_fast_threads > 0 ? g_fast_threads : omp_get_max_threads();
#endif
    /* A = sum of the accumulators (thread order: deterministic for a fixed thread count), mirrored */
    double* A = (double*)calloc((size_t)W * W, sizeof(double));
#pragma omp parallel for schedule(static) num_threads(nt2)
    for (int i = 0; i < W; ++i) {
        double* arow = A + (size_t)i * W;
        for (int t = 0; t < nthreads; ++t) {
            const double* src = Aacc + (size_t)t * W * W + (size_t)i * W;
            for (int k = 0; k <= i; ++k) arow[k] += src[k];
        }
    }
    for (int i = 0; i < W; ++i)
        for (int k = i + 1; k < W; ++k) A[(size_t)i * W + k] = A[(size_t)k * W + i];
    free(Aacc);
    {
        const int rc = orc_fast_sqrt_solve(n, NA, A, P, s2, dx, P_out, (double*)0, nt2);
        free(A);
        return rc;
    }
}

/*
 * object_fast.c -- TEST / BENCH INFRASTRUCTURE ONLY: a best-effort CPU implementation of the object update
 * (System::processObjects -> OrcVIO::removeLostObjects, ros_wrapper/src/orcvio/src/System.cpp:622-708, src/orcvio.cpp:2154-2193;
 * per-object projection, SURVEY note N3) from row blocks, used
 * (a) by bench.py as the "all host cores" CPU figure beside the literal single-threaded port (object_oracle.c), and
 * (b) by tests/test_oracle_objects.py as a second, independently written evaluation that must agree with the literal one.
 *
 * Minimum work -- the structure the device path exploits, so that the CPU figure beside it is not a strawman: every row of an
 * object touches ONE clone (6 Jacobian non-zeros + the residual), so with X = [H_x | r] and H_f = Q R
 *     H'^T H' = X^T X - Y^T Y,   Y = R^-T (H_f^T X)          (the left-nullspace projection as a Schur complement)
 * needs the QR of H_f (rows x 45), a sparse cross product, a triangular solve and a (45 x 181) Gram per object -- the literal port
 * applies the 45 reflectors to a dense rows x (n + 1) block and takes the QR of the stack.  Objects in parallel (OpenMP), one
 * (NA+1)^2 accumulator per thread; then the square-root Kalman solve of msckf_fast.c and the joint gate
 *     gamma = (|r'|^2 - |L_M^-1 L_a^T b|^2) / s2,   dof = sum (rows - columns)   (:2172-2176).
 * A pivot of R below 1e-11 of the largest is dropped (its row of Y is zero): projection onto the WHOLE left null space of a
 * rank-deficient H_f, the convention of the device path; for full-rank H_f this is the reference's projection.
 * Nothing under orcvio_amd/ may link or load this file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void orc_oracle_house_qr(double* A, int m, int n, int lda, double* beta);
double orc_oracle_chi2_quantile(int dof, double p);
int orc_fast_sqrt_solve(int n, int NA, const double* A, const double* P, double s2, double* dx, double* P_out, double* zg2, int nt2);

static int g_obj_threads = 0;
void orc_fast_objects_set_threads(int t) { g_obj_threads = t > 0 ? t : 0; }

int orc_fast_objects_update(int n_clones, int leg, int nobj, const int* row_ptr, const int* ncol, int ncol_max, const int* row_clone,
                            const double* Hx6, const double* Hf, const double* res, const double* P, double sigma, double chi2_prob,
                            int* accept, double* gamma, int* dof_out, double* dx, double* P_out, int* threads_used) {
    const int N = n_clones, n = leg + 6 * N, NA = n - 15, W = NA + 1, cb0 = leg - 15;
    const double s2 = sigma * sigma;
    int tot = 0;
    for (int o = 0; o < nobj; ++o) {
        const int m = row_ptr[o + 1] - row_ptr[o];
        if (m > ncol[o]) tot += m - ncol[o];
    }
    *accept = 0; *gamma = NAN; *dof_out = tot;
    memset(dx, 0, (size_t)n * sizeof(double));
    memcpy(P_out, P, (size_t)n * n * sizeof(double));
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = g_obj_threads > 0 ? g_obj_threads : omp_get_max_threads();
    if (nthreads > nobj && nobj > 0) nthreads = nobj;
#endif
    if (threads_used) *threads_used = nthreads;
    if (tot == 0) return 0;
    double* Aacc = (double*)calloc((size_t)nthreads * W * W, sizeof(double));
#pragma omp parallel num_threads(nthreads)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double* A = Aacc + (size_t)tid * W * W;
        int loc[64];                                   /* clone -> its position among the clones this object sees */
#pragma omp for schedule(dynamic, 1)
        for (int o = 0; o < nobj; ++o) {
            const int r0 = row_ptr[o], m = row_ptr[o + 1] - r0, nc = ncol[o];
            if (m <= nc) continue;                     /* nullspace_project_inplace_svd returns false */
            int K = 0, clones[64];
            for (int c = 0; c < N; ++c) loc[c] = -1;
            for (int i = 0; i < m; ++i) {
                const int c = row_clone[r0 + i];
                if (loc[c] < 0) { loc[c] = K; clones[K++] = c; }
            }
            const int w = 6 * K + 1;                   /* compact columns: six per clone seen, then the residual */
            double* C = (double*)calloc((size_t)nc * w, sizeof(double));      /* H_f^T X */
            double* R = (double*)malloc((size_t)m * nc * sizeof(double));     /* H_f, then its QR */
            double* beta = (double*)calloc((size_t)nc, sizeof(double));
            for (int i = 0; i < m; ++i) {
                const double* hf = Hf + (size_t)(r0 + i) * ncol_max;
                const double* hx = Hx6 + (size_t)(r0 + i) * 6;
                const double ri = res[r0 + i];
                const int c = row_clone[r0 + i], k0 = 6 * loc[c];
                memcpy(R + (size_t)i * nc, hf, (size_t)nc * sizeof(double));
                for (int j = 0; j < nc; ++j) {
                    const double h = hf[j];
                    if (h == 0.0) continue;
                    double* cj = C + (size_t)j * w;
                    for (int e = 0; e < 6; ++e) cj[k0 + e] += h * hx[e];
                    cj[w - 1] += h * ri;
                }
                /* X^T X: the row's 7 x 7 block straight into the accumulator (lower triangle; index NA = the residual) */
                const int a0 = cb0 + 6 * c;
                for (int e = 0; e < 6; ++e) {
                    for (int f = 0; f <= e; ++f) A[(size_t)(a0 + e) * W + a0 + f] += hx[e] * hx[f];
                    A[(size_t)NA * W + a0 + e] += ri * hx[e];
                }
                A[(size_t)NA * W + NA] += ri * ri;
            }
            orc_oracle_house_qr(R, m, nc, nc, beta);   /* R in the upper triangle */
            double dmax = 0.0;
            for (int k = 0; k < nc; ++k) if (fabs(R[(size_t)k * nc + k]) > dmax) dmax = fabs(R[(size_t)k * nc + k]);
            /* Y = R^-T C by forward substitution, row by row (R^T is lower triangular) */
            for (int k = 0; k < nc; ++k) {
                double* yk = C + (size_t)k * w;
                const double d = R[(size_t)k * nc + k];
                if (!(fabs(d) > 1e-11 * dmax)) { memset(yk, 0, (size_t)w * sizeof(double)); continue; }
                for (int j = 0; j < k; ++j) {
                    const double rjk = R[(size_t)j * nc + k];
                    if (rjk == 0.0) continue;
                    const double* yj = C + (size_t)j * w;
                    for (int q = 0; q < w; ++q) yk[q] -= rjk * yj[q];
                }
                const double inv = 1.0 / d;
                for (int q = 0; q < w; ++q) yk[q] *= inv;
            }
            /* A -= Y^T Y on the object's columns */
            for (int ka = 0; ka < K; ++ka)
                for (int ea = 0; ea < 6; ++ea) {
                    const int qa = 6 * ka + ea, ia = cb0 + 6 * clones[ka] + ea;
                    for (int kb = 0; kb < K; ++kb)
                        for (int eb = 0; eb < 6; ++eb) {
                            const int qb = 6 * kb + eb, ib = cb0 + 6 * clones[kb] + eb;
                            if (ib > ia) continue;
                            double s = 0.0;
                            for (int k = 0; k < nc; ++k) s += C[(size_t)k * w + qa] * C[(size_t)k * w + qb];
                            A[(size_t)ia * W + ib] -= s;
                        }
                    double s = 0.0;
                    for (int k = 0; k < nc; ++k) s += C[(size_t)k * w + qa] * C[(size_t)k * w + w - 1];
                    A[(size_t)NA * W + ia] -= s;
                }
            {
                double s = 0.0;
                for (int k = 0; k < nc; ++k) s += C[(size_t)k * w + w - 1] * C[(size_t)k * w + w - 1];
                A[(size_t)NA * W + NA] -= s;
            }
            free(C); free(R); free(beta);
        }
    }
    /* the sum of the accumulators (thread order: deterministic for a fixed team), mirrored */
    double* A = (double*)calloc((size_t)W * W, sizeof(double));
    for (int t = 0; t < nthreads; ++t) {
        const double* src = Aacc + (size_t)t * W * W;
        for (int i = 0; i < W; ++i)
            for (int k = 0; k <= i; ++k) A[(size_t)i * W + k] += src[(size_t)i * W + k];
    }
    for (int i = 0; i < W; ++i)
        for (int k = i + 1; k < W; ++k) A[(size_t)i * W + k] = A[(size_t)k * W + i];
    free(Aacc);
    const double rr = A[(size_t)NA * W + NA];
    double* dxs = (double*)malloc((size_t)n * sizeof(double));
    double* Ps = (double*)malloc((size_t)n * n * sizeof(double));
    double zg2 = 0.0;
    int nt2 = 1;
#ifdef _OPENMP
    nt2 = g_obj_threads > 0 ? g_obj_threads : omp_get_max_threads();
#endif
    int rc = orc_fast_sqrt_solve(n, NA, A, P, s2, dxs, Ps, &zg2, nt2);
    if (rc == 0) {
        const double g = (rr - zg2) / s2;
        *gamma = g;
        const double thr = orc_oracle_chi2_quantile(tot, chi2_prob);
        if (g == g && g < thr) {
            *accept = 1;
            memcpy(dx, dxs, (size_t)n * sizeof(double));
            memcpy(P_out, Ps, (size_t)n * n * sizeof(double));
        }
    }
    free(A); free(dxs); free(Ps);
    return rc;
}

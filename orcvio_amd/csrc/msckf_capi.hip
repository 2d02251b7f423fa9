// msckf_capi.hip -- C-ABI (include/orcvio_msckf.h) over the gfx950 kernels.
// No CPU fallback exists: every compute entry point fails with ORCVIO_ERR_NO_DEVICE /
// ORCVIO_ERR_HIP when the HIP runtime or a gfx950 device is missing.
#include "../../include/orcvio_msckf.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is loaded with dlopen (comm_* section), never linked
#include <dlfcn.h>
#include <chrono>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <functional>
#include "msckf_kernels.hpp"
#include "triangulate.hpp"
#include "cov_ops.hpp"
#include "ekf_rows.hpp"
#include "object_rows.hpp"
#include "io_ops.hpp"
#include <immintrin.h>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

using namespace orcvio_amd;

static thread_local std::string g_last_error;

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                \
            return ORCVIO_ERR_HIP;                                                           \
        }                                                                                    \
    } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static void so3_exp_decl(const double w[3], double R[9]);   // Sophus SO3d::exp (defined with incrementState_IMUCam below)

struct ObjUse { int t, row0, rows, ncol; size_t off_d, off_i; };   // a usable object track of the current object update

struct orcvio_msckf_handle {
    int device = 0;
    int maxN = 0, maxF = 0, maxObs = 0;
    int n_max = 0, NAP_max = 0, NP_max = 0;
    hipStream_t stream = nullptr;       // default launch stream
    hipStream_t side = nullptr;         // Cholesky of the prior, overlapped with the feature kernels
    hipEvent_t ev_fork = nullptr, ev_side = nullptr;
    // problem of the current upload
    orcvio_msckf_flags flags{};
    int N = 0, F = 0, nobs = 0, n = 0, NA = 0, NAP = 0, NP = 0, ldz = 0, m_tot = 0, Mmax = 0;
    int chunks = 0, rows_per_chunk = 0;
    bool uploaded = false, ran = false;
    bool reg_path = true;               // register-resident Cholesky (n <= 224), else the LDS-panel kernel
    bool fused_solve = true;            // chol(M) and the triangular solve in one launch (k_potrf_solve), reg path only
    int* d_skip = nullptr;              // [maxF] tracks dropped by triangulate_uploaded (nullptr semantics: skip_active)
    bool skip_active = false;
    // host <-> device traffic of the one-shot calls: small inputs share one device arena (one copy), everything is staged
    // through pinned host memory (a copy from pageable memory is synchronous and several times slower), the small
    // outputs share one arena (one copy back)
    // ONE input arena and ONE output arena, laid out compactly for the sizes of the current upload (layout_inputs /
    // layout_outputs) and mirrored in pinned host memory: an update moves one block in and one block out
    char *d_in = nullptr, *d_outs = nullptr;         // device arenas
    char* h_stage = nullptr;                          // pinned: [inputs | outputs]
    size_t in_cap = 0, outs_cap = 0, in_used = 0, outs_small = 0, stage_bytes = 0;
    size_t io_poses = 0, io_optr = 0, io_rptr = 0, io_cptr = 0, io_pw = 0, io_oclone = 0, io_cobs = 0, io_z = 0, io_zvel = 0, io_P = 0;
    size_t oo_dx = 0, oo_gamma = 0, oo_accept = 0, oo_Pout = 0;
    bool dl_pending = false, dl_with_P = false;      // a device -> host copy of the outputs is in flight on dl_stream
    hipStream_t dl_stream = nullptr;
    // graph policy: a launch graph is captured when a launch signature is seen for the SECOND time (not necessarily in a
    // row); a slot keeps up to GRAPH_WAYS captured graphs (least recently used one replaced).  More than one way, because
    // the signature contains every pointer a graph bakes in and some of them alternate: the resident square-root factor
    // is double-buffered (cov_commit / cov_prefactor / cov_augment swap d_Sres and d_Stmp), so a filter that repeats one
    // shape replays two graphs in turn (a single slot keyed without that pointer replayed a stale factor).
    static constexpr int GRAPH_WAYS = 6;
    struct GraphSlot {
        struct Way { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; unsigned long long sig = 0, used = 0; };
        Way way[GRAPH_WAYS];
        unsigned long long seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // signatures met once (no graph yet), ring
        int seen_at = 0;
        unsigned long long tick = 0;
    };
    GraphSlot g_update, g_local, g_finish, g_io;
    // ---- zero-copy boundary (orcvio_msckf_io_*): the pinned arena is device-visible and host-coherent, the first kernel of the
    // graph pulls the inputs out of it, the last one pushes the results into it and raises h_flag
    char* h_stage_dev = nullptr;              // device-visible address of h_stage
    unsigned long long* h_flag = nullptr;     // host-coherent word: publications so far (k_epilogue)
    unsigned long long* h_flag_dev = nullptr;
    unsigned long long* d_seq = nullptr;      // device-side twin (the value k_epilogue stores to the flag)
    int* d_pubcnt = nullptr;                  // arrival counter of k_epilogue's workgroups
    unsigned long long pub_enqueued = 0;      // publications enqueued so far (k_epilogue launches, stream order): the flag value to wait for
    bool io_open = false, io_with_P = false;  // orcvio_msckf_io_begin has laid the arena out and the caller is filling it
    double io_spin_seconds = 2.0;             // bound of the host's wait on h_flag (then: stream synchronisation, then ERR_TIMEOUT)
    bool last_sharded = false;                // the last finished update went through the handle's all-gather (status words in info[9..12])
    int shard_status = 0;                     // sharded calls: this rank's own status travelling with its block (ORCVIO_ERR_PEER)
    hipStream_t last_stream = nullptr;   // stream of the last run_update / run_finish (download waits for it)
    double *d_Pres = nullptr, *d_Ptmp = nullptr, *d_covT = nullptr;   // resident covariance, scratch, Phi*P rows
    // Resident SQUARE-ROOT FACTOR of the resident covariance: P_res = S S^T with S (fac_n x fac_k), stored like the
    // Cholesky factor of the prior (d_RP): S(i,j) = d_Sres[j * fac_ld + i].  cov_commit keeps S+ = sigma Z^T of the update
    // that has just run (Z = L_M^-1 S^T is what the solve produces anyway), so the NEXT update of the same frame
    // (pruneImuStateBuffer, processObjects: src/orcvio.cpp:591-594, System.cpp:551-555) skips the Cholesky of its prior.
    // cov_augment / cov_remove_clones carry it along (rows copied / deleted); cov_set / cov_propagate invalidate it.
    double *d_Sres = nullptr, *d_Stmp = nullptr;
    int fac_n = 0, fac_k = 0, fac_ld = 0;
    bool fac_valid = false;
    bool factor_opt = true;             // ORCVIO_OPT_RESIDENT_FACTOR
    int n_nui = 0;                      // ORCVIO_OPT_SCHMIDT_STATES: Schmidt nuisance states (6 columns each) at the END of the extra states
    bool obj_dof_rank = false;          // ORCVIO_OPT_OBJECT_DOF = 1: the object gate counts rows - rank(H_f) degrees of freedom (default: rows - columns, the reference's count)
    bool ref_h2_ldlt = false;           // ORCVIO_OPT_REF_H2_LDLT: the reference's literal H_2.ldlt() (diag(H_2)) in the tail of the hybrid update
    bool ref_stack_hf = false;          // ORCVIO_OPT_REF_STACK_HF: the reference's literal shared-Hf stacking of several objects
    bool arrow_opt = true;              // ORCVIO_OPT_OBJECT_QR: structured Householder QR of Hf (0: chol(Hf^T Hf), round 1's route)
    bool use_factor = false;            // the current upload's prior comes with its factor: no Cholesky of P
    // The prior's Cholesky is taken of the REVERSED matrix (potrf_reg_body, rev): P = S S^T with S(i, c) = L'(n-1-i, c), whose
    // rows 15 .. n-1 (the states measurement rows touch) are zero in the last 15 columns -- M = s2 I + L_a^T A L_a is
    // diag(M', s2 I_15) and only M' (kf - 15) is formed and factored.  `tail` = those trailing columns of the factor in use
    // (15, or 0: LDS-panel path, unfused solve, a resident factor that has been through an augmentation); fac_tail = the same
    // for the resident factor (kept by commit / remove_clones / clones_to_nuisance, dropped by augment).
    bool rev_prior_opt = true;          // ORCVIO_REV_PRIOR=0: the plain Cholesky (tail 0), same results
    int tail = 0, fac_tail = 0;
    int last_run_kind = 0;              // 0: run_update (single GPU), 1: run_local / run_finish (staged or sharded), 2: objects
    bool front_retry_forked = false;    // (download: the fused front end lost a hand-off; re-running on the forked path)
    int front_spin_limit = 1 << 19;    // polls before a workgroup of k_front gives up at the device-wide counter (tens of ms)
    int front_fallbacks = 0;            // how often that happened (orcvio_msckf_debug_read 'fallbacks')
    bool last_update_objects = false;   // the last finished update was a (gated) object update
    bool prior_forked = false;          // the Cholesky of the prior runs on the side stream (ev_side must be joined)
    int kf = 0;                         // columns of the prior's factor = dimension of M (n unless a resident factor is used)
    int* d_covmap = nullptr;
    int res_n = 0;                      // dimension of the resident covariance (0 = none)
    bool pw_missing = false;            // uploaded without positions: triangulate_uploaded must run before the update
    int *d_tri_valid = nullptr, *d_tri_flags = nullptr, *d_tri_init = nullptr;
    double *d_tri_sol = nullptr, *d_tri_cost = nullptr;
    int* d_flag = nullptr;              // step counter of that launch (inside the d_info allocation, own 128-byte line)
    // device buffers
    double *d_poses = nullptr, *d_pw = nullptr, *d_obs_z = nullptr, *d_obs_zvel = nullptr, *d_P = nullptr;
    int *d_obs_ptr = nullptr, *d_obs_clone = nullptr, *d_row_ptr = nullptr, *d_accept = nullptr, *d_info = nullptr;
    double *d_chi2 = nullptr, *d_Hs = nullptr, *d_gamma = nullptr, *d_Gpart = nullptr, *d_Ab = nullptr, *d_A = nullptr;
    int front_fused = 1;                // ORCVIO_OPT_FUSED_FRONT
    int n_extra = 0;                    // ORCVIO_OPT_EXTRA_STATES
    int ekf_mode = 0;                   // ORCVIO_OPT_EKF_ROWS: the extra states are active columns (EKF-SLAM rows may follow an upload)
    int ekf_F = 0, ekf_idp = 3;         // SLAM features of the current upload (orcvio_msckf_upload_ekf_rows)
    int ekf_cap = 0;
    int* d_ekf_i = nullptr;             // [3 cap] anchor | state | slot
    double* d_ekf_d = nullptr;          // [46 cap] H_e 12 | H_a 12 | H_x 12 | H_f 6 | z_vel 2 | r 2
    double* d_ekf_E = nullptr;          // [2 cap][NAP_max] dense accepted rows [H | r]
    double* d_Gekf = nullptr;           // [NAP_max^2] their Gram (lower tiles)
    double* d_ekf_gamma = nullptr; int* d_ekf_accept = nullptr;
    int dense_rows = 0, dense_cap = 0;  // caller-projected dense rows [H | r] stacked as they are (orcvio_msckf_upload_dense_rows)
    double* d_dense = nullptr;          // [dense_cap][NAP_max]
    bool ekf_eval = false;              // the four blocks are evaluated on the device from the SLAM features (k_ekf_eval)
    double* d_slam = nullptr;           // [12 cap] param 3 | inv_depth 1 | p_w 3 | p_fej 3 | z 2
    char* d_aug = nullptr; size_t aug_cap = 0;   // scratch of orcvio_msckf_cov_commit_new_features
    char* d_new = nullptr; size_t new_cap = 0, new_out_off = 0;   // entering features (orcvio_msckf_upload_new_features): inputs, then H_1 | H_2 | r_1
    int new_F = 0, new_idp = 3;
    double* d_Rf = nullptr;             // [maxF][6] R factor of every track's H_f (k_feature), for orcvio_msckf_augment_new_features
    int* d_sync = nullptr;              // device-wide counter of k_front (own allocation, zero between launches)
    // multi-GPU: RCCL communicator of this handle (orcvio_msckf_comm_init), the all-gather buffer [world][NAP_max^2] and
    // the gathered degrees of freedom of a sharded object update
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    double *d_gather = nullptr, *d_dofs = nullptr;
    hipStream_t comm_stream = nullptr;  // carries the early exchange of the degrees of freedom of a sharded object update
    double* h_dofs = nullptr;           // pinned [2 * world]
    bool A_deferred = false;            // the last run left S / Gpart only: d_A is assembled on demand (assemble_deferred)
    int front_chunks = 1;               // T3 row chunks of the last k_front launch
    int n_cus = 0;                      // compute units of the device
    int clean_NP = -1, clean_path = -1;   // layout for which the strictly-lower tiles of d_RP / d_RM are known to be zero
    double *d_RP = nullptr, *d_DinvP = nullptr, *d_U = nullptr, *d_M = nullptr, *d_RM = nullptr, *d_DinvM = nullptr;
    double *d_Z = nullptr, *d_Pout = nullptr, *d_dx = nullptr;
    double *d_La = nullptr, *d_DinvA = nullptr, *d_W = nullptr, *d_Y = nullptr, *d_KG = nullptr;   // optional outputs
    // object blocks (allocated on first use, grown on demand)
    double *d_Gobj = nullptr, *d_RF = nullptr, *d_DinvF = nullptr, *d_Yobj = nullptr, *d_objH = nullptr;
    double *d_obj_gamma = nullptr;
    double obj_thr = -1.0;   // chi-square threshold of the object update being finished (host value, passed to k_finish_sqrt)
    int *d_obj_i = nullptr, *d_obj_accept = nullptr;
    size_t cap_Gobj = 0, cap_RF = 0, cap_Yobj = 0, cap_objH = 0, cap_obj_i = 0;
    char* h_obj_stage_dev = nullptr;   // device-visible address of h_obj_stage (k_ingest pulls the staged tracks / rows out of it)
    bool pub_pending = false;          // the results of the last one-shot object update are on their way to the pinned block (k_epilogue): wait on h_flag
    char *h_obj_stage = nullptr, *d_obj_in = nullptr;   // input arena of an object update: pinned mirror + device copy (grown on demand)
    size_t obj_stage_cap = 0;
    std::vector<ObjUse> obj_use;        // (objects_local_tracks: per-track records, kept across calls: no allocation per frame)
    std::vector<int> obj_fnr;           // rows of every frame of the track being staged
    std::vector<int> obj_rowkp, obj_Ks; // keypoint block of every row / keypoint count of every object (structured QR of Hf)
    std::vector<unsigned long long> obj_vmask;   // (objects_local_tracks) observed keypoints of every frame of the track being staged
    // per-stage device times of the last object update (orcvio_msckf_profile_stages): events recorded between the stages
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<const char*> prof_names;
    int prof_n = 0;
    bool objects_mode = false;
    int obj_dof = 0, obj_rows = 0, obj_count = 0;
    double *d_T3 = nullptr, *d_Xobs = nullptr, *d_S = nullptr;
    int *d_clone_ptr = nullptr, *d_clone_obs = nullptr;   // d_clone_ptr: [0..N] first sparse row of every clone; d_clone_obs: obs_pos
    std::vector<int> h_clone_ptr, h_clone_obs;
    int s_chunks = 0;
    bool materialize = false;
    int feat_ablate = 0;
    // captured launch graph of run_update (valid for the current upload and launch stream)
    // every captured graph bakes in device pointers and launch arguments: anything that reallocates a buffer a graph may
    // reference, or changes an option / gate threshold, bumps this epoch; the epoch is part of the launch signature, so a
    // stale graph can never be replayed (use-after-free through ekf_reserve / upload_dense_rows)
    unsigned long long graph_epoch = 1;
    bool use_graph = true;
    size_t hs_rows_cap = 0;
    int gram_chunks_cap = 64;
    // host staging
    std::vector<double> h_poses, h_chi2;
    std::vector<int> h_row_ptr;
    double chi2_prob_cached = -1.0;
};

template <typename T>
static int grow(T** p, size_t* cap, size_t need) {
    if (*cap >= need) return ORCVIO_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIPCHK(hipMalloc(p, sizeof(T) * need));
    *cap = need;
    return ORCVIO_OK;
}

extern "C" {

int32_t orcvio_msckf_abi_version(void) { return ORCVIO_MSCKF_ABI_VERSION; }
const char* orcvio_msckf_last_error(void) { return g_last_error.c_str(); }

// ---- chi-square quantile (replaces boost::math::quantile, src/orcvio.cpp:486-494) --------
static double reg_lower_gamma(double a, double x) {
    if (x <= 0.0) return 0.0;
    const double lg = std::lgamma(a);
    if (x < a + 1.0) {   // series
        double term = 1.0 / a, sum = term, ap = a;
        for (int i = 0; i < 200000; ++i) {
            ap += 1.0;
            term *= x / ap;
            sum += term;
            if (std::fabs(term) < std::fabs(sum) * 1e-17) break;
        }
        return sum * std::exp(a * std::log(x) - x - lg);
    }
    // modified Lentz continued fraction for Q(a,x)
    const double tiny = 1e-300;
    double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
    for (int i = 1; i < 200000; ++i) {
        const double an = -(double)i * ((double)i - a);
        b += 2.0;
        d = an * d + b;
        if (std::fabs(d) < tiny) d = tiny;
        c = b + an / c;
        if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) < 1e-16) break;
    }
    return 1.0 - std::exp(a * std::log(x) - x - lg) * h;
}

double orcvio_msckf_chi2_quantile(int32_t dof, double prob) {
    if (dof < 1 || !(prob > 0.0 && prob < 1.0)) return NAN;
    const double a = 0.5 * dof;
    // normal quantile by bisection, Wilson-Hilferty start, then safeguarded Newton
    double zl = -12.0, zh = 12.0;
    for (int i = 0; i < 100; ++i) {
        const double zm = 0.5 * (zl + zh);
        if (0.5 * std::erfc(-zm / std::sqrt(2.0)) < prob) zl = zm; else zh = zm;
    }
    const double zq = 0.5 * (zl + zh);
    const double wh = 1.0 - 2.0 / (9.0 * dof) + zq * std::sqrt(2.0 / (9.0 * dof));
    double x = dof * wh * wh * wh;
    if (!(x > 0.0)) x = 1e-3;
    double lo = 0.0, hi = 4.0 * x + 60.0;
    for (int it = 0; it < 300; ++it) {
        const double f = reg_lower_gamma(a, 0.5 * x) - prob;
        if (f > 0.0) hi = x; else lo = x;
        const double lpdf = (a - 1.0) * std::log(0.5 * x) - 0.5 * x - std::lgamma(a) - std::log(2.0);
        double xn = x - f / std::exp(lpdf);
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        const bool done = std::fabs(xn - x) <= 1e-15 * std::fabs(x);
        x = xn;
        if (done) break;
    }
    return x;
}

// ---- arenas -----------------------------------------------------------------------------------
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
// [poses | obs_ptr | row_ptr | clone_ptr | p_w | obs_clone | clone_obs | obs_z | (obs_zvel) | (P)], 256-byte aligned parts
static size_t inputs_bytes(int N, int F, int nobs, bool zvel, size_t n_with_P) {
    return al256(sizeof(double) * POSE_STRIDE * N) + 2 * al256(sizeof(int) * (F + 1)) + al256(sizeof(int) * (2 * N + 4)) +
           al256(sizeof(double) * 3 * F) + 2 * al256(sizeof(int) * nobs) + al256(sizeof(double) * 2 * nobs) +
           (zvel ? al256(sizeof(double) * 2 * nobs) : 0) + al256(sizeof(double) * n_with_P * n_with_P);
}
static size_t outputs_bytes(int n, int F) {
    return 256 + al256(sizeof(double) * n) + al256(sizeof(double) * F) + al256(sizeof(int) * F) + al256(sizeof(double) * (size_t)n * n);
}
// Compact layout of the inputs for the sizes of this upload; with_P = false: the prior is the resident covariance and
// does not travel.  The device pointers are functions of (N, F, nobs, zvel, with_P) only, so equal shapes give equal
// pointers (the captured launch graphs stay valid; d_P is part of the launch signature all the same).
static void layout_inputs(orcvio_msckf_handle* h, int N, int F, int nobs, bool zvel, bool with_P, int n) {
    size_t o = 0;
    h->io_poses = o; o += al256(sizeof(double) * POSE_STRIDE * N);
    h->io_optr = o; o += al256(sizeof(int) * (F + 1));
    h->io_rptr = o; o += al256(sizeof(int) * (F + 1));
    h->io_cptr = o; o += al256(sizeof(int) * (2 * N + 4));
    h->io_pw = o; o += al256(sizeof(double) * 3 * F);
    h->io_oclone = o; o += al256(sizeof(int) * nobs);
    h->io_cobs = o; o += al256(sizeof(int) * nobs);
    h->io_z = o; o += al256(sizeof(double) * 2 * nobs);
    h->io_zvel = h->io_z;   // (never read without estimate_td: any valid address)
    if (zvel) { h->io_zvel = o; o += al256(sizeof(double) * 2 * nobs); }
    h->io_P = o;
    h->in_used = o;         // what travels besides P
    char* d = h->d_in;
    h->d_poses = reinterpret_cast<double*>(d + h->io_poses);
    h->d_obs_ptr = reinterpret_cast<int*>(d + h->io_optr);
    h->d_row_ptr = reinterpret_cast<int*>(d + h->io_rptr);
    h->d_clone_ptr = reinterpret_cast<int*>(d + h->io_cptr);
    h->d_pw = reinterpret_cast<double*>(d + h->io_pw);
    h->d_obs_clone = reinterpret_cast<int*>(d + h->io_oclone);
    h->d_clone_obs = reinterpret_cast<int*>(d + h->io_cobs);
    h->d_obs_z = reinterpret_cast<double*>(d + h->io_z);
    h->d_obs_zvel = reinterpret_cast<double*>(d + h->io_zvel);
    h->d_P = with_P ? reinterpret_cast<double*>(d + h->io_P) : h->d_Pres;
    (void)n;
}
// [info 64 ints | dx | gamma | accept | P+]: the small part is one copy, with P+ behind it one longer copy
static void layout_outputs(orcvio_msckf_handle* h, int n, int F) {
    size_t o = 256;
    h->d_info = reinterpret_cast<int*>(h->d_outs);
    h->d_flag = h->d_info + 32;
    h->oo_dx = o; o += al256(sizeof(double) * n);
    h->oo_gamma = o; o += al256(sizeof(double) * (F > 0 ? F : 1));
    h->oo_accept = o; o += al256(sizeof(int) * (F > 0 ? F : 1));
    h->oo_Pout = o;
    h->outs_small = o;
    h->d_dx = reinterpret_cast<double*>(h->d_outs + h->oo_dx);
    h->d_gamma = reinterpret_cast<double*>(h->d_outs + h->oo_gamma);
    h->d_accept = reinterpret_cast<int*>(h->d_outs + h->oo_accept);
    h->d_Pout = reinterpret_cast<double*>(h->d_outs + h->oo_Pout);
}

// host-pinned (device-visible) -> HBM by a kernel instead of a copy-engine transfer: lower latency for the few hundred KB an
// update moves, and it can be a node of a captured graph like any other launch
static int launch_ingest(orcvio_msckf_handle* h, hipStream_t s, const void* src_dev, void* dst, size_t bytes, bool kernel = true) {
    if (bytes == 0) return ORCVIO_OK;
    if (!kernel) {   // the copy engine (pinned memory: asynchronous)
        HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyHostToDevice, s));
        return ORCVIO_OK;
    }
    const size_t n16 = (bytes + 15) / 16;
    int grid = (int)((n16 + 255) / 256);
    if (grid > 4 * h->n_cus) grid = 4 * h->n_cus;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_ingest, dim3(grid), dim3(256), 0, s, reinterpret_cast<const u32x4*>(src_dev), reinterpret_cast<u32x4*>(dst), n16);
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}
static bool obj_ingest_kernel() {   // diagnostics: ORCVIO_OBJ_INGEST=0 moves the inputs of an object update with the copy engine
    static const bool v = [] { const char* e = getenv("ORCVIO_OBJ_INGEST"); return e ? atoi(e) != 0 : true; }();
    return v;
}
// The results of a one-shot object update: device-to-host copy + stream synchronisation (default), or ORCVIO_OBJ_PUBLISH=1:
// k_epilogue + the flag, as the feature updates do.  Measured on one box (scripts/gpu_obj_timing.py, config 3, median ms, host
// buffers / resident prior): ingest kernel + copy out 0.2118 / 0.1625, ingest kernel + flag 0.2111 / 0.1684, copy engine both
// ways 0.2175 / 0.1684 -- the object update is a chain of a dozen plain launches whose enqueue the host is still busy with
// when the first kernels run, so the wake-up is not what it waits for.
static bool obj_publish_kernel() {
    static const bool v = [] { const char* e = getenv("ORCVIO_OBJ_PUBLISH"); return e ? atoi(e) != 0 : false; }();
    return v;
}
// the results of the update on `s` -> the pinned output block, then the flag (k_epilogue without a commit): what replaces
// the device-to-host copy + stream synchronisation of the one-shot calls
static int publish_enqueue(orcvio_msckf_handle* h, hipStream_t s, bool want_P) {
    EpilogueArgs ea{};
    ea.small_src = reinterpret_cast<const u32x4*>(h->d_outs);
    ea.small_dst = reinterpret_cast<u32x4*>(h->h_stage_dev + h->in_cap);
    ea.small16 = h->outs_small / 16;
    ea.P_src = reinterpret_cast<const u32x4*>(h->d_outs + h->oo_Pout);
    ea.P_dst = reinterpret_cast<u32x4*>(h->h_stage_dev + h->in_cap + h->oo_Pout);
    ea.P16 = want_P ? (sizeof(double) * (size_t)h->n * h->n + 15) / 16 : 0;
    ea.nb_P = want_P ? 40 : 0;
    ea.commit = 0;
    ea.counter = h->d_pubcnt; ea.seq = h->d_seq; ea.flag = h->h_flag_dev;
    hipLaunchKernelGGL(k_epilogue, dim3(1 + ea.nb_P), dim3(256), 0, s, ea);
    HIPCHK(hipGetLastError());
    h->pub_pending = true;
    h->pub_enqueued++;
    return ORCVIO_OK;
}

// ---- create / destroy ---------------------------------------------------------------------
static void free_all(orcvio_msckf_handle* h) {
    void* ptrs[] = {h->d_in, h->d_outs,
                    h->d_chi2, h->d_Hs, h->d_Gpart, h->d_Ab, h->d_A, h->d_RP,
                    h->d_DinvP, h->d_U, h->d_M, h->d_RM, h->d_DinvM, h->d_Z, h->d_La, h->d_DinvA,
                    h->d_W, h->d_Y, h->d_KG, h->d_Gobj, h->d_RF, h->d_DinvF, h->d_Yobj, h->d_objH,
                    h->d_obj_gamma, h->d_obj_i, h->d_obj_accept, h->d_T3, h->d_Xobs, h->d_S,
                    h->d_Pres, h->d_Ptmp, h->d_Sres, h->d_Stmp, h->d_covT, h->d_covmap, h->d_skip, h->d_tri_valid, h->d_tri_flags, h->d_tri_init, h->d_tri_sol, h->d_tri_cost, h->d_sync,
                    h->d_ekf_i, h->d_ekf_d, h->d_ekf_E, h->d_Gekf, h->d_ekf_gamma, h->d_ekf_accept, h->d_slam, h->d_dense, h->d_Rf, h->d_new, h->d_aug};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->h_obj_stage) (void)hipHostFree(h->h_obj_stage);
    if (h->d_obj_in) (void)hipFree(h->d_obj_in);
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_side) (void)hipEventDestroy(h->ev_side);
    for (auto* g : {&h->g_update, &h->g_local, &h->g_finish, &h->g_io})
        for (auto& w : g->way) {
            if (w.exec) (void)hipGraphExecDestroy(w.exec);
            if (w.graph) (void)hipGraphDestroy(w.graph);
        }
    if (h->h_flag) (void)hipHostFree(h->h_flag);
    if (h->d_seq) (void)hipFree(h->d_seq);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->side) (void)hipStreamDestroy(h->side);
}

int32_t orcvio_msckf_create(int32_t device, int32_t max_clones, int32_t max_features, int32_t max_observations,
                            orcvio_msckf_handle** out) {
    if (!out || max_clones < 1 || max_clones > ORCVIO_MAX_CLONES || max_features < 1 || max_observations < 1) {
        g_last_error = "orcvio_msckf_create: invalid capacity";
        return ORCVIO_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_last_error = "orcvio_msckf_create: no HIP device (this library has no CPU path)";
        return ORCVIO_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        g_last_error = "hipGetDeviceProperties failed";
        return ORCVIO_ERR_NO_DEVICE;
    }
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_last_error = std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
        return ORCVIO_ERR_NO_DEVICE;
    }
    auto* h = new orcvio_msckf_handle();
    h->device = device;
    h->n_cus = prop.multiProcessorCount;
    if (const char* e = getenv("ORCVIO_FUSED_SOLVE")) h->fused_solve = atoi(e);   // diagnostics: defaults of the options
    if (const char* e = getenv("ORCVIO_FUSED_FRONT")) h->front_fused = atoi(e);
    if (const char* e = getenv("ORCVIO_FRONT_SPIN")) h->front_spin_limit = atoi(e);
    if (const char* e = getenv("ORCVIO_REV_PRIOR")) h->rev_prior_opt = atoi(e) != 0;
    h->maxN = max_clones;
    h->maxF = max_features;
    h->maxObs = max_observations;
    h->n_max = 46 + 6 * max_clones;
    h->NAP_max = round_up(h->n_max - 15 + 1, 16);
    h->NP_max = round_up(h->n_max + 1, 16);
    if (h->NP_max > POTRF_MAXN || h->NP_max / 16 > TRSM_MAXBLK) {
        delete h;
        g_last_error = "orcvio_msckf_create: window too large for the single-workgroup factorisation";
        return ORCVIO_ERR_CAPACITY;
    }
    int rc = [&]() -> int {
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming));
        const size_t nn = (size_t)h->n_max * h->n_max, pp = (size_t)h->NAP_max * h->NAP_max;
        const size_t np2 = (size_t)h->NP_max * h->NP_max;
        h->hs_rows_cap = (size_t)2 * max_observations + 16;
        {   // input / output arenas at their worst-case size, pinned mirror of both
            h->in_cap = inputs_bytes(max_clones, max_features, max_observations, true, (size_t)h->n_max);
            h->outs_cap = outputs_bytes(h->n_max, max_features);
            HIPCHK(hipMalloc(&h->d_in, h->in_cap));
            HIPCHK(hipMalloc(&h->d_outs, h->outs_cap));
            HIPCHK(hipMemset(h->d_in, 0, h->in_cap));
            h->stage_bytes = h->in_cap + h->outs_cap;
            // pinned, mapped into the device's address space, host-coherent (fine-grained): kernels read the inputs from it
            // (k_ingest) and write the results into it (k_epilogue); the copy engines can use it as before
            HIPCHK(hipHostMalloc(&h->h_stage, h->stage_bytes, hipHostMallocMapped | hipHostMallocCoherent));
            HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->h_stage_dev), h->h_stage, 0));
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->h_flag), 256, hipHostMallocMapped | hipHostMallocCoherent));
            HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->h_flag_dev), h->h_flag, 0));
            std::memset(h->h_flag, 0, 256);
            HIPCHK(hipMalloc(&h->d_seq, 256));
            HIPCHK(hipMemset(h->d_seq, 0, 256));
            h->d_pubcnt = reinterpret_cast<int*>(h->d_seq) + 32;   // (own 128-byte line)
            if (const char* e = getenv("ORCVIO_IO_SPIN_SECONDS")) h->io_spin_seconds = atof(e);
            layout_inputs(h, max_clones, max_features, max_observations, true, true, h->n_max);
            layout_outputs(h, h->n_max, max_features);
        }
        HIPCHK(hipMalloc(&h->d_skip, sizeof(int) * max_features));
        HIPCHK(hipMalloc(&h->d_Pres, sizeof(double) * nn));
        HIPCHK(hipMalloc(&h->d_Ptmp, sizeof(double) * nn));
        HIPCHK(hipMalloc(&h->d_Sres, sizeof(double) * np2));
        HIPCHK(hipMalloc(&h->d_Stmp, sizeof(double) * np2));
        HIPCHK(hipMalloc(&h->d_covT, sizeof(double) * (46 * (size_t)h->n_max + 2 * 46 * 46)));   // Phi P rows, then Phi and Q
        HIPCHK(hipMalloc(&h->d_covmap, sizeof(int) * h->n_max));
        HIPCHK(hipMalloc(&h->d_tri_valid, sizeof(int) * max_features));
        HIPCHK(hipMalloc(&h->d_tri_flags, sizeof(int) * max_features));
        HIPCHK(hipMalloc(&h->d_tri_init, sizeof(int) * max_features));
        HIPCHK(hipMalloc(&h->d_tri_sol, sizeof(double) * 3 * max_features));
        HIPCHK(hipMalloc(&h->d_tri_cost, sizeof(double) * max_features));
        HIPCHK(hipMalloc(&h->d_chi2, sizeof(double) * ORCVIO_CHI2_TABLE));
        HIPCHK(hipMalloc(&h->d_T3, sizeof(double) * (size_t)3 * max_features * h->NAP_max));
        HIPCHK(hipMalloc(&h->d_Rf, sizeof(double) * (size_t)6 * max_features));
        HIPCHK(hipMalloc(&h->d_Xobs, sizeof(double) * (size_t)32 * max_observations));
        HIPCHK(hipMalloc(&h->d_S, sizeof(double) * (size_t)256 * (2 * max_observations / 256 + max_clones + 2)));
        HIPCHK(hipMalloc(&h->d_Gpart, sizeof(double) * pp * h->gram_chunks_cap));
        HIPCHK(hipMalloc(&h->d_Ab, sizeof(double) * pp));
        HIPCHK(hipMalloc(&h->d_A, sizeof(double) * pp));
        HIPCHK(hipMalloc(&h->d_RP, sizeof(double) * np2));
        HIPCHK(hipMalloc(&h->d_DinvP, sizeof(double) * 256 * TRSM_MAXBLK));
        HIPCHK(hipMalloc(&h->d_U, sizeof(double) * np2));
        HIPCHK(hipMalloc(&h->d_M, sizeof(double) * np2));
        HIPCHK(hipMalloc(&h->d_RM, sizeof(double) * np2));
        HIPCHK(hipMalloc(&h->d_DinvM, sizeof(double) * 256 * TRSM_MAXBLK));
        HIPCHK(hipMalloc(&h->d_Z, sizeof(double) * np2));
        HIPCHK(hipMalloc(&h->d_La, sizeof(double) * pp));
        HIPCHK(hipMalloc(&h->d_DinvA, sizeof(double) * 256 * TRSM_MAXBLK));
        HIPCHK(hipMalloc(&h->d_W, sizeof(double) * np2));
        HIPCHK(hipMalloc(&h->d_Y, sizeof(double) * np2));
        HIPCHK(hipMalloc(&h->d_KG, sizeof(double) * np2));
        HIPCHK(hipMemset(h->d_info, 0, sizeof(int) * 64));   // (head of the outputs arena)
        HIPCHK(hipMalloc(&h->d_sync, 256));
        HIPCHK(hipMemset(h->d_sync, 0, 256));
        HIPCHK(hipMemset(h->d_RP, 0, sizeof(double) * np2));   // strictly-lower tiles of the upper factors stay 0
        HIPCHK(hipMemset(h->d_RM, 0, sizeof(double) * np2));
        // opt in to large dynamic LDS for the feature kernel instantiations
        const int lds_max = 160 * 1024;
        HIPCHK(hipFuncSetAttribute((const void*)k_feature<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        HIPCHK(hipFuncSetAttribute((const void*)k_feature<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        HIPCHK(hipFuncSetAttribute((const void*)k_feature<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        HIPCHK(hipFuncSetAttribute((const void*)k_feature<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        HIPCHK(hipFuncSetAttribute((const void*)k_feature<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        HIPCHK(hipFuncSetAttribute((const void*)k_feature<6>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        HIPCHK(hipFuncSetAttribute((const void*)k_feature<7>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        return ORCVIO_OK;
    }();
    if (rc != ORCVIO_OK) {
        free_all(h);
        delete h;
        return rc;
    }
    *out = h;
    return ORCVIO_OK;
}

void orcvio_msckf_destroy(orcvio_msckf_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)orcvio_msckf_comm_destroy(h);
    free_all(h);
    delete h;
}

int32_t orcvio_msckf_set_option(orcvio_msckf_handle* h, int32_t option, int32_t value) {
    if (!h) return ORCVIO_ERR_INVALID;
    if (option == ORCVIO_OPT_MATERIALIZE_STACK) {
        HIPCHK(hipSetDevice(h->device));
        if (value && !h->d_Hs) HIPCHK(hipMalloc(&h->d_Hs, sizeof(double) * h->hs_rows_cap * h->NAP_max));
        h->materialize = value != 0;
        h->graph_epoch++;
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_FUSED_SOLVE) {
        h->fused_solve = value != 0;
        h->graph_epoch++;
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_FUSED_FRONT) {
        h->front_fused = value != 0;
        h->graph_epoch++;
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_EKF_ROWS) {
        h->ekf_mode = value != 0;   // takes effect with the next upload (part of the launch signature)
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_EXTRA_STATES) {
        if (value < 0 || 22 + 6 + value > h->n_max) { g_last_error = "orcvio_msckf_set_option: extra states out of range"; return ORCVIO_ERR_INVALID; }
        h->n_extra = value;   // takes effect with the next upload / update call (part of the launch signature)
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_RESIDENT_FACTOR) {
        h->factor_opt = value != 0;
        if (!h->factor_opt) h->fac_valid = false;
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_REF_STACK_HF) {
        h->ref_stack_hf = value != 0;
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_SCHMIDT_STATES) {
        if (value < 0 || value > h->maxN) { g_last_error = "orcvio_msckf_set_option: nuisance states out of range"; return ORCVIO_ERR_INVALID; }
        h->n_nui = value;   // takes effect with the next upload (part of the launch signature)
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_OBJECT_DOF) {
        h->obj_dof_rank = value != 0;
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_REF_H2_LDLT) {
        h->ref_h2_ldlt = value != 0;
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_OBJECT_QR) {
        h->arrow_opt = value != 0;
        return ORCVIO_OK;
    }
    if (option == ORCVIO_OPT_STAGE_PROFILE) {
        h->prof_on = value != 0;
        h->prof_n = 0;
        return ORCVIO_OK;
    }
    g_last_error = "orcvio_msckf_set_option: unknown option";
    return ORCVIO_ERR_INVALID;
}

static int factor_layout_clean(orcvio_msckf_handle* h);
static int launch_ekf(orcvio_msckf_handle* h, hipStream_t s);
static int io_wait(orcvio_msckf_handle* h, hipStream_t s);
static int comm_stream_wait(orcvio_msckf_handle* h, hipStream_t s, const char* who);
static int feature_outcome(orcvio_msckf_handle* h, const char* so, int32_t* stats);
static int run_finish_impl(orcvio_msckf_handle* h, const double* d_blocks, int n_blocks, size_t stride, const double* meta0, hipStream_t s);
static int objects_finish_impl(orcvio_msckf_handle* h, const double* d_blocks, int n_blocks, size_t stride, const double* meta0, int dof_total, hipStream_t s);
// Gram of the rows stacked under the MSCKF rows (EKF-SLAM rows that passed their gate, caller-projected dense rows)
static inline const double* extra_gram(const orcvio_msckf_handle* h) { return (h->ekf_F > 0 || h->dense_rows > 0) ? h->d_Gekf : nullptr; }

// The prior of this update comes as P == NULL (resident covariance): if its square-root factor is resident too, the update
// uses it instead of factoring P (kf = its column count = the dimension of M).
static void select_prior_factor(orcvio_msckf_handle* h, bool with_P) {
    h->use_factor = !with_P && h->factor_opt && h->fac_valid && h->fac_n == h->n && h->res_n == h->n &&
                    round_up(h->fac_k, 16) <= POTRF_MAXN && round_up(h->fac_k, 16) / 16 <= TRSM_MAXBLK &&
                    round_up(h->fac_k, 16) <= h->NP_max;
    h->kf = h->use_factor ? h->fac_k : h->n;
}
// ... and whether the factor in use ends in columns that are zero in the active rows (call when reg_path is known)
static inline bool rev_prior_active(const orcvio_msckf_handle* h) { return h->rev_prior_opt && h->reg_path && h->fused_solve && !h->use_factor; }
static void select_tail(orcvio_msckf_handle* h) {
    const int t = h->use_factor ? h->fac_tail : (rev_prior_active(h) ? 15 : 0);
    h->tail = (h->reg_path && h->fused_solve && t > 0 && h->kf - t >= 16) ? t : 0;
}
// L(i,j) of the prior's factor = base[i * sLi + j * sLj], i < n, j < kf
struct PriorFactor { const double* base; long sLi, sLj; };
static inline void factor_strides(const orcvio_msckf_handle* h, long& sLi, long& sLj);
static PriorFactor prior_factor(const orcvio_msckf_handle* h) {
    if (h->use_factor) return PriorFactor{h->d_Sres, 1L, (long)h->fac_ld};
    if (rev_prior_active(h)) return PriorFactor{h->d_RP + (h->n - 1), -1L, (long)h->NP};   // S(i, c) = L'(n-1-i, c) = RP[c NP + n-1-i]
    long sLi, sLj;
    factor_strides(h, sLi, sLj);
    return PriorFactor{h->d_RP, sLi, sLj};
}

// ---- upload --------------------------------------------------------------------------------
// Three steps: upload_begin (sizes -> problem dimensions, arena layout), the caller's arrays written into the pinned arena
// (by orcvio_msckf_upload from its arguments, or by the caller itself through orcvio_msckf_io_begin's pointers), and
// upload_finalize (validation of what stands in the arena, the derived index arrays).  The arena reaches the device by one
// asynchronous copy (staged callers) or by the first kernel of the update's graph (k_ingest, orcvio_msckf_io_update).
static int upload_begin(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, int N, int F, int nobs, bool with_P, bool have_zvel,
                        const char* who) {
    if (flags->leg_dim != 22 && flags->leg_dim != 46) { g_last_error = std::string(who) + ": leg_dim must be 22 or 46"; return ORCVIO_ERR_INVALID; }
    if (N < 1 || F < 0) { g_last_error = std::string(who) + ": bad sizes"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));   // the pinned staging buffer of the previous upload is free again
    if (h->dl_pending) { HIPCHK(hipStreamSynchronize(h->dl_stream)); h->dl_pending = false; }
    if (N > h->maxN || F > h->maxF) { g_last_error = std::string(who) + ": exceeds handle capacity"; return ORCVIO_ERR_CAPACITY; }
    if (nobs < 0) { g_last_error = std::string(who) + ": obs_ptr not monotone"; return ORCVIO_ERR_INVALID; }
    if (nobs > h->maxObs) { g_last_error = std::string(who) + ": too many observations"; return ORCVIO_ERR_CAPACITY; }
    if (flags->estimate_td && F > 0 && !have_zvel) { g_last_error = std::string(who) + ": obs_zvel required with estimate_td"; return ORCVIO_ERR_INVALID; }
    const int n = flags->leg_dim + 6 * N + h->n_extra;   // (n_extra: states behind the clones that no row of this update touches)
    if (n > h->n_max) { g_last_error = "window + extra states exceed the handle's capacity"; return ORCVIO_ERR_CAPACITY; }
    if (6 * h->n_nui > h->n_extra || N + h->n_nui > h->maxN) { g_last_error = std::string(who) + ": nuisance states do not fit the extra states / the pose capacity"; return ORCVIO_ERR_CAPACITY; }
    if (!with_P && h->res_n != n) { g_last_error = std::string(who) + ": P == NULL but the resident covariance does not match the window"; return ORCVIO_ERR_INVALID; }
    h->uploaded = false; h->ran = false; h->io_open = false;
    h->flags = *flags;
    h->N = N; h->F = F; h->nobs = nobs;
    h->n = n;
    h->NA = h->ekf_mode ? h->n - 15 : flags->leg_dim + 6 * N - 15;   // (EKF-SLAM rows reach into the extra states)
    h->ekf_F = 0; h->dense_rows = 0; h->new_F = 0;
    h->NAP = round_up(h->NA + 1, 16);
    // chi-square table (src/orcvio.cpp:481-494)
    if (h->chi2_prob_cached != flags->chi2_prob) {
        h->h_chi2.assign(ORCVIO_CHI2_TABLE, 0.0);
        for (int d = 1; d < ORCVIO_CHI2_TABLE; ++d) h->h_chi2[d] = orcvio_msckf_chi2_quantile(d, flags->chi2_prob);
        HIPCHK(hipMemcpyAsync(h->d_chi2, h->h_chi2.data(), sizeof(double) * ORCVIO_CHI2_TABLE, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->chi2_prob_cached = flags->chi2_prob;
    }
    const bool with_zvel = have_zvel && flags->estimate_td;   // read by the kernels only under estimate_td
    layout_inputs(h, N + h->n_nui, F, nobs, with_zvel, with_P, h->n);   // (pose slots of the nuisance states behind the window's)
    layout_outputs(h, h->n, F);
    h->io_with_P = with_P;
    return ORCVIO_OK;
}

// bytes of the arena that travel: everything in front of P, and P itself when the prior comes from the host
static inline size_t upload_bytes(const orcvio_msckf_handle* h) {
    return h->io_with_P ? h->io_P + sizeof(double) * (size_t)h->n * h->n : h->in_used;
}

// The arena holds poses, obs_ptr, p_w, obs_clone, obs_z (obs_zvel, P): validate the index arrays and derive row_ptr (rows of
// every projected block), clone_obs / clone_ptr (observations grouped by clone: the sparse part of the compression).
static int upload_finalize(orcvio_msckf_handle* h, const char* who) {
    const int N = h->N, F = h->F, nobs = h->nobs;
    // the prior: the caller's P in the arena, or the resident covariance -- with its square-root factor if that is known NOW
    // (an orcvio_msckf_io_update may follow a commit, a cov_set, an augmentation of the previous one)
    if (!h->io_with_P) {
        if (h->res_n != h->n) { g_last_error = std::string(who) + ": P == NULL but the resident covariance does not match the window"; return ORCVIO_ERR_INVALID; }
        h->d_P = h->d_Pres;
    }
    select_prior_factor(h, h->io_with_P);
    h->NP = round_up(h->n > h->kf ? h->n : h->kf, 16);
    h->ldz = round_up(h->n + 1, 16);
    h->reg_path = (h->NP / 16) <= 14;
    select_tail(h);
    { const int rcl = factor_layout_clean(h); if (rcl != ORCVIO_OK) return rcl; }
    char* st = h->h_stage;
    const int* obs_ptr = reinterpret_cast<const int*>(st + h->io_optr);
    const int* obs_clone = reinterpret_cast<const int*>(st + h->io_oclone);
    if (F > 0 && obs_ptr[0] < 0) { g_last_error = std::string(who) + ": obs_ptr starts below zero"; return ORCVIO_ERR_INVALID; }
    if (F > 0 && obs_ptr[F] != nobs) { g_last_error = std::string(who) + ": obs_ptr[F] differs from the number of observations"; return ORCVIO_ERR_INVALID; }
    // row offsets, track-length limits
    h->h_row_ptr.resize(F + 1);
    int* row_ptr = h->h_row_ptr.data();
    row_ptr[0] = 0;
    int Mmax = 2;
    for (int j = 0; j < F; ++j) {
        const int M = obs_ptr[j + 1] - obs_ptr[j];
        if (M < 0) { g_last_error = std::string(who) + ": obs_ptr not monotone"; return ORCVIO_ERR_INVALID; }
        if (M > ORCVIO_MAX_TRACK) { g_last_error = std::string(who) + ": track longer than ORCVIO_MAX_TRACK"; return ORCVIO_ERR_TRACK_TOO_LONG; }
        if (M > Mmax) Mmax = M;
        row_ptr[j + 1] = row_ptr[j] + (M >= 2 ? 2 * M - 3 : 0);
    }
    std::memcpy(st + h->io_rptr, row_ptr, sizeof(int) * (F + 1));
    // observations grouped by clone: position of every observation in the clone-sorted order, and the row range of
    // every clone (two rows per observation) for the sparse part of the compression; the clone indices are checked on the way.
    // (Two scalar passes over the observations, ~8 us at 12 000 of them, on the critical path of a zero-copy update.  Tried in
    //  round 3 and not kept: the same stable counting sort by one workgroup of the ingest kernel while the others copy -- the
    //  host's part fell to 1.6 us, but that workgroup needed ~30 us (a PCIe round trip for the keys, then 47 dependent LDS
    //  byte updates per thread, twice) against 9.5 us for the copy it was to hide under: 0.148 ms per update instead of 0.127.)
    {
        int cnt[ORCVIO_MAX_CLONES + 2] = {0};
        unsigned bad = 0;
        for (int o = 0; o < nobs; ++o) {
            const unsigned c = (unsigned)obs_clone[o];
            bad |= (c >= (unsigned)N);
            cnt[(c < (unsigned)N ? c : 0u) + 1]++;
        }
        if (bad) { g_last_error = std::string(who) + ": obs_clone out of range"; return ORCVIO_ERR_INVALID; }
        for (int i = 0; i < N; ++i) cnt[i + 1] += cnt[i];
        int* clone_obs = reinterpret_cast<int*>(st + h->io_cobs);
        int fill[ORCVIO_MAX_CLONES + 2];
        std::memcpy(fill, cnt, sizeof(int) * (N + 1));
        for (int o = 0; o < nobs; ++o) clone_obs[o] = fill[obs_clone[o]]++;
        int* cptr = reinterpret_cast<int*>(st + h->io_cptr);   // [0..N] row offsets
        for (int i = 0; i <= N; ++i) cptr[i] = 2 * cnt[i];
        h->s_chunks = N;
    }
    h->m_tot = row_ptr[F];
    h->Mmax = Mmax;
    // Gram chunking: a workgroup of 16 wavefronts per (tile, chunk); up to 1024 rows per chunk keeps every wavefront
    // at one batch of loads (64 rows) and the number of partial Grams small
    int chunks = (3 * F + 1023) / 1024;
    if (chunks > h->gram_chunks_cap) chunks = h->gram_chunks_cap;
    if (chunks < 1) chunks = 1;
    const int t3rows = 3 * F;   // the dense part of the compression: three rows per track
    int rpc = round_up((t3rows + chunks - 1) / chunks, 8);
    if (rpc < 8) rpc = 8;
    h->rows_per_chunk = rpc;
    h->chunks = t3rows > 0 ? (t3rows + rpc - 1) / rpc : 1;
    h->uploaded = true;
    h->ran = false;
    h->skip_active = false;
    h->objects_mode = false;   // (a staged object update may have left it set)
    return ORCVIO_OK;
}

// the caller's arrays -> the arena (orcvio_msckf_upload and the copying one-shot calls)
static void stage_inputs(orcvio_msckf_handle* h, const orcvio_msckf_window* w, const orcvio_msckf_tracks* tr, const double* P) {
    const int N = h->N, F = h->F, nobs = h->nobs;
    char* st = h->h_stage;
    if (P) std::memcpy(st + h->io_P, P, sizeof(double) * (size_t)h->n * h->n);
    double* poses = reinterpret_cast<double*>(st + h->io_poses);
    const double* tfej = w->t_fej ? w->t_fej : w->t_b_w;
    for (int i = 0; i < N; ++i) {
        double* r = poses + (size_t)POSE_STRIDE * i;
        std::memcpy(r + POSE_R_B2W, w->R_b2w + 9 * i, 9 * sizeof(double));
        std::memcpy(r + POSE_T_B_W, w->t_b_w + 3 * i, 3 * sizeof(double));
        std::memcpy(r + POSE_T_FEJ, tfej + 3 * i, 3 * sizeof(double));
        std::memcpy(r + POSE_R_B2C, w->R_b2c + 9 * i, 9 * sizeof(double));
        std::memcpy(r + POSE_T_C_B, w->t_c_b + 3 * i, 3 * sizeof(double));
        r[27] = 0.0;
    }
    std::memcpy(st + h->io_optr, tr->obs_ptr, sizeof(int) * (F + 1));
    if (F > 0) {
        if (tr->p_w) std::memcpy(st + h->io_pw, tr->p_w, sizeof(double) * 3 * F);
        else std::memset(st + h->io_pw, 0, sizeof(double) * 3 * F);   // positions come from orcvio_msckf_triangulate_uploaded
        if (nobs > 0) {
            std::memcpy(st + h->io_oclone, tr->obs_clone, sizeof(int) * nobs);
            std::memcpy(st + h->io_z, tr->obs_z, sizeof(double) * 2 * nobs);
            if (h->io_zvel != h->io_z) std::memcpy(st + h->io_zvel, tr->obs_zvel, sizeof(double) * 2 * nobs);
        }
    }
}

static int upload_to_arena(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, const orcvio_msckf_window* w,
                           const orcvio_msckf_tracks* tr, const double* P, const char* who) {
    if (!h || !flags || !w || !tr || !w->R_b2w || !w->t_b_w || !w->R_b2c || !w->t_c_b || !tr->obs_ptr) {
        g_last_error = std::string(who) + ": null argument";
        return ORCVIO_ERR_INVALID;
    }
    const int N = w->n_clones, F = tr->n_features;
    if (N < 1 || F < 0) { g_last_error = std::string(who) + ": bad sizes"; return ORCVIO_ERR_INVALID; }
    if (F > h->maxF) { g_last_error = std::string(who) + ": exceeds handle capacity"; return ORCVIO_ERR_CAPACITY; }
    if (F > 0 && tr->obs_ptr[0] < 0) { g_last_error = std::string(who) + ": obs_ptr starts below zero"; return ORCVIO_ERR_INVALID; }
    const int nobs = F > 0 ? tr->obs_ptr[F] : 0;   // (observations in front of obs_ptr[0] are carried along unused)
    if (F > 0 && nobs > 0 && (!tr->obs_clone || !tr->obs_z)) { g_last_error = std::string(who) + ": null track arrays"; return ORCVIO_ERR_INVALID; }
    int rc = upload_begin(h, flags, N, F, nobs, P != nullptr, tr->obs_zvel != nullptr, who);
    if (rc != ORCVIO_OK) return rc;
    stage_inputs(h, w, tr, P);
    rc = upload_finalize(h, who);
    if (rc != ORCVIO_OK) return rc;
    h->pw_missing = (F > 0 && !tr->p_w);
    return ORCVIO_OK;
}

int32_t orcvio_msckf_upload(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, const orcvio_msckf_window* w,
                            const orcvio_msckf_tracks* tr, const double* P) {
    const int rc = upload_to_arena(h, flags, w, tr, P, "orcvio_msckf_upload");
    if (rc != ORCVIO_OK) return rc;
    // ONE asynchronous copy (no synchronisation: the staging buffer is rewritten only by the next upload, which the caller
    // issues after the download / sync of this update; the kernels are ordered behind the copy on the same stream)
    HIPCHK(hipMemcpyAsync(h->d_in, h->h_stage, upload_bytes(h), hipMemcpyHostToDevice, h->stream));
    return ORCVIO_OK;
}

// ---- launches --------------------------------------------------------------------------------
static FeatArgs feature_args(const orcvio_msckf_handle* h) {
    FeatArgs a;
    a.poses = h->d_poses; a.p_w = h->d_pw; a.obs_ptr = h->d_obs_ptr; a.obs_clone = h->d_obs_clone;
    a.obs_z = h->d_obs_z; a.obs_zvel = h->d_obs_zvel; a.P = h->d_P; a.row_ptr = h->d_row_ptr; a.chi2 = h->d_chi2;
    a.skip = h->skip_active ? h->d_skip : nullptr;
    a.Hs = h->materialize ? h->d_Hs : nullptr; a.T3 = h->d_T3; a.Xobs = h->d_Xobs; a.obs_pos = h->d_clone_obs; a.gamma = h->d_gamma; a.accept = h->d_accept; a.Rf = h->d_Rf;
    a.sigma2 = h->flags.noise_feature * h->flags.noise_feature;
    a.n = h->n; a.leg = h->flags.leg_dim; a.N = h->N; a.NA = h->NA; a.NAP = h->NAP; a.Mmax = h->Mmax; a.F = h->F;
    a.use_larvio = h->flags.use_larvio; a.use_left = h->flags.use_left_perturbation; a.if_fej = h->flags.if_fej;
    a.estimate_td = h->flags.estimate_td;
    a.ablate = h->feat_ablate;
    return a;
}

static int launch_feature(orcvio_msckf_handle* h, hipStream_t s) {
    if (h->F == 0) return ORCVIO_OK;
    const FeatArgs a = feature_args(h);
    const size_t lds = feat_lds_bytes(h->Mmax, h->NAP, h->N);
    const int npass = (h->NAP + 63) / 64;
    dim3 grid(h->F), block(256);
    switch (npass) {
        case 1: hipLaunchKernelGGL(k_feature<1>, grid, block, lds, s, a); break;
        case 2: hipLaunchKernelGGL(k_feature<2>, grid, block, lds, s, a); break;
        case 3: hipLaunchKernelGGL(k_feature<3>, grid, block, lds, s, a); break;
        case 4: hipLaunchKernelGGL(k_feature<4>, grid, block, lds, s, a); break;
        case 5: hipLaunchKernelGGL(k_feature<5>, grid, block, lds, s, a); break;
        case 6: hipLaunchKernelGGL(k_feature<6>, grid, block, lds, s, a); break;
        case 7: hipLaunchKernelGGL(k_feature<7>, grid, block, lds, s, a); break;
        default: g_last_error = "window too wide for k_feature"; return ORCVIO_ERR_CAPACITY;
    }
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

// The Cholesky of the prior and the feature tracks in one launch (k_front) when everything is co-resident: register
// path (n <= 224), 1 + ceil(F/2) workgroups on the device's CUs, two feature teams + the factorisation's LDS within one
// CU's 160 KB.  Otherwise the caller forks the factorisation to the side stream and launches k_feature.
static bool front_fused_active(const orcvio_msckf_handle* h) {
    if (!h->front_fused || !h->reg_path || h->F < 1) return false;
    if (1 + (h->F + 1) / 2 > h->n_cus) return false;   // one workgroup per CU (by LDS size), all resident at once
    if (h->front_retry_forked) return false;   // a hand-off of the fused launch timed out: this update is re-run on the forked path
    const size_t team = feat_lds_bytes(h->Mmax, h->NAP, h->N);
    const size_t lds = 2 * team > sizeof(double) * POTRF_LDS_DOUBLES ? 2 * team : sizeof(double) * POTRF_LDS_DOUBLES;
    return lds <= (size_t)160 * 1024 && (h->NAP + 63) / 64 <= 4;
}

static int front_row_chunks(const orcvio_msckf_handle* h) {
    const int c = (3 * h->F + 639) / 640;   // eight wavefronts x 80 rows: one batch of loads each
    return c < 1 ? 1 : c;
}
// U = [A; b^T] L_a can assemble A on the fly (k_gemm_asmA) when there are at most four partial Grams: k_front then stops
// after the Grams (one device-wide barrier instead of two, no assembly pass)
static bool front_defers_assembly(const orcvio_msckf_handle* h) { return front_fused_active(h) && front_row_chunks(h) <= 4 && h->NA <= 192; }

// compress_dst != nullptr: the compression (Grams + assembly of A into compress_dst) runs inside the same launch;
// grams_only: ... without the assembly (the caller's next kernel is k_gemm_asmA)
static int launch_front(orcvio_msckf_handle* h, hipStream_t s, double* compress_dst, bool grams_only = false) {
    const FeatArgs a = feature_args(h);
    const double eps = 2.220446049250313e-16;
    FrontPotrfArgs q{h->d_P, h->n, h->n, 8.0 * eps, h->d_RP, h->NP, h->d_DinvP, h->d_info, h->use_factor ? 1 : 0, rev_prior_active(h) ? 1 : 0};
    const size_t team = feat_lds_bytes(h->Mmax, h->NAP, h->N);
    const size_t lds = 2 * team > sizeof(double) * POTRF_LDS_DOUBLES ? 2 * team : sizeof(double) * POTRF_LDS_DOUBLES;
    const int team_doubles = (int)(team / sizeof(double));
    FrontGramArgs g{};
    g.enabled = compress_dst != nullptr ? (grams_only ? 2 : 1) : 0;
    const int t3rows = 3 * h->F;
    g.chunks = front_row_chunks(h);
    h->front_chunks = g.chunks;
    g.rows_per_chunk = round_up((t3rows + g.chunks - 1) / g.chunks, 4);
    g.Gpart = h->d_Gpart; g.S = h->d_S; g.clone_rows = h->d_clone_ptr; g.counter = h->d_sync; g.lost = h->d_info + 8;
    g.A_dst = compress_dst; g.cb0 = h->flags.leg_dim - 15; g.plus = extra_gram(h); g.spin_limit = h->front_spin_limit;
    if (g.enabled && g.chunks > h->gram_chunks_cap) { g_last_error = "launch_front: too many row chunks"; return ORCVIO_ERR_CAPACITY; }
    dim3 grid(1 + (h->F + 1) / 2), block(512);
    // the window width fixes both template arguments: NPASS = ceil(NAP/64) column passes, and enough register slots
    // for the widest matrix of that class (n <= 79 / 143 / 207 / 224)
#define LAUNCH_FRONT(NPS, NSL)                                                                                                  \
    do {                                                                                                                        \
        static bool attr_set = false;                                                                                           \
        if (!attr_set) {                                                                                                        \
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front<NPS, NSL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024))); \
            attr_set = true;                                                                                                    \
        }                                                                                                                       \
        hipLaunchKernelGGL((k_front<NPS, NSL>), grid, block, lds, s, a, q, team_doubles, g);                                    \
    } while (0)
    switch ((h->NAP + 63) / 64) {
        case 1: LAUNCH_FRONT(1, 4); break;
        case 2: LAUNCH_FRONT(2, 8); break;
        case 3: LAUNCH_FRONT(3, 16); break;
        case 4: LAUNCH_FRONT(4, 16); break;
        default: g_last_error = "launch_front: window too wide"; return ORCVIO_ERR_CAPACITY;
    }
#undef LAUNCH_FRONT
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

static int asm_dbg() {   // diagnostics switch of the assembly kernels, read once (getenv on the hot path)
    static const int v = [] { const char* e = getenv("ORCVIO_ASM_DBG"); return e ? atoi(e) : 0; }();
    return v;
}

// compression: A = X^T X - T3^T T3  (sparse rows summed per clone, dense rows by MFMA Gram)
static int launch_gram(orcvio_msckf_handle* h, hipStream_t s) {
    const int nb = h->NAP / 16, ntiles = nb * (nb + 1) / 2;
    if (h->F == 0) {
        HIPCHK(hipMemsetAsync(h->d_Gpart, 0, sizeof(double) * (size_t)h->NAP * h->NAP, s));
        return ORCVIO_OK;
    }
    // grid.y < chunks: T3 tiles; grid.y == chunks: one workgroup per clone for the sparse rows
    dim3 grid(ntiles > h->N ? ntiles : h->N, h->chunks + 1), block(1024);
    hipLaunchKernelGGL(k_gram_pair, grid, block, 0, s, h->d_T3, 3 * h->F, h->NAP, h->rows_per_chunk, h->chunks, h->d_Gpart, h->d_Xobs,
                       h->d_S, (const int*)h->d_clone_ptr, h->N);
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

static int launch_assemble(orcvio_msckf_handle* h, hipStream_t s, double* dst) {
    const int total = h->NAP * h->NAP;
    hipLaunchKernelGGL(k_assemble_A, dim3((total + 255) / 256), dim3(256), 0, s, h->d_S, h->F > 0 ? h->N : 0,
                       h->flags.leg_dim - 15, h->NA, h->NAP, h->d_Gpart, h->chunks, (size_t)total, dst,
                       asm_dbg(), extra_gram(h));
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

// d_A on demand (optional outputs, tests) after an update whose k_front left the Grams only
static int assemble_deferred(orcvio_msckf_handle* h, hipStream_t s) {
    if (!h->A_deferred) return ORCVIO_OK;
    const int total = h->NAP * h->NAP;
    hipLaunchKernelGGL(k_assemble_A, dim3((total + 255) / 256), dim3(256), 0, s, h->d_S, h->F > 0 ? h->N : 0,
                       h->flags.leg_dim - 15, h->NA, h->NAP, h->d_Gpart, h->front_chunks, (size_t)total, h->d_A, 0,
                       extra_gram(h));
    HIPCHK(hipGetLastError());
    h->A_deferred = false;
    return ORCVIO_OK;
}

// stride: doubles between consecutive blocks (0: packed); meta0: the status words behind the first block of a sharded update
// (nullptr: none) -- the launch leaves [first failing rank + 1, its status, total dof, total accepted rows] in info[9..12]
static int launch_reduce(orcvio_msckf_handle* h, hipStream_t s, const double* parts, int nparts, double* dst, size_t stride = 0,
                         const double* meta0 = nullptr) {
    const int total = h->NAP * h->NAP;
    hipLaunchKernelGGL(k_gram_reduce, dim3((total + 255) / 256), dim3(256), 0, s, parts, nparts, stride ? stride : (size_t)total, h->NAP, dst,
                       meta0, h->d_info + 9);
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

// k_potrf_reg / k_potrf_solve write the upper tiles of d_RP / d_RM only and rely on the strictly-lower tiles being zero
// (the consumers read the factors as dense matrices).  They are: zeroed at creation, and again whenever the leading
// dimension or the factorisation path changes -- a change of window size, not a per-update event.
static int factor_layout_clean(orcvio_msckf_handle* h) {
    if (h->NP == h->clean_NP && (int)h->reg_path == h->clean_path) return ORCVIO_OK;
    HIPCHK(hipDeviceSynchronize());   // an earlier update may still be reading the factors
    const size_t bytes = sizeof(double) * (size_t)h->NP_max * h->NP_max;
    HIPCHK(hipMemset(h->d_RP, 0, bytes));
    HIPCHK(hipMemset(h->d_RM, 0, bytes));
    HIPCHK(hipDeviceSynchronize());
    h->clean_NP = h->NP;
    h->clean_path = (int)h->reg_path;
    return ORCVIO_OK;
}

// Cholesky X = L L^T.  reg path: upper factor R (L = R^T) written to `out` (ld = NP), L(i,j) = out[j*NP + i];
// LDS-panel path: X copied to `out`, factored in place (lower), L(i,j) = out[i*NP + j].
static int launch_potrf(orcvio_msckf_handle* h, hipStream_t s, const double* X, int ldx, int nn, double tol_rel, double* out,
                        double* Dinv, int* info, int rev = 0) {
    const int NP = h->NP;
    if (h->reg_path) {
        const int nb = (nn + 15) / 16, noff = nb * (nb - 1) / 2;
        const int need = potrf_slots_needed(nb);   // wave 0 keeps the diagonal tiles in LDS, the workers the rest in registers
        (void)noff;
        // zero_lower = 0: `out` is d_RP / d_RM, whose strictly-lower tiles factor_layout_clean() keeps zero
#define LAUNCH_PR(NS) hipLaunchKernelGGL(k_potrf_reg<NS>, dim3(1), dim3(512), 0, s, X, ldx, nn, tol_rel, out, NP, Dinv, info, \
                                         (unsigned long long*)nullptr, (size_t)0, (size_t)0, (size_t)0, 0, 0, 0, rev)
        if (need <= 4) LAUNCH_PR(4);
        else if (need <= 8) LAUNCH_PR(8);
        else if (need <= 12) LAUNCH_PR(12);
        else LAUNCH_PR(16);
#undef LAUNCH_PR
    } else {
        HIPCHK(hipMemcpy2DAsync(out, sizeof(double) * NP, X, sizeof(double) * ldx, sizeof(double) * nn, nn, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_potrf, dim3(1), dim3(1024), 0, s, out, nn, NP, tol_rel, Dinv, info);
    }
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

static inline void factor_strides(const orcvio_msckf_handle* h, long& sLi, long& sLj) {
    if (h->reg_path) { sLi = 1; sLj = h->NP; } else { sLi = h->NP; sLj = 1; }
}

static int launch_gemm(hipStream_t s, const double* A, long sAi, long sAk, const double* B, long sBk, long sBj, int M, int N,
                       int K, double alpha, double diag_add, int upper_only, double* C, long sCi, long sCj, int* clear = nullptr) {
    const int tiles = ((M + 15) / 16) * ((N + 15) / 16);
    hipLaunchKernelGGL(k_gemm, dim3(tiles), dim3(256), 0, s, A, sAi, sAk, B, sBk, sBj, M, N, K, alpha, diag_add,
                       upper_only, C, sCi, sCj, (const double*)nullptr, clear);
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

static int launch_trsm(orcvio_msckf_handle* h, hipStream_t s, const double* L, const double* Dinv, int nn, const double* B1,
                       long sB1i, long sB1c, int nc1, const double* bx, long sbx, double* Z, int ldz) {
    const int ncols = nc1 + (bx ? 1 : 0);
    if (h->reg_path) {   // L = R^T, R row-major with ld NP: LDS-staged panels
        const int nwave = (ncols + 15) / 16;
        hipLaunchKernelGGL(k_trsm_lds, dim3((nwave + 3) / 4), dim3(256), 0, s, L, h->NP, Dinv, nn, B1, sB1i, sB1c, nc1, bx, sbx, Z, ldz);
    } else {
        long sLi, sLj;
        factor_strides(h, sLi, sLj);
        hipLaunchKernelGGL(k_trsm_rl<TRSM_MAXBLK>, dim3((ncols + 15) / 16), dim3(64), 0, s, L, sLi, sLj, Dinv, nn, B1, sB1i, sB1c,
                           nc1, bx, sbx, Z, ldz);
    }
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

static inline bool fused_solve_active(const orcvio_msckf_handle* h) { return h->reg_path && h->fused_solve; }

// stages of the square-root Kalman solve (see msckf_kernels.hpp)
enum { ST_POTRF_P = 0, ST_FORM_U, ST_FORM_M, ST_POTRF_M, ST_TRSM, ST_FINISH, ST_COUNT };

static int launch_solve_stage(orcvio_msckf_handle* h, hipStream_t s, int stage) {
    // n = states (rows of the prior's factor), kf = columns of that factor = dimension of M (kf == n unless the resident
    // factor of an earlier update of the frame is used)
    const int NA = h->NA, NAP = h->NAP, n = h->n, kf = h->kf, NP = h->NP, ldz = h->ldz;
    const double sigma2 = h->flags.noise_feature * h->flags.noise_feature;
    const double eps = 2.220446049250313e-16;
    const PriorFactor pf = prior_factor(h);
    const long sLi = pf.sLi, sLj = pf.sLj;
    const double* La = pf.base + 15 * sLi;   // L_a(k, j) = Lf(15 + k, j)
    const int kfa = kf - h->tail;            // columns of the factor that are not zero in the active rows: dimension of M' (M = diag(M', s2 I))
    switch (stage) {
        case ST_POTRF_P:   // P = Lf Lf^T
            if (h->use_factor) return ORCVIO_OK;   // the factor is resident
            return launch_potrf(h, s, h->d_P, n, n, 8.0 * eps, h->d_RP, h->d_DinvP, h->d_info, rev_prior_active(h) ? 1 : 0);
        case ST_FORM_U:    // U[(NA+1) x kf] = [A; b^T] * L_a
            if (h->A_deferred) {   // A = scatter(S) - sum Gpart assembled inside the product (k_front left the Grams only)
                AsmArgs aa{h->d_S, h->N, h->flags.leg_dim - 15, NA, NAP, h->d_Gpart, h->front_chunks, (size_t)NAP * NAP, asm_dbg(),
                           extra_gram(h)};
                const int tiles = ((NA + 1 + 15) / 16) * ((kfa + 15) / 16);
                hipLaunchKernelGGL(k_gemm_asmA, dim3(tiles), dim3(256), 0, s, aa, La, sLi, sLj, NA + 1, kfa, NA, h->d_U, (long)NP, 1L, (int*)nullptr);
                HIPCHK(hipGetLastError());
                return ORCVIO_OK;
            }
            return launch_gemm(s, h->d_A, NAP, 1, La, sLi, sLj, NA + 1, kfa, NA, 1.0, 0.0, 0, h->d_U, NP, 1);
        case ST_FORM_M:    // M = s2 I + L_a^T U[0:NA]   (upper tiles)
            // (the register-resident Cholesky reads the upper tiles only; the LDS-panel fallback factors the lower triangle in place)
            return launch_gemm(s, La, sLj, sLi, h->d_U, NP, 1, kfa, kfa, NA, 1.0, sigma2, h->reg_path ? 1 : 0, h->d_M, NP, 1, h->d_flag);
        case ST_POTRF_M:
            if (fused_solve_active(h)) {   // chol(M) + Z = L_M^-1 [Lf^T | g] in one launch (solver workgroups trail the factorisation)
                const int nbm = (kfa + 15) / 16, need = potrf_slots_needed(nbm);
                const int ncb = (n + 1 + 15) / 16;
                const dim3 grid(1 + (ncb + SOLVE_WPB - 1) / SOLVE_WPB), block(512);
                const double* g = h->d_U + (size_t)NA * NP;
#define LAUNCH_PS(NS) hipLaunchKernelGGL(k_potrf_solve<NS>, grid, block, 0, s, h->d_M, NP, kfa, 0.0, h->d_RM, NP, h->d_DinvM, h->d_info + 2, \
                                         h->d_flag, h->d_info + 8, pf.base, sLj, sLi, n, g, 1L, h->d_Z, ldz, h->tail, 1.0 / h->flags.noise_feature)
                if (need <= 4) LAUNCH_PS(4);
                else if (need <= 8) LAUNCH_PS(8);
                else if (need <= 12) LAUNCH_PS(12);
                else LAUNCH_PS(16);
#undef LAUNCH_PS
                HIPCHK(hipGetLastError());
                return ORCVIO_OK;
            }
            return launch_potrf(h, s, h->d_M, NP, kf, 0.0, h->d_RM, h->d_DinvM, h->d_info + 2);
        case ST_TRSM:      // Z = L_M^-1 [Lf^T | g],  g = U[NA][:]
            if (fused_solve_active(h)) return ORCVIO_OK;   // done inside k_potrf_solve
            return launch_trsm(h, s, h->d_RM, h->d_DinvM, kf, pf.base, sLj, sLi, n, h->d_U + (size_t)NA * NP, 1, h->d_Z, ldz);
        case ST_FINISH: {
            const int nb = (n + 1 + 15) / 16, tiles = nb * (nb + 1) / 2;
            ObjGate gate;
            gate.fail = h->d_info + 2;   // chol(M) of this update met a non-positive pivot: P+ = P, dx = 0
            if (h->objects_mode) {
                gate.rr = h->d_A + (size_t)NA * h->NAP + NA; gate.thr = h->obj_thr;
                gate.gamma = h->d_obj_gamma; gate.accept = h->d_obj_accept; gate.gamma_out = h->d_gamma; gate.accept_out = h->d_accept;
            }
            hipLaunchKernelGGL(k_finish_sqrt, dim3(tiles), dim3(256), 0, s, h->d_Z, ldz, n, kf, sigma2, h->d_Pout, h->d_dx, gate, h->d_P, 6 * h->n_nui);
            HIPCHK(hipGetLastError());
            return ORCVIO_OK;
        }
        default: return ORCVIO_ERR_INVALID;
    }
}

// fork: Cholesky of the prior on the side stream (depends on P only)
static int launch_prior_fork(orcvio_msckf_handle* h, hipStream_t s) {
    h->prior_forked = !h->use_factor;
    if (h->use_factor) return ORCVIO_OK;   // the prior's factor is resident: nothing to fork, nothing to join
    HIPCHK(hipEventRecord(h->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    int rc = launch_solve_stage(h, h->side, ST_POTRF_P);   // writes d_info[0..1] itself
    if (rc != ORCVIO_OK) return rc;
    HIPCHK(hipEventRecord(h->ev_side, h->side));
    return ORCVIO_OK;
}

static int launch_solve_tail(orcvio_msckf_handle* h, hipStream_t s) {
    if (h->prior_forked) HIPCHK(hipStreamWaitEvent(s, h->ev_side, 0));   // join the Cholesky of the prior
    int rc = ORCVIO_OK;
    for (int st = ST_FORM_U; st < ST_COUNT && rc == ORCVIO_OK; ++st) rc = launch_solve_stage(h, s, st);
    return rc;
}

static hipStream_t pick_stream(orcvio_msckf_handle* h, void* stream) { return stream ? (hipStream_t)stream : h->stream; }

// ---- launch graphs ------------------------------------------------------------------------------------
// A sequence of launches can be replayed from a captured hipGraph: same kernels, same arguments, fewer host calls and
// tighter dispatch.  Capturing costs several hundred microseconds, so a slot captures only when the same launch
// signature (sizes, flags, options, pointers, stream) shows up twice in a row -- a caller that replays one shape (the
// benchmark, a fixed-size window) gets the graph, a caller whose track count changes every frame gets plain launches.
static unsigned long long launch_signature(const orcvio_msckf_handle* h, hipStream_t s, const void* p0, long extra) {
    unsigned long long sig = 1469598103934665603ull;
    auto mix = [&](unsigned long long v) { sig = (sig ^ v) * 1099511628211ull; };
    mix(h->N); mix(h->F); mix(h->nobs); mix(h->Mmax); mix(h->chunks); mix(h->s_chunks); mix(h->rows_per_chunk);
    mix(h->flags.leg_dim); mix(h->flags.use_larvio); mix(h->flags.use_left_perturbation); mix(h->flags.if_fej);
    mix(h->flags.estimate_td); mix(h->materialize); mix(h->skip_active); mix(h->fused_solve); mix(h->front_fused); mix(h->feat_ablate); mix(h->ekf_F); mix(h->ekf_mode); mix(h->n_extra); mix(h->n_nui); mix(h->dense_rows);
    unsigned long long bits;
    double sg = h->flags.noise_feature;
    std::memcpy(&bits, &sg, 8); mix(bits);
    double cp = h->flags.chi2_prob;
    std::memcpy(&bits, &cp, 8); mix(bits);
    mix(h->front_retry_forked); mix(h->front_spin_limit); mix(h->use_factor); mix(h->kf); mix(h->fac_ld); mix(h->graph_epoch); mix(h->ekf_idp); mix(h->ekf_eval); mix(h->ekf_cap); mix(h->dense_cap);
    mix((unsigned long long)(size_t)h->d_ekf_i); mix((unsigned long long)(size_t)h->d_ekf_d); mix((unsigned long long)(size_t)h->d_ekf_E);
    mix((unsigned long long)(size_t)h->d_slam); mix((unsigned long long)(size_t)h->d_dense); mix((unsigned long long)(size_t)h->d_Gekf);
    mix((unsigned long long)(size_t)h->d_Hs); mix((unsigned long long)(size_t)h->d_P);
    // the resident square-root factor is double-buffered and every user of it is a captured kernel argument (prior_factor):
    // its address belongs to the signature (a same-shape update after cov_commit replayed the OTHER buffer)
    mix(h->use_factor ? (unsigned long long)(size_t)h->d_Sres : 0ull);
    mix(h->tail); mix(rev_prior_active(h));
    mix((unsigned long long)(size_t)s); mix((unsigned long long)(size_t)p0); mix((unsigned long long)extra);
    return sig;
}

static int run_with_graph(orcvio_msckf_handle* h, orcvio_msckf_handle::GraphSlot& slot, unsigned long long sig, hipStream_t s,
                          const std::function<int(bool)>& enqueue) {
    typedef orcvio_msckf_handle::GraphSlot::Way Way;
    if (h->use_graph && s != nullptr) {
        Way* hit = nullptr;
        for (Way& w : slot.way)
            if (w.exec && w.sig == sig) { hit = &w; break; }
        if (!hit) {
            bool seen = false;
            for (unsigned long long v : slot.seen) seen = seen || (v == sig && sig != 0);
            if (seen) {   // second time: capture into a free way, or over the least recently used one
                Way* dst = &slot.way[0];
                for (Way& w : slot.way) {
                    if (!w.exec) { dst = &w; break; }
                    if (w.used < dst->used) dst = &w;
                }
                if (dst->exec) { (void)hipGraphExecDestroy(dst->exec); dst->exec = nullptr; }
                if (dst->graph) { (void)hipGraphDestroy(dst->graph); dst->graph = nullptr; }
                if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
                    const int rc_c = enqueue(true);
                    hipGraph_t g = nullptr;
                    const hipError_t e_end = hipStreamEndCapture(s, &g);
                    if (rc_c == ORCVIO_OK && e_end == hipSuccess && g && hipGraphInstantiate(&dst->exec, g, nullptr, nullptr, 0) == hipSuccess) {
                        dst->graph = g;
                        dst->sig = sig;
                        hit = dst;
                    } else {
                        if (g) (void)hipGraphDestroy(g);
                        dst->exec = nullptr;
                        (void)hipGetLastError();
                        h->use_graph = false;   // capture is not available here: plain launches from now on
                    }
                } else {
                    (void)hipGetLastError();
                    h->use_graph = false;
                }
            } else {
                slot.seen[slot.seen_at] = sig;
                slot.seen_at = (slot.seen_at + 1) & 7;
            }
        }
        if (hit) {
            hit->used = ++slot.tick;
            HIPCHK(hipGraphLaunch(hit->exec, s));
            return ORCVIO_OK;
        }
    }
    return enqueue(false);
}

// this rank's part of a sharded update: the Cholesky of the prior goes to the side stream with plain launches (it is
// joined by run_finish, so it overlaps the feature kernels AND the collective), the rest replays from a graph
static int run_local_impl(orcvio_msckf_handle* h, hipStream_t s, double* dst) {
    h->last_stream = s;
    { const int re = launch_ekf(h, s); if (re != ORCVIO_OK) return re; }
    if (front_fused_active(h)) { h->prior_forked = false; return launch_front(h, s, dst); }   // one launch: tracks, compression, and the prior's factor
    int rc = launch_prior_fork(h, s);
    if (rc != ORCVIO_OK) return rc;
    return run_with_graph(h, h->g_local, launch_signature(h, s, dst, 0), s, [&](bool) {
        int r = launch_feature(h, s);
        if (r == ORCVIO_OK) r = launch_gram(h, s);
        if (r == ORCVIO_OK) r = launch_assemble(h, s, dst);
        return r;
    });
}

int32_t orcvio_msckf_run_local(orcvio_msckf_handle* h, void* stream) {
    if (!h || !h->uploaded || h->pw_missing) { g_last_error = "run_local: nothing uploaded (or positions missing)"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    return run_local_impl(h, pick_stream(h, stream), h->d_Ab);
}

int32_t orcvio_msckf_run_local_to(orcvio_msckf_handle* h, double* d_dst, void* stream) {
    if (!h || !h->uploaded || !d_dst || h->pw_missing) { g_last_error = "run_local_to: invalid"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    return run_local_impl(h, pick_stream(h, stream), d_dst);
}

int32_t orcvio_msckf_block_ptr(orcvio_msckf_handle* h, double** d_block, int64_t* n_elems) {
    if (!h || !h->uploaded || !d_block || !n_elems) { g_last_error = "block_ptr: invalid"; return ORCVIO_ERR_INVALID; }
    *d_block = h->d_Ab;
    *n_elems = (int64_t)h->NAP * h->NAP;
    return ORCVIO_OK;
}

static int run_finish_impl(orcvio_msckf_handle* h, const double* d_blocks, int n_blocks, size_t stride, const double* meta0, hipStream_t s) {
    h->last_stream = s;
    if (h->dl_pending) { HIPCHK(hipStreamSynchronize(h->dl_stream)); h->dl_pending = false; }
    // the Cholesky of the prior was forked by run_local: join it here (outside the captured part)
    if (h->prior_forked) HIPCHK(hipStreamWaitEvent(s, h->ev_side, 0));
    h->A_deferred = false;   // d_A is the sum of the gathered blocks
    unsigned long long sig = launch_signature(h, s, d_blocks, n_blocks);
    sig = (sig ^ (unsigned long long)stride) * 1099511628211ull;
    sig = (sig ^ (unsigned long long)(size_t)meta0) * 1099511628211ull;
    int rc = run_with_graph(h, h->g_finish, sig, s, [&](bool) {
        int r = launch_reduce(h, s, d_blocks, n_blocks, h->d_A, stride, meta0);   // rank-ordered sum of the gathered blocks
        for (int st = ST_FORM_U; st < ST_COUNT && r == ORCVIO_OK; ++st) r = launch_solve_stage(h, s, st);
        return r;
    });
    if (rc == ORCVIO_OK) { h->ran = true; h->last_update_objects = false; h->last_run_kind = 1; h->last_sharded = false; }
    return rc;
}

int32_t orcvio_msckf_run_finish(orcvio_msckf_handle* h, const double* d_blocks, int32_t n_blocks, void* stream) {
    if (!h || !h->uploaded || !d_blocks || n_blocks < 1) { g_last_error = "run_finish: invalid"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    return run_finish_impl(h, d_blocks, n_blocks, 0, nullptr, pick_stream(h, stream));
}

// EKF-SLAM rows of this upload: gate every feature against the prior (2 degrees of freedom), accepted rows -> dense
// [H | r] rows -> their Gram, which assembly adds to the compressed block (ekf_rows.hpp)
static int launch_ekf(orcvio_msckf_handle* h, hipStream_t s) {
    if (h->ekf_F <= 0 && h->dense_rows <= 0) return ORCVIO_OK;
    const int F = h->ekf_F, cap = h->ekf_cap;
    const int nb = h->NAP / 16, ntiles = nb * (nb + 1) / 2;
    if (F > 0) {
        HIPCHK(hipMemsetAsync(h->d_ekf_E, 0, sizeof(double) * (size_t)2 * F * h->NAP, s));
        if (h->ekf_eval) {   // measurementJacobian_ekf_{3,1}didp on the device: the compact row blocks from the SLAM features
            EkfEvalArgs e;
            e.F = F; e.idp_dim = h->ekf_idp; e.if_fej = h->flags.if_fej; e.poses = h->d_poses;
            e.anchor = h->d_ekf_i; e.state = h->d_ekf_i + cap;
            e.param = h->d_slam; e.inv_depth = e.param + (size_t)3 * cap; e.p_w = e.inv_depth + cap; e.p_fej = e.p_w + (size_t)3 * cap;
            e.z = e.p_fej + (size_t)3 * cap;
            e.He = h->d_ekf_d; e.Ha = e.He + (size_t)12 * cap; e.Hx = e.Ha + (size_t)12 * cap; e.Hf = e.Hx + (size_t)12 * cap;
            e.r = e.Hf + (size_t)6 * cap + (size_t)2 * cap;
            hipLaunchKernelGGL(k_ekf_eval, dim3((F + 63) / 64), dim3(64), 0, s, e);
        }
        EkfGateArgs a;
        a.F = F; a.idp_dim = h->ekf_idp; a.n = h->n; a.leg = h->flags.leg_dim; a.N = h->N; a.NA = h->NA; a.NAP = h->NAP;
        a.estimate_td = h->flags.estimate_td; a.n_nui = h->n_nui;
        a.anchor = h->d_ekf_i; a.state = h->d_ekf_i + cap; a.slot = h->d_ekf_i + 2 * cap;
        a.He = h->d_ekf_d; a.Ha = a.He + (size_t)12 * cap; a.Hx = a.Ha + (size_t)12 * cap; a.Hf = a.Hx + (size_t)12 * cap;
        a.zvel = a.Hf + (size_t)6 * cap; a.r = a.zvel + (size_t)2 * cap;
        a.P = h->d_P; a.sigma2 = h->flags.noise_feature * h->flags.noise_feature;
        a.threshold = orcvio_msckf_chi2_quantile(2, h->flags.chi2_prob);
        a.E = h->d_ekf_E; a.gamma = h->d_ekf_gamma; a.accept = h->d_ekf_accept;
        hipLaunchKernelGGL(k_ekf_gate, dim3(F), dim3(64), 0, s, a);
        hipLaunchKernelGGL(k_gram, dim3((ntiles + 3) / 4, 1), dim3(256), 0, s, (const double*)h->d_ekf_E, 2 * F, h->NAP, round_up(2 * F, 8),
                           h->d_Gekf, (const int*)nullptr);
    }
    if (h->dense_rows > 0) {   // their Gram into the scratch behind d_Gekf, then summed (fixed order)
        double* G2 = h->d_Gekf + (size_t)h->NAP_max * h->NAP_max;
        hipLaunchKernelGGL(k_gram, dim3((ntiles + 3) / 4, 1), dim3(256), 0, s, (const double*)h->d_dense, h->dense_rows, h->NAP,
                           round_up(h->dense_rows, 8), F > 0 ? G2 : h->d_Gekf, (const int*)nullptr);
        if (F > 0) {
            const int total = h->NAP * h->NAP;
            hipLaunchKernelGGL(k_add_inplace, dim3((total + 255) / 256), dim3(256), 0, s, h->d_Gekf, (const double*)G2, total);
        }
    }
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

static int enqueue_update(orcvio_msckf_handle* h, hipStream_t s) {
    { const int re = launch_ekf(h, s); if (re != ORCVIO_OK) return re; }
    // (the other way round -- prior on the origin stream, feature branch forked -- measured 15 us slower)
    h->A_deferred = front_defers_assembly(h);
    if (front_fused_active(h)) {   // one stream, no fork: the prior is factored by workgroup 0 of the feature launch
        // ... and the compression behind the tracks, under the factorisation
        h->prior_forked = false;
        int rc = launch_front(h, s, h->d_A, h->A_deferred);
        for (int st = ST_FORM_U; st < ST_COUNT && rc == ORCVIO_OK; ++st) rc = launch_solve_stage(h, s, st);
        return rc;
    }
    int rc = launch_prior_fork(h, s);
    if (rc == ORCVIO_OK) rc = launch_feature(h, s);
    if (rc == ORCVIO_OK) rc = launch_gram(h, s);
    if (rc == ORCVIO_OK) rc = launch_assemble(h, s, h->d_A);
    if (rc == ORCVIO_OK) rc = launch_solve_tail(h, s);
    return rc;
}

// ---- EKF-SLAM rows ---------------------------------------------------------------------
static int ekf_reserve(orcvio_msckf_handle* h, int F) {
    if (F <= h->ekf_cap) return ORCVIO_OK;
    HIPCHK(hipDeviceSynchronize());
    void* old[] = {h->d_ekf_i, h->d_ekf_d, h->d_ekf_E, h->d_ekf_gamma, h->d_ekf_accept, h->d_slam};
    for (void* q : old) if (q) (void)hipFree(q);
    const int cap = round_up(F, 32);
    HIPCHK(hipMalloc(&h->d_ekf_i, sizeof(int) * 3 * cap));
    HIPCHK(hipMalloc(&h->d_ekf_d, sizeof(double) * 46 * cap));
    HIPCHK(hipMalloc(&h->d_ekf_E, sizeof(double) * (size_t)2 * cap * h->NAP_max));
    HIPCHK(hipMalloc(&h->d_ekf_gamma, sizeof(double) * cap));
    HIPCHK(hipMalloc(&h->d_ekf_accept, sizeof(int) * cap));
    HIPCHK(hipMalloc(&h->d_slam, sizeof(double) * 12 * cap));
    if (!h->d_Gekf) HIPCHK(hipMalloc(&h->d_Gekf, sizeof(double) * (size_t)2 * h->NAP_max * h->NAP_max));
    h->ekf_cap = cap;
    h->graph_epoch++;   // captured graphs hold the freed pointers
    return ORCVIO_OK;
}

int32_t orcvio_msckf_upload_slam_features(orcvio_msckf_handle* h, const orcvio_msckf_slam_features* ft) {
    if (!h || !ft || !h->uploaded) { g_last_error = "upload_slam_features: upload the window first"; return ORCVIO_ERR_INVALID; }
    if (!h->ekf_mode) { g_last_error = "upload_slam_features: set ORCVIO_OPT_EKF_ROWS before the upload"; return ORCVIO_ERR_INVALID; }
    const int F = ft->n_features, d = ft->idp_dim;
    if (F < 0 || (d != 1 && d != 3)) { g_last_error = "upload_slam_features: idp_dim must be 1 or 3"; return ORCVIO_ERR_INVALID; }
    if (F > 0 && (!ft->anchor || !ft->state || !ft->slot || !ft->param || !ft->p_w || !ft->z || (d == 1 && !ft->inv_depth) ||
                  (h->flags.if_fej && !ft->p_fej) || (h->flags.estimate_td && !ft->z_vel))) { g_last_error = "upload_slam_features: null array"; return ORCVIO_ERR_INVALID; }
    for (int f = 0; f < F; ++f) {
        if (ft->anchor[f] < 0 || ft->anchor[f] >= h->N + h->n_nui || ft->state[f] < 0 || ft->state[f] >= h->N) { g_last_error = "upload_slam_features: clone index outside the window"; return ORCVIO_ERR_INVALID; }
        if (ft->slot[f] < 0 || d * (ft->slot[f] + 1) > h->n_extra - 6 * h->n_nui) { g_last_error = "upload_slam_features: feature slot outside the extra states"; return ORCVIO_ERR_INVALID; }
    }
    HIPCHK(hipSetDevice(h->device));
    { const int rc = ekf_reserve(h, F); if (rc != ORCVIO_OK) return rc; }
    h->ekf_F = F; h->ekf_idp = d; h->ekf_eval = true;
    if (F == 0) return ORCVIO_OK;
    const int cap = h->ekf_cap;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->d_ekf_i, ft->anchor, sizeof(int) * F, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_ekf_i + cap, ft->state, sizeof(int) * F, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_ekf_i + 2 * cap, ft->slot, sizeof(int) * F, hipMemcpyHostToDevice));
    double* q = h->d_slam;
    HIPCHK(hipMemcpy(q, ft->param, sizeof(double) * 3 * F, hipMemcpyHostToDevice)); q += (size_t)3 * cap;
    if (ft->inv_depth) HIPCHK(hipMemcpy(q, ft->inv_depth, sizeof(double) * F, hipMemcpyHostToDevice));
    q += cap;
    HIPCHK(hipMemcpy(q, ft->p_w, sizeof(double) * 3 * F, hipMemcpyHostToDevice)); q += (size_t)3 * cap;
    if (ft->p_fej) HIPCHK(hipMemcpy(q, ft->p_fej, sizeof(double) * 3 * F, hipMemcpyHostToDevice));
    q += (size_t)3 * cap;
    HIPCHK(hipMemcpy(q, ft->z, sizeof(double) * 2 * F, hipMemcpyHostToDevice));
    if (ft->z_vel) HIPCHK(hipMemcpy(h->d_ekf_d + (size_t)42 * cap, ft->z_vel, sizeof(double) * 2 * F, hipMemcpyHostToDevice));   // (the gate's z_vel slot)
    return ORCVIO_OK;
}

int32_t orcvio_msckf_upload_ekf_rows(orcvio_msckf_handle* h, const orcvio_msckf_ekf_rows* rows) {
    if (!h || !rows || !h->uploaded) { g_last_error = "upload_ekf_rows: upload the window first"; return ORCVIO_ERR_INVALID; }
    if (!h->ekf_mode) { g_last_error = "upload_ekf_rows: set ORCVIO_OPT_EKF_ROWS before the upload"; return ORCVIO_ERR_INVALID; }
    const int F = rows->n_features, d = rows->idp_dim;
    if (F < 0 || (d != 1 && d != 3)) { g_last_error = "upload_ekf_rows: idp_dim must be 1 or 3"; return ORCVIO_ERR_INVALID; }
    if (F > 0 && (!rows->anchor || !rows->state || !rows->slot || !rows->H_e || !rows->H_a || !rows->H_x || !rows->H_f || !rows->r ||
                  (h->flags.estimate_td && !rows->z_vel))) { g_last_error = "upload_ekf_rows: null array"; return ORCVIO_ERR_INVALID; }
    for (int f = 0; f < F; ++f) {
        if (rows->anchor[f] < 0 || rows->anchor[f] >= h->N + h->n_nui || rows->state[f] < 0 || rows->state[f] >= h->N) { g_last_error = "upload_ekf_rows: clone index outside the window"; return ORCVIO_ERR_INVALID; }
        if (rows->slot[f] < 0 || d * (rows->slot[f] + 1) > h->n_extra - 6 * h->n_nui) { g_last_error = "upload_ekf_rows: feature slot outside the extra states"; return ORCVIO_ERR_INVALID; }
    }
    HIPCHK(hipSetDevice(h->device));
    { const int rc = ekf_reserve(h, F); if (rc != ORCVIO_OK) return rc; }
    h->ekf_F = F; h->ekf_idp = d; h->ekf_eval = false;
    if (F == 0) return ORCVIO_OK;
    const int cap = h->ekf_cap;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->d_ekf_i, rows->anchor, sizeof(int) * F, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_ekf_i + cap, rows->state, sizeof(int) * F, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_ekf_i + 2 * cap, rows->slot, sizeof(int) * F, hipMemcpyHostToDevice));
    double* q = h->d_ekf_d;
    HIPCHK(hipMemcpy(q, rows->H_e, sizeof(double) * 12 * F, hipMemcpyHostToDevice)); q += (size_t)12 * cap;
    HIPCHK(hipMemcpy(q, rows->H_a, sizeof(double) * 12 * F, hipMemcpyHostToDevice)); q += (size_t)12 * cap;
    HIPCHK(hipMemcpy(q, rows->H_x, sizeof(double) * 12 * F, hipMemcpyHostToDevice)); q += (size_t)12 * cap;
    HIPCHK(hipMemcpy(q, rows->H_f, sizeof(double) * 2 * d * F, hipMemcpyHostToDevice)); q += (size_t)6 * cap;
    if (rows->z_vel) HIPCHK(hipMemcpy(q, rows->z_vel, sizeof(double) * 2 * F, hipMemcpyHostToDevice));
    q += (size_t)2 * cap;
    HIPCHK(hipMemcpy(q, rows->r, sizeof(double) * 2 * F, hipMemcpyHostToDevice));
    return ORCVIO_OK;
}

static int dense_reserve(orcvio_msckf_handle* h, int rows);
int32_t orcvio_msckf_upload_dense_rows(orcvio_msckf_handle* h, int32_t n_rows, const double* H, const double* r) {
    if (!h || !h->uploaded || n_rows < 0 || (n_rows > 0 && (!H || !r))) { g_last_error = "upload_dense_rows: upload the window first"; return ORCVIO_ERR_INVALID; }
    if (h->n_extra > 0 && !h->ekf_mode) { g_last_error = "upload_dense_rows: with extra states set ORCVIO_OPT_EKF_ROWS before the upload"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    h->dense_rows = 0;   // (these rows come first; orcvio_msckf_upload_new_features appends)
    { const int rc = dense_reserve(h, n_rows); if (rc != ORCVIO_OK) return rc; }
    if (!h->d_Gekf) HIPCHK(hipMalloc(&h->d_Gekf, sizeof(double) * (size_t)2 * h->NAP_max * h->NAP_max));
    h->dense_rows = n_rows;
    if (n_rows == 0) return ORCVIO_OK;
    const int n = h->n, NA = h->NA, NAP = h->NAP;
    for (int i = 0; i < n_rows; ++i)
        for (int c = 0; c < 15; ++c)
            if (H[(size_t)i * n + c] != 0.0) { g_last_error = "upload_dense_rows: the first 15 columns (IMU state) must be zero"; return ORCVIO_ERR_INVALID; }
    std::vector<double> st((size_t)n_rows * NAP, 0.0);   // [H(:, 15:15+NA) | r | 0]
    for (int i = 0; i < n_rows; ++i) {
        std::memcpy(&st[(size_t)i * NAP], H + (size_t)i * n + 15, sizeof(double) * NA);
        st[(size_t)i * NAP + NA] = r[i];
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->d_dense, st.data(), sizeof(double) * st.size(), hipMemcpyHostToDevice));
    return ORCVIO_OK;
}

// Schmidt-EKF: the poses of the nuisance states (state_server.nui_imu_states: clones that left the window but stay in state_cov as
// nuisance parameters, src/orcvio.cpp:2881-2920), in the pose slots behind the window's: SLAM features may be anchored there
// (anchor index N + j, :1247-1256, :1591-1606).  After orcvio_msckf_upload, with ORCVIO_OPT_SCHMIDT_STATES = nui->n_clones.
int32_t orcvio_msckf_upload_nuisance_poses(orcvio_msckf_handle* h, const orcvio_msckf_window* nui) {
    if (!h || !nui || !h->uploaded) { g_last_error = "upload_nuisance_poses: upload the window first"; return ORCVIO_ERR_INVALID; }
    if (nui->n_clones != h->n_nui) { g_last_error = "upload_nuisance_poses: count differs from ORCVIO_OPT_SCHMIDT_STATES"; return ORCVIO_ERR_INVALID; }
    if (h->n_nui == 0) return ORCVIO_OK;
    if (!nui->R_b2w || !nui->t_b_w || !nui->R_b2c || !nui->t_c_b) { g_last_error = "upload_nuisance_poses: null array"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> rec((size_t)POSE_STRIDE * h->n_nui, 0.0);
    const double* tfej = nui->t_fej ? nui->t_fej : nui->t_b_w;
    for (int i = 0; i < h->n_nui; ++i) {
        double* r = rec.data() + (size_t)POSE_STRIDE * i;
        std::memcpy(r + POSE_R_B2W, nui->R_b2w + 9 * i, 72); std::memcpy(r + POSE_T_B_W, nui->t_b_w + 3 * i, 24);
        std::memcpy(r + POSE_T_FEJ, tfej + 3 * i, 24);
        std::memcpy(r + POSE_R_B2C, nui->R_b2c + 9 * i, 72); std::memcpy(r + POSE_T_C_B, nui->t_c_b + 3 * i, 24);
    }
    HIPCHK(hipStreamSynchronize(h->stream));   // (behind the copy of the input arena)
    HIPCHK(hipMemcpy(h->d_poses + (size_t)POSE_STRIDE * h->N, rec.data(), sizeof(double) * rec.size(), hipMemcpyHostToDevice));
    return ORCVIO_OK;
}

// Dense-row scratch [cap][NAP_max] (caller-projected rows, V parts of entering features); the contents are kept.
static int dense_reserve(orcvio_msckf_handle* h, int rows) {
    if (rows <= h->dense_cap) return ORCVIO_OK;
    HIPCHK(hipDeviceSynchronize());
    const int cap = round_up(rows, 64);
    double* nb = nullptr;
    HIPCHK(hipMalloc(&nb, sizeof(double) * (size_t)cap * h->NAP_max));
    if (h->d_dense) {
        if (h->dense_rows > 0) HIPCHK(hipMemcpy(nb, h->d_dense, sizeof(double) * (size_t)h->dense_rows * h->NAP, hipMemcpyDeviceToDevice));
        (void)hipFree(h->d_dense);
    }
    h->d_dense = nb; h->dense_cap = cap;
    h->graph_epoch++;   // captured graphs hold the freed pointer
    return ORCVIO_OK;
}

// Features ENTERING the state, on the device: featureJacobian_ekf_new (src/orcvio.cpp:1481-1572) for every listed feature and
// the rotation of its rows by W = [V | U] (:2416-2436), k_ekf_new.  The V parts (zero in the new columns) are appended to the
// dense rows of this upload and take part in the update; the U parts H_1, H_2, r_1 stay on the device for
// orcvio_msckf_download_new_feature_blocks (-> orcvio_msckf_augment_state after the update).
int32_t orcvio_msckf_upload_new_features(orcvio_msckf_handle* h, const orcvio_msckf_new_features* nf) {
    if (!h || !nf || !h->uploaded) { g_last_error = "upload_new_features: upload the window first"; return ORCVIO_ERR_INVALID; }
    if (h->n_extra > 0 && !h->ekf_mode) { g_last_error = "upload_new_features: with extra states set ORCVIO_OPT_EKF_ROWS before the upload"; return ORCVIO_ERR_INVALID; }
    const int k = nf->n_features, d = nf->idp_dim, N = h->N;
    if (k < 0 || (d != 1 && d != 3)) { g_last_error = "upload_new_features: idp_dim must be 1 or 3"; return ORCVIO_ERR_INVALID; }
    if (h->new_F > 0) { g_last_error = "upload_new_features: already called for this upload (list all entering features in one call)"; return ORCVIO_ERR_INVALID; }
    if (k == 0) return ORCVIO_OK;
    if (!nf->anchor || !nf->param || !nf->p_w || !nf->obs_ptr || !nf->obs_clone || !nf->obs_z || (d == 1 && !nf->inv_depth) ||
        (h->flags.if_fej && !nf->p_fej) || (h->flags.estimate_td && !nf->obs_zvel)) { g_last_error = "upload_new_features: null array"; return ORCVIO_ERR_INVALID; }
    if (nf->obs_ptr[0] < 0) { g_last_error = "upload_new_features: obs_ptr starts below zero"; return ORCVIO_ERR_INVALID; }
    const int nobs = nf->obs_ptr[k];
    std::vector<int> row0(k + 1, 0);
    for (int j = 0; j < k; ++j) {
        const int a = nf->anchor[j];
        if (a < 0 || a >= N + h->n_nui) { g_last_error = "upload_new_features: anchor outside the window"; return ORCVIO_ERR_INVALID; }
        const int M = nf->obs_ptr[j + 1] - nf->obs_ptr[j];
        if (M < 0 || M > ORCVIO_MAX_TRACK) { g_last_error = "upload_new_features: track longer than ORCVIO_MAX_TRACK"; return ORCVIO_ERR_TRACK_TOO_LONG; }
        int kept = 0;
        for (int o = nf->obs_ptr[j]; o < nf->obs_ptr[j + 1]; ++o) {
            if (nf->obs_clone[o] < 0 || nf->obs_clone[o] >= N) { g_last_error = "upload_new_features: obs_clone outside the window"; return ORCVIO_ERR_INVALID; }
            if (!(d == 1 && nf->obs_clone[o] == a)) ++kept;   // :1494-1496
        }
        if (2 * kept <= d) { g_last_error = "upload_new_features: a feature with too few observations"; return ORCVIO_ERR_INVALID; }
        row0[j + 1] = row0[j] + 2 * kept;
    }
    HIPCHK(hipSetDevice(h->device));
    const int base = h->dense_rows, rows = row0[k];
    { const int rc = dense_reserve(h, base + rows); if (rc != ORCVIO_OK) return rc; }
    if (!h->d_Gekf) HIPCHK(hipMalloc(&h->d_Gekf, sizeof(double) * (size_t)2 * h->NAP_max * h->NAP_max));
    for (int j = 0; j <= k; ++j) row0[j] += base;
    // one staging block: doubles [param 3k | inv_depth k | p_w 3k | p_fej 3k | obs_z 2 nobs | obs_zvel 2 nobs], ints [anchor k |
    // obs_ptr k+1 | obs_clone nobs | row0 k+1]; outputs behind it on the device [H_1 d k n | H_2 k d d | r_1 d k]
    const size_t nd_in = (size_t)10 * k + (size_t)4 * nobs, ni = (size_t)3 * k + 2 + nobs, n_out = (size_t)d * k * h->n + (size_t)k * d * d + (size_t)d * k;
    const size_t bytes_in = nd_in * 8 + ((ni * 4 + 7) & ~(size_t)7);
    if (bytes_in + n_out * 8 > h->new_cap) {
        HIPCHK(hipDeviceSynchronize());
        if (h->d_new) (void)hipFree(h->d_new);
        h->new_cap = (bytes_in + n_out * 8) * 2;
        HIPCHK(hipMalloc(&h->d_new, h->new_cap));
    }
    std::vector<char> st(bytes_in, 0);
    double* sd = reinterpret_cast<double*>(st.data());
    int* si = reinterpret_cast<int*>(st.data() + nd_in * 8);
    std::memcpy(sd, nf->param, sizeof(double) * 3 * k);
    if (nf->inv_depth) std::memcpy(sd + 3 * k, nf->inv_depth, sizeof(double) * k);
    std::memcpy(sd + 4 * k, nf->p_w, sizeof(double) * 3 * k);
    if (nf->p_fej) std::memcpy(sd + 7 * k, nf->p_fej, sizeof(double) * 3 * k);
    std::memcpy(sd + 10 * k, nf->obs_z, sizeof(double) * 2 * nobs);
    if (nf->obs_zvel) std::memcpy(sd + 10 * k + 2 * nobs, nf->obs_zvel, sizeof(double) * 2 * nobs);
    std::memcpy(si, nf->anchor, sizeof(int) * k);
    std::memcpy(si + k, nf->obs_ptr, sizeof(int) * (k + 1));
    std::memcpy(si + 2 * k + 1, nf->obs_clone, sizeof(int) * nobs);
    std::memcpy(si + 2 * k + 1 + nobs, row0.data(), sizeof(int) * (k + 1));
    hipStream_t s = h->stream;
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(h->d_new, st.data(), bytes_in, hipMemcpyHostToDevice));
    const double* dd = reinterpret_cast<const double*>(h->d_new);
    const int* di = reinterpret_cast<const int*>(h->d_new + nd_in * 8);
    double* dout = reinterpret_cast<double*>(h->d_new + bytes_in);
    EkfNewArgs a;
    a.n_new = k; a.idp_dim = d; a.if_fej = h->flags.if_fej; a.estimate_td = h->flags.estimate_td; a.leg = h->flags.leg_dim;
    a.NA = h->NA; a.NAP = h->NAP; a.n = h->n; a.N = h->N; a.n_nui = h->n_nui;
    a.poses = h->d_poses;
    a.param = dd; a.inv_depth = dd + 3 * k; a.p_w = dd + 4 * k; a.p_fej = nf->p_fej ? dd + 7 * k : nullptr;
    a.obs_z = dd + 10 * k; a.obs_zvel = dd + 10 * k + 2 * nobs;
    a.anchor = di; a.obs_ptr = di + k; a.obs_clone = di + 2 * k + 1; a.row0 = di + 2 * k + 1 + nobs;
    a.dense = h->d_dense;
    a.H1 = dout; a.H2 = dout + (size_t)d * k * h->n; a.r1 = a.H2 + (size_t)k * d * d;
    hipLaunchKernelGGL(k_ekf_new, dim3(k), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    h->dense_rows = base + rows;
    h->new_F = k; h->new_idp = d; h->new_out_off = bytes_in;
    return ORCVIO_OK;
}

// H_1 [d k][n], H_2 [k][d][d] (upper triangular blocks), r_1 [d k] of the features of the last orcvio_msckf_upload_new_features
int32_t orcvio_msckf_download_new_feature_blocks(orcvio_msckf_handle* h, double* H_1, double* H_2, double* r_1) {
    if (!h || h->new_F <= 0 || !H_1 || !H_2 || !r_1) { g_last_error = "download_new_feature_blocks: no entering features uploaded"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int k = h->new_F, d = h->new_idp, n = h->n;
    const double* dout = reinterpret_cast<const double*>(h->d_new + h->new_out_off);
    HIPCHK(hipMemcpy(H_1, dout, sizeof(double) * (size_t)d * k * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(H_2, dout + (size_t)d * k * n, sizeof(double) * (size_t)k * d * d, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(r_1, dout + (size_t)d * k * n + (size_t)k * d * d, sizeof(double) * (size_t)d * k, hipMemcpyDeviceToHost));
    return ORCVIO_OK;
}

// New SLAM features, 3-parameter form, listed among the tracks of the last update: their correction and the augmented
// covariance (measurementUpdate_hybrid, src/orcvio.cpp:1811-1821 and :1904-1947, without nuisance states).  Host
// arithmetic on a few 3 x n blocks.  For a track, k_feature left T3 = Q1^T [J_msckf | r] and R = Q1^T H_f(xyz); the rows
// of featureJacobian_ekf_new differ from the MSCKF rows by H_f(xyz) X, X = d p_w / d(state) at fixed inverse-depth
// parameters (anchor pose, extrinsics), and H_f(idp) = H_f(xyz) J_pf, J_pf = R_ca2w J_f, so that
//   H_1 = T3_J + R X,   H_2 = R J_pf,   r_1 = T3_r        (:2433-2436 with U = Q1).
int32_t orcvio_msckf_augment_new_features(orcvio_msckf_handle* h, const orcvio_msckf_window* win, int32_t n_new,
                                          const int32_t* track, const int32_t* anchor, const double* inv_param,
                                          const double* dx, const double* P_upd, double* dx_new, double* P_aug) {
    if (!h || !win || !h->ran || n_new < 0 || (n_new > 0 && (!track || !anchor || !inv_param)) || !dx || !P_upd || !dx_new || !P_aug) {
        g_last_error = "augment_new_features: null argument or no finished update"; return ORCVIO_ERR_INVALID;
    }
    if (h->flags.if_fej || !h->flags.use_larvio) {
        // the equivalence "V part = MSCKF block" needs the MSCKF rows in the SLAM rows' own error-state convention
        g_last_error = "augment_new_features: needs use_larvio = 1 and if_FEJ = 0 (use orcvio_msckf_new_feature_rows / _augment_state otherwise)";
        return ORCVIO_ERR_INVALID;
    }
    const int n = h->n, NA = h->NA, NAP = h->NAP, leg = h->flags.leg_dim, N = h->N, k3 = 3 * n_new, nt = n + k3;
    for (int j = 0; j < n_new; ++j)
        if (track[j] < 0 || track[j] >= h->F || anchor[j] < 0 || anchor[j] >= N) { g_last_error = "augment_new_features: index out of range"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->last_stream ? h->last_stream : h->stream));
    std::vector<double> T3((size_t)3 * NAP), Rf(6), H1((size_t)k3 * n, 0.0), H2i((size_t)n_new * 9), HtH_i((size_t)n_new * 9), r1s(k3);
    const double s2 = h->flags.noise_feature * h->flags.noise_feature;
    auto inv3 = [](const double* A, double* B) {   // B = A^-1 (3 x 3)
        const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
        const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
        B[0] = c00 / det; B[1] = (A[2] * A[7] - A[1] * A[8]) / det; B[2] = (A[1] * A[5] - A[2] * A[4]) / det;
        B[3] = c01 / det; B[4] = (A[0] * A[8] - A[2] * A[6]) / det; B[5] = (A[2] * A[3] - A[0] * A[5]) / det;
        B[6] = c02 / det; B[7] = (A[1] * A[6] - A[0] * A[7]) / det; B[8] = (A[0] * A[4] - A[1] * A[3]) / det;
    };
    for (int j = 0; j < n_new; ++j) {
        HIPCHK(hipMemcpy(T3.data(), h->d_T3 + (size_t)3 * track[j] * NAP, sizeof(double) * 3 * NAP, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(Rf.data(), h->d_Rf + (size_t)6 * track[j], sizeof(double) * 6, hipMemcpyDeviceToHost));
        const double R[9] = {Rf[0], Rf[1], Rf[2], 0.0, Rf[3], Rf[4], 0.0, 0.0, Rf[5]};
        const int a = anchor[j];
        const double* Ra = win->R_b2w + 9 * a;      // R_ba2w
        const double* ta = win->t_b_w + 3 * a;
        const double* Rbc = win->R_b2c + 9 * a;
        const double* tcb = win->t_c_b + 3 * a;
        const double* f = inv_param + 3 * j;
        const double p_ca[3] = {f[0] / f[2], f[1] / f[2], 1.0 / f[2]};
        double q[3], pb[3], pw_rel[3];            // q = R_b2c^T p_ca ; pb = q + t_c_b ; pw_rel = R_ba2w pb = p_w - t_ba
        for (int i = 0; i < 3; ++i) q[i] = Rbc[i] * p_ca[0] + Rbc[3 + i] * p_ca[1] + Rbc[6 + i] * p_ca[2];
        for (int i = 0; i < 3; ++i) pb[i] = q[i] + tcb[i];
        for (int i = 0; i < 3; ++i) pw_rel[i] = Ra[3 * i] * pb[0] + Ra[3 * i + 1] * pb[1] + Ra[3 * i + 2] * pb[2];
        // X (3 x n): anchor clone [-skew(p_w - t_ba), I]; extrinsics [-R_ba2w skew(q), R_ba2w]
        std::vector<double> X((size_t)3 * n, 0.0);
        const double S[9] = {0, -pw_rel[2], pw_rel[1], pw_rel[2], 0, -pw_rel[0], -pw_rel[1], pw_rel[0], 0};
        const double Sq[9] = {0, -q[2], q[1], q[2], 0, -q[0], -q[1], q[0], 0};
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < 3; ++c) {
                X[(size_t)i * n + leg + 6 * a + c] = -S[3 * i + c];
                X[(size_t)i * n + leg + 6 * a + 3 + c] = (i == c) ? 1.0 : 0.0;
                double m = 0.0;
                for (int k = 0; k < 3; ++k) m += Ra[3 * i + k] * Sq[3 * k + c];
                X[(size_t)i * n + 15 + c] = -m;
                X[(size_t)i * n + 18 + c] = Ra[3 * i + c];
            }
        // H_1 = T3_J (active columns 15 .. 15+NA) + R X
        for (int i = 0; i < 3; ++i) {
            double* row = &H1[(size_t)(3 * j + i) * n];
            for (int c = 0; c < NA; ++c) row[15 + c] = T3[(size_t)i * NAP + c];
            for (int c = 0; c < n; ++c) {
                double m = 0.0;
                for (int k = i; k < 3; ++k) m += R[3 * i + k] * X[(size_t)k * n + c];
                row[c] += m;
            }
            r1s[3 * j + i] = T3[(size_t)i * NAP + NA];
        }
        // H_2 = R R_ca2w J_f,  R_ca2w = R_ba2w R_b2c^T,  J_f = d p_ca / d(alpha, beta, rho)
        double Rca[9], Jf[9] = {1.0 / f[2], 0.0, -f[0] / (f[2] * f[2]), 0.0, 1.0 / f[2], -f[1] / (f[2] * f[2]), 0.0, 0.0, -1.0 / (f[2] * f[2])};
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < 3; ++c) Rca[3 * i + c] = Ra[3 * i] * Rbc[3 * c] + Ra[3 * i + 1] * Rbc[3 * c + 1] + Ra[3 * i + 2] * Rbc[3 * c + 2];
        double M1[9], H2[9];
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < 3; ++c) M1[3 * i + c] = Rca[3 * i] * Jf[c] + Rca[3 * i + 1] * Jf[3 + c] + Rca[3 * i + 2] * Jf[6 + c];
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < 3; ++c) H2[3 * i + c] = R[3 * i] * M1[c] + R[3 * i + 1] * M1[3 + c] + R[3 * i + 2] * M1[6 + c];
        if (h->ref_h2_ldlt) {
            // The reference's literal tail: its U is the Q factor of H_f in the INVERSE-DEPTH parametrisation (SPQR, :2421-2428), so
            // its H_2 is that matrix's upper-triangular R, and H_2.ldlt() (:1826-1827) reads the lower triangle of it -- the diagonal.
            // Here U = Q_1 of H_f(xyz) (same column space): H_2 = Q_3 R_idp by a 3 x 3 Gram-Schmidt QR, the three rows of H_1 and r_1
            // are rotated by Q_3^T into the reference's basis (up to the signs of the rows, which the diagonal division cancels).
            double Q3[9], Rr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int c = 0; c < 3; ++c) {
                double v[3] = {H2[c], H2[3 + c], H2[6 + c]};
                for (int q = 0; q < c; ++q) {
                    const double dq = Q3[q] * v[0] + Q3[3 + q] * v[1] + Q3[6 + q] * v[2];
                    Rr[3 * q + c] = dq;
                    for (int i = 0; i < 3; ++i) v[i] -= dq * Q3[3 * i + q];
                }
                const double nv = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                Rr[3 * c + c] = nv;
                for (int i = 0; i < 3; ++i) Q3[3 * i + c] = v[i] / nv;
            }
            double rot[3];
            for (int c = 0; c < n; ++c) {
                for (int i = 0; i < 3; ++i) rot[i] = Q3[i] * H1[(size_t)(3 * j) * n + c] + Q3[3 + i] * H1[(size_t)(3 * j + 1) * n + c] + Q3[6 + i] * H1[(size_t)(3 * j + 2) * n + c];
                for (int i = 0; i < 3; ++i) H1[(size_t)(3 * j + i) * n + c] = rot[i];
            }
            for (int i = 0; i < 3; ++i) rot[i] = Q3[i] * r1s[3 * j] + Q3[3 + i] * r1s[3 * j + 1] + Q3[6 + i] * r1s[3 * j + 2];
            for (int i = 0; i < 3; ++i) r1s[3 * j + i] = rot[i];
            std::memcpy(H2, Rr, sizeof(Rr));
            for (int i = 0; i < 9; ++i) H2i[9 * j + i] = 0.0;
            for (int i = 0; i < 3; ++i) H2i[9 * j + 4 * i] = 1.0 / H2[4 * i];
        } else inv3(H2, &H2i[9 * j]);
        double HtH[9];
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < 3; ++c) HtH[3 * i + c] = H2[i] * H2[c] + H2[3 + i] * H2[3 + c] + H2[6 + i] * H2[6 + c];
        inv3(HtH, &HtH_i[9 * j]);
    }
    // HH = H_2^-1 H_1 (block rows), dx_new = -HH dx + H_2^-1 r_1, nHHP = -HH P, P22 = HH P HH^T + s2 (H_2^T H_2)^-1
    std::vector<double> HH((size_t)k3 * n, 0.0), nHHP((size_t)k3 * n, 0.0);
    for (int j = 0; j < n_new; ++j)
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < n; ++c) {
                double m = 0.0;
                for (int k = 0; k < 3; ++k) m += H2i[9 * j + 3 * i + k] * H1[(size_t)(3 * j + k) * n + c];
                HH[(size_t)(3 * j + i) * n + c] = m;
            }
    for (int j = 0; j < n_new; ++j)
        for (int i = 0; i < 3; ++i) {
            double m = 0.0;
            for (int k = 0; k < 3; ++k) m += H2i[9 * j + 3 * i + k] * r1s[3 * j + k];
            for (int c = 0; c < n; ++c) m -= HH[(size_t)(3 * j + i) * n + c] * dx[c];
            dx_new[3 * j + i] = m;
        }
    for (int r = 0; r < k3; ++r)
        for (int k = 0; k < n; ++k) {
            const double hv = HH[(size_t)r * n + k];
            if (hv == 0.0) continue;
            const double* prow = P_upd + (size_t)k * n;
            double* o = &nHHP[(size_t)r * n];
            for (int c = 0; c < n; ++c) o[c] -= hv * prow[c];
        }
    for (int r = 0; r < n; ++r) std::memcpy(P_aug + (size_t)r * nt, P_upd + (size_t)r * n, sizeof(double) * n);
    for (int r = 0; r < k3; ++r)
        for (int c = 0; c < n; ++c) { P_aug[(size_t)(n + r) * nt + c] = nHHP[(size_t)r * n + c]; P_aug[(size_t)c * nt + n + r] = nHHP[(size_t)r * n + c]; }
    for (int r = 0; r < k3; ++r)
        for (int c = 0; c < k3; ++c) {
            double m = 0.0;
            for (int k = 0; k < n; ++k) m -= nHHP[(size_t)r * n + k] * HH[(size_t)c * n + k];
            if (r / 3 == c / 3) m += s2 * HtH_i[9 * (r / 3) + 3 * (r % 3) + (c % 3)];
            P_aug[(size_t)(n + r) * nt + n + c] = m;
        }
    for (int r = 0; r < nt; ++r)   // (:1946) symmetrise
        for (int c = r + 1; c < nt; ++c) { const double m = 0.5 * (P_aug[(size_t)r * nt + c] + P_aug[(size_t)c * nt + r]); P_aug[(size_t)r * nt + c] = m; P_aug[(size_t)c * nt + r] = m; }
    return ORCVIO_OK;
}

// ---- host arithmetic for features entering the state, either parametrisation (no device, no handle) -----------------
// Rows of featureJacobian_ekf_new (src/orcvio.cpp:1481-1572) for every listed feature and their rotation by W = [V | U]
// (:2416-2436): H_f of different features share no column, so the split is one small Householder QR per feature
// (U = the first d columns of Q, V the rest; only the two subspaces matter).
//   out: H_top [*rows_top][n_cols], r_top  -- the V parts, zero in the new columns: to orcvio_msckf_upload_dense_rows
//        H_1 [d n_new][n_cols], H_2 [n_new][d][d] (upper triangular blocks), r_1 [d n_new]  -- the U parts
int32_t orcvio_msckf_new_feature_rows(const orcvio_msckf_flags* flags, const orcvio_msckf_window* win, int32_t idp_dim, int32_t n_cols,
                                      int32_t n_new, const int32_t* anchor, const double* param, const double* inv_depth,
                                      const double* p_w, const double* p_fej, const int32_t* obs_ptr, const int32_t* obs_clone,
                                      const double* obs_z, const double* obs_zvel, int32_t* rows_top, double* H_top, double* r_top,
                                      double* H_1, double* H_2, double* r_1) {
    if (!flags || !win || n_new < 0 || (idp_dim != 1 && idp_dim != 3) || !rows_top || (n_new > 0 && (!anchor || !param || !p_w || !obs_ptr ||
        !obs_clone || !obs_z || !H_top || !r_top || !H_1 || !H_2 || !r_1 || (idp_dim == 1 && !inv_depth) || (flags->if_fej && !p_fej) ||
        (flags->estimate_td && !obs_zvel)))) { g_last_error = "new_feature_rows: null argument"; return ORCVIO_ERR_INVALID; }
    const int d = idp_dim, N = win->n_clones, leg = flags->leg_dim;
    if (n_cols < leg + 6 * N) { g_last_error = "new_feature_rows: n_cols smaller than the window"; return ORCVIO_ERR_INVALID; }
    auto pose = [&](int i, double* rec) {
        std::memcpy(rec + POSE_R_B2W, win->R_b2w + 9 * i, 72); std::memcpy(rec + POSE_T_B_W, win->t_b_w + 3 * i, 24);
        std::memcpy(rec + POSE_T_FEJ, (win->t_fej ? win->t_fej : win->t_b_w) + 3 * i, 24);
        std::memcpy(rec + POSE_R_B2C, win->R_b2c + 9 * i, 72); std::memcpy(rec + POSE_T_C_B, win->t_c_b + 3 * i, 24);
    };
    int top = 0;
    for (int j = 0; j < n_new; ++j) {
        const int a = anchor[j];
        if (a < 0 || a >= N) { g_last_error = "new_feature_rows: anchor outside the window"; return ORCVIO_ERR_INVALID; }
        std::vector<int> obs;
        for (int o = obs_ptr[j]; o < obs_ptr[j + 1]; ++o) {
            if (obs_clone[o] < 0 || obs_clone[o] >= N) { g_last_error = "new_feature_rows: obs_clone outside the window"; return ORCVIO_ERR_INVALID; }
            if (d == 1 && obs_clone[o] == a) continue;   // :1494-1496
            obs.push_back(o);
        }
        const int m = 2 * (int)obs.size();
        if (m <= d) { g_last_error = "new_feature_rows: a feature with too few observations"; return ORCVIO_ERR_INVALID; }
        std::vector<double> Hx((size_t)m * n_cols, 0.0), Hf((size_t)m * d, 0.0), r(m, 0.0);
        double Pa[POSE_STRIDE], Pk[POSE_STRIDE];
        pose(a, Pa);
        for (size_t c = 0; c < obs.size(); ++c) {
            const int o = obs[c], k = obs_clone[o];
            pose(k, Pk);
            double He[12], Ha[12], Hxk[12], Hfk[6], rr[2];
            ekf_row_blocks(Pk, Pa, k == a, d, flags->if_fej, param + 3 * j, d == 1 ? inv_depth[j] : 0.0, p_w + 3 * j,
                           p_fej ? p_fej + 3 * j : nullptr, obs_z + 2 * o, He, Ha, Hxk, Hfk, rr);
            for (int b = 0; b < 2; ++b) {
                double* row = &Hx[(size_t)(2 * c + b) * n_cols];
                for (int e = 0; e < 6; ++e) row[leg + 6 * a + e] = Ha[6 * b + e];            // :1561
                for (int e = 0; e < 6; ++e) row[leg + 6 * k + e] = Hxk[6 * b + e];           // :1562 (overwrites if k == a)
                for (int e = 0; e < 6; ++e) row[15 + e] = He[6 * b + e];                     // :1563
                if (flags->estimate_td) row[21] = obs_zvel[2 * o + b];                       // :1564-1565
                for (int e = 0; e < d; ++e) Hf[(size_t)(2 * c + b) * d + e] = Hfk[b * d + e];
                r[2 * c + b] = rr[b];
            }
        }
        // Householder QR of H_f (m x d), applied to [H_x | r]
        for (int q = 0; q < d; ++q) {
            double nrm2 = 0.0;
            for (int i = q + 1; i < m; ++i) nrm2 += Hf[(size_t)i * d + q] * Hf[(size_t)i * d + q];
            const double alpha = Hf[(size_t)q * d + q];
            if (nrm2 == 0.0) continue;
            const double nu = std::sqrt(alpha * alpha + nrm2), bk = alpha >= 0.0 ? -nu : nu;
            const double beta = (bk - alpha) / bk, sc = 1.0 / (alpha - bk);
            std::vector<double> v(m, 0.0);
            v[q] = 1.0;
            for (int i = q + 1; i < m; ++i) v[i] = Hf[(size_t)i * d + q] * sc;
            auto apply = [&](double* M, int ld, int c0, int c1) {
                for (int c = c0; c < c1; ++c) {
                    double w = 0.0;
                    for (int i = q; i < m; ++i) w += v[i] * M[(size_t)i * ld + c];
                    w *= beta;
                    for (int i = q; i < m; ++i) M[(size_t)i * ld + c] -= w * v[i];
                }
            };
            apply(Hf.data(), d, q, d);
            apply(Hx.data(), n_cols, 0, n_cols);
            apply(r.data(), 1, 0, 1);
        }
        for (int i = 0; i < d; ++i) {   // U part
            std::memcpy(H_1 + (size_t)(d * j + i) * n_cols, &Hx[(size_t)i * n_cols], sizeof(double) * n_cols);
            for (int e = 0; e < d; ++e) H_2[(size_t)j * d * d + i * d + e] = e >= i ? Hf[(size_t)i * d + e] : 0.0;
            r_1[d * j + i] = r[i];
        }
        for (int i = d; i < m; ++i) {   // V part
            std::memcpy(H_top + (size_t)top * n_cols, &Hx[(size_t)i * n_cols], sizeof(double) * n_cols);
            r_top[top++] = r[i];
        }
    }
    *rows_top = top;
    return ORCVIO_OK;
}

// measurementUpdate_hybrid, the part behind the update of the legacy state (src/orcvio.cpp:1818-1821, :1904-1947, no
// nuisance states): dx_new = H_2^-1 (r_1 - H_1 dx), P_aug = [[P, (-HH P)^T], [-HH P, HH P HH^T + s2 (H_2^T H_2)^-1]],
// HH = H_2^-1 H_1, with H_2 block diagonal (one upper-triangular d x d block per feature).
// (The reference writes H_2.ldlt().solve(.), which for d = 3 reads only the lower triangle of an upper-triangular H_2;
// the triangular system is solved here.  For the 1-parameter form of the shipped configurations H_2 is diagonal and the
// two coincide.)
// diag_only: the reference's literal arithmetic for HH and dx_new -- `H_2.ldlt().solve(..)` (src/orcvio.cpp:1826-1827) on an
// UPPER-triangular H_2: Eigen's LDLT reads the lower triangle only, i.e. diag(H_2) (the same thing for the 1 x 1 blocks of
// feature_idp_dim = 1, every shipped configuration).  P22 uses (H_2^T H_2)^-1 in both modes, as the reference does (:1907-1908).
static int augment_state_impl(int32_t n, int32_t n_new, int32_t idp_dim, const double* H_1, const double* H_2, const double* r_1,
                              double sigma2, const double* dx, const double* P_upd, bool diag_only, double* dx_new, double* P_aug) {
    if (n < 1 || n_new < 0 || (idp_dim != 1 && idp_dim != 3) || !dx || !P_upd || !dx_new || !P_aug || (n_new > 0 && (!H_1 || !H_2 || !r_1))) {
        g_last_error = "augment_state: invalid argument"; return ORCVIO_ERR_INVALID;
    }
    const int d = idp_dim, sz = d * n_new, nt = n + sz;
    std::vector<double> HH((size_t)sz * n, 0.0), nHHP((size_t)sz * n, 0.0), W((size_t)n_new * d * d, 0.0);
    for (int j = 0; j < n_new; ++j) {
        const double* R = H_2 + (size_t)j * d * d;
        for (int i = 0; i < d; ++i)
            if (R[i * d + i] == 0.0) { g_last_error = "augment_state: singular H_2"; return ORCVIO_ERR_NOT_SPD; }
        // back substitution on the block: HH_j = R^-1 H_1_j, x_j = R^-1 r_1_j
        for (int c = 0; c <= n; ++c)
            for (int i = d - 1; i >= 0; --i) {
                double m = c < n ? H_1[(size_t)(d * j + i) * n + c] : r_1[d * j + i];
                if (!diag_only)
                    for (int k = i + 1; k < d; ++k) m -= R[i * d + k] * (c < n ? HH[(size_t)(d * j + k) * n + c] : dx_new[d * j + k]);
                m /= R[i * d + i];
                if (c < n) HH[(size_t)(d * j + i) * n + c] = m; else dx_new[d * j + i] = m;
            }
        // (R^T R)^-1 = R^-1 R^-T
        double Ri[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < d; ++c)
            for (int i = d - 1; i >= 0; --i) {
                double m = i == c ? 1.0 : 0.0;
                for (int k = i + 1; k < d; ++k) m -= R[i * d + k] * Ri[k * d + c];
                Ri[i * d + c] = m / R[i * d + i];
            }
        for (int i = 0; i < d; ++i)
            for (int c = 0; c < d; ++c) {
                double m = 0.0;
                for (int k = 0; k < d; ++k) m += Ri[i * d + k] * Ri[c * d + k];
                W[(size_t)j * d * d + i * d + c] = m;
            }
    }
    for (int r = 0; r < sz; ++r) {
        double m = 0.0;
        for (int c = 0; c < n; ++c) m += HH[(size_t)r * n + c] * dx[c];
        dx_new[r] -= m;
    }
    for (int r = 0; r < sz; ++r)
        for (int k = 0; k < n; ++k) {
            const double hv = HH[(size_t)r * n + k];
            if (hv == 0.0) continue;
            const double* prow = P_upd + (size_t)k * n;
            double* o = &nHHP[(size_t)r * n];
            for (int c = 0; c < n; ++c) o[c] -= hv * prow[c];
        }
    for (int r = 0; r < n; ++r) std::memcpy(P_aug + (size_t)r * nt, P_upd + (size_t)r * n, sizeof(double) * n);
    for (int r = 0; r < sz; ++r)
        for (int c = 0; c < n; ++c) { P_aug[(size_t)(n + r) * nt + c] = nHHP[(size_t)r * n + c]; P_aug[(size_t)c * nt + n + r] = nHHP[(size_t)r * n + c]; }
    for (int r = 0; r < sz; ++r)
        for (int c = 0; c < sz; ++c) {
            double m = 0.0;
            for (int k = 0; k < n; ++k) m -= nHHP[(size_t)r * n + k] * HH[(size_t)c * n + k];
            if (r / d == c / d) m += sigma2 * W[(size_t)(r / d) * d * d + (r % d) * d + (c % d)];
            P_aug[(size_t)(n + r) * nt + n + c] = m;
        }
    for (int r = 0; r < nt; ++r)   // (:1946)
        for (int c = r + 1; c < nt; ++c) { const double m = 0.5 * (P_aug[(size_t)r * nt + c] + P_aug[(size_t)c * nt + r]); P_aug[(size_t)r * nt + c] = m; P_aug[(size_t)c * nt + r] = m; }
    return ORCVIO_OK;
}

int32_t orcvio_msckf_augment_state(int32_t n, int32_t n_new, int32_t idp_dim, const double* H_1, const double* H_2, const double* r_1,
                                   double sigma2, const double* dx, const double* P_upd, double* dx_new, double* P_aug) {
    return augment_state_impl(n, n_new, idp_dim, H_1, H_2, r_1, sigma2, dx, P_upd, false, dx_new, P_aug);
}

// ... with Schmidt nuisance states: the new feature states go IN FRONT of the trailing nui_rows nuisance rows / columns
// (src/orcvio.cpp:1920-1935).  n counts the nuisance states; P_aug [(n + d n_new)^2] in the order [old | new | nuisance].
static int augment_state_nuisance_impl(int32_t n, int32_t n_new, int32_t idp_dim, int32_t nui_rows, const double* H_1, const double* H_2,
                                       const double* r_1, double sigma2, const double* dx, const double* P_upd, bool diag_only, double* dx_new,
                                       double* P_aug) {
    if (nui_rows < 0 || nui_rows > n) { g_last_error = "augment_state_nuisance: invalid argument"; return ORCVIO_ERR_INVALID; }
    const int sz = idp_dim * n_new, nt = n + sz;
    std::vector<double> T((size_t)nt * nt);
    const int rc = augment_state_impl(n, n_new, idp_dim, H_1, H_2, r_1, sigma2, dx, P_upd, diag_only, dx_new, T.data());   // [old + nuisance | new]
    if (rc != ORCVIO_OK) return rc;
    std::vector<int> map(nt);   // map[i] = index in T of row / column i of P_aug
    const int n0 = n - nui_rows;
    for (int i = 0; i < n0; ++i) map[i] = i;
    for (int i = 0; i < sz; ++i) map[n0 + i] = n + i;
    for (int i = 0; i < nui_rows; ++i) map[n0 + sz + i] = n0 + i;
    for (int i = 0; i < nt; ++i)
        for (int j = 0; j < nt; ++j) P_aug[(size_t)i * nt + j] = T[(size_t)map[i] * nt + map[j]];
    return ORCVIO_OK;
}

int32_t orcvio_msckf_augment_state_nuisance(int32_t n, int32_t n_new, int32_t idp_dim, int32_t nui_rows, const double* H_1, const double* H_2,
                                            const double* r_1, double sigma2, const double* dx, const double* P_upd, double* dx_new,
                                            double* P_aug) {
    return augment_state_nuisance_impl(n, n_new, idp_dim, nui_rows, H_1, H_2, r_1, sigma2, dx, P_upd, false, dx_new, P_aug);
}
int32_t orcvio_msckf_augment_state_ref_ldlt(int32_t n, int32_t n_new, int32_t idp_dim, int32_t nui_rows, const double* H_1, const double* H_2,
                                            const double* r_1, double sigma2, const double* dx, const double* P_upd, double* dx_new,
                                            double* P_aug) {
    return augment_state_nuisance_impl(n, n_new, idp_dim, nui_rows, H_1, H_2, r_1, sigma2, dx, P_upd, true, dx_new, P_aug);
}

int32_t orcvio_msckf_download_ekf(orcvio_msckf_handle* h, double* gamma, int32_t* accept) {
    if (!h || !h->ran) { g_last_error = "download_ekf: no finished update"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->last_stream ? h->last_stream : h->stream));
    if (h->ekf_F > 0) {
        if (gamma) HIPCHK(hipMemcpy(gamma, h->d_ekf_gamma, sizeof(double) * h->ekf_F, hipMemcpyDeviceToHost));
        if (accept) HIPCHK(hipMemcpy(accept, h->d_ekf_accept, sizeof(int) * h->ekf_F, hipMemcpyDeviceToHost));
    }
    return ORCVIO_OK;
}

int32_t orcvio_msckf_run_update(orcvio_msckf_handle* h, void* stream) {
    if (!h || !h->uploaded) { g_last_error = "run_update: nothing uploaded"; return ORCVIO_ERR_INVALID; }
    if (h->pw_missing) { g_last_error = "run_update: tracks were uploaded without positions and have not been triangulated"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = pick_stream(h, stream);
    h->last_stream = s;
    if (h->dl_pending) { HIPCHK(hipStreamSynchronize(h->dl_stream)); h->dl_pending = false; }   // (a copy of the previous results nobody fetched)
    int rc = run_with_graph(h, h->g_update, launch_signature(h, s, nullptr, 0), s, [&](bool) { return enqueue_update(h, s); });
    h->A_deferred = front_defers_assembly(h);   // (a replayed graph does not pass through enqueue_update)
    if (rc == ORCVIO_OK) { h->ran = true; h->last_update_objects = false; h->last_run_kind = 0; h->last_sharded = false; }
    return rc;
}

int32_t orcvio_msckf_sync(orcvio_msckf_handle* h, void* stream) {
    if (!h) return ORCVIO_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    if (h->comm) {   // the stream may carry a collective: a bounded wait (ORCVIO_ERR_TIMEOUT instead of a hang)
        const int rw = comm_stream_wait(h, pick_stream(h, stream), "orcvio_msckf_sync");
        if (rw != ORCVIO_OK) return rw;
    } else
        HIPCHK(hipStreamSynchronize(pick_stream(h, stream)));
    HIPCHK(hipStreamSynchronize(h->side));
    return ORCVIO_OK;
}

// ---- download --------------------------------------------------------------------------------
// Optional outputs in the reference's own terms (computed on the device, on request only):
//   H_thin = R_A (A = R_A^T R_A, zero rows on rank-deficient directions), r_thin = R_A^-T b,
//   K = P H_thin^T S^-1 = Lf M^-1 L_a^T R_A^T,   G = K H_thin = Lf M^-1 L_a^T A.
static int compute_optional(orcvio_msckf_handle* h, bool want_thin_or_K, bool want_K, bool want_G) {
    // (kf: the columns of the prior's factor that take part in M' -- the trailing h->tail ones are zero in the active rows, so
    //  U, W and the products below have nothing there)
    const int NA = h->NA, NAP = h->NAP, n = h->n, kf = h->kf - h->tail, NP = h->NP, ldz = h->ldz;
    hipStream_t s = h->stream;
    const PriorFactor pf = prior_factor(h);
    const long sLi = pf.sLi, sLj = pf.sLj;
    const double* La_P = pf.base + 15 * sLi;
    { const int ra = assemble_deferred(h, s); if (ra != ORCVIO_OK) return ra; }
    if (want_thin_or_K) {   // lower Cholesky factor of the Gram block with the LDS-panel kernel
        HIPCHK(hipMemcpyAsync(h->d_La, h->d_A, sizeof(double) * (size_t)NAP * NAP, hipMemcpyDeviceToDevice, s));
        // the Gram block is singular in every update (gauge freedom): factor A + 1e-11 max(diag) I, which needs no
        // rank decision and changes H_thin^T H_thin by 1e-11 relative
        hipLaunchKernelGGL(k_potrf, dim3(1), dim3(1024), 0, s, h->d_La, NA + 1, NAP, 0.0, h->d_DinvA, h->d_info + 4, 1e-11);
        HIPCHK(hipGetLastError());
    }
    if (want_G) {   // W = L_M^-1 U[0:NA]^T (n x NA);  G[:, 15:] = Zn^T W
        int rc = launch_trsm(h, s, h->d_RM, h->d_DinvM, kf, h->d_U, 1, NP, NA, nullptr, 0, h->d_W, NP);
        if (rc == ORCVIO_OK) rc = launch_gemm(s, h->d_Z, 1, ldz, h->d_W, NP, 1, n, NA, kf, 1.0, 0.0, 0, h->d_KG, NP, 1);
        if (rc != ORCVIO_OK) return rc;
    }
    if (want_K) {   // Y = L_a^T R_A^T (n x NA); W = L_M^-1 Y; K = Zn^T W  -> stored after G in d_KG
        int rc = launch_gemm(s, La_P, sLj, sLi, h->d_La, NAP, 1, kf, NA, NA, 1.0, 0.0, 0, h->d_Y, NP, 1);
        if (rc == ORCVIO_OK) rc = launch_trsm(h, s, h->d_RM, h->d_DinvM, kf, h->d_Y, NP, 1, NA, nullptr, 0, h->d_W, NP);
        if (rc == ORCVIO_OK) rc = launch_gemm(s, h->d_Z, 1, ldz, h->d_W, NP, 1, n, NA, kf, 1.0, 0.0, 0, h->d_Y, NP, 1);
        if (rc != ORCVIO_OK) return rc;
    }
    HIPCHK(hipStreamSynchronize(s));
    return ORCVIO_OK;
}

// Device -> host copy of the outputs arena ([info | dx | gamma | accept], with P+ behind it if wanted) into the pinned
// mirror, enqueued on `sd` behind the update: ONE copy.  download() waits for it (or issues it itself).
static int download_enqueue(orcvio_msckf_handle* h, hipStream_t sd, bool with_P) {
    const size_t bytes = with_P ? h->oo_Pout + sizeof(double) * (size_t)h->n * h->n : h->outs_small;
    HIPCHK(hipMemcpyAsync(h->h_stage + h->in_cap, h->d_outs, bytes, hipMemcpyDeviceToHost, sd));
    h->dl_pending = true; h->dl_with_P = with_P; h->dl_stream = sd;
    return ORCVIO_OK;
}

int32_t orcvio_msckf_download(orcvio_msckf_handle* h, orcvio_msckf_result* res) {
    if (!h || !res || !h->ran) { g_last_error = "download: no finished update"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    const int n = h->n, NA = h->NA, NAP = h->NAP, NP = h->NP, F = h->F;
    const bool want_P = res->P_out != nullptr;
    if (!(h->dl_pending && (h->dl_with_P || !want_P))) {   // not enqueued behind the update (staged callers), or without P+
        hipStream_t sd = h->last_stream ? h->last_stream : h->stream;
        if (h->dl_pending) HIPCHK(hipStreamSynchronize(h->dl_stream));
        HIPCHK(hipStreamSynchronize(h->side));
        const int rq = download_enqueue(h, sd, want_P);
        if (rq != ORCVIO_OK) return rq;
    }
    HIPCHK(hipStreamSynchronize(h->dl_stream));
    h->dl_pending = false;
    const char* so = h->h_stage + h->in_cap;
    const double* dx = reinterpret_cast<const double*>(so + h->oo_dx);
    const double* gam = reinterpret_cast<const double*>(so + h->oo_gamma);
    const int* acc = reinterpret_cast<const int*>(so + h->oo_accept);
    if (res->dx) std::memcpy(res->dx, dx, sizeof(double) * n);
    if (res->P_out) std::memcpy(res->P_out, so + h->oo_Pout, sizeof(double) * (size_t)n * n);
    if (res->accept && F > 0) std::memcpy(res->accept, acc, sizeof(int) * F);
    if (res->gamma && F > 0) std::memcpy(res->gamma, gam, sizeof(double) * F);
    int info[16] = {0};
    std::memcpy(info, so, sizeof(int) * 16);
    if (info[8] != 0) {   // a wait inside a launch gave up: a workgroup of k_front at its device-wide counter (somebody else's kernel
                          // held CUs it needed), or a solver wavefront of k_potrf_solve
        HIPCHK(hipMemset(h->d_info + 8, 0, sizeof(int)));
        HIPCHK(hipMemset(h->d_sync, 0, 256));
        if (h->last_run_kind == 0 && !h->front_retry_forked) {
            // single-GPU update: run it again, on the forked path (plain launches, no in-launch device-wide wait), inside this call
            h->front_retry_forked = true;
            h->front_fallbacks++;
            int rr = orcvio_msckf_run_update(h, h->last_stream);
            if (rr == ORCVIO_OK) rr = download_enqueue(h, h->last_stream ? h->last_stream : h->stream, want_P);
            h->front_retry_forked = false;
            if (rr != ORCVIO_OK) return rr;
            return orcvio_msckf_download(h, res);   // (a second time-out in the forked form is reported as an error below)
        }
        // the solve ran through on stale bytes: nothing of this update may be committed
        h->ran = false;
        g_last_error = "k_potrf_solve / k_front: an in-launch hand-off timed out";
        return ORCVIO_ERR_TIMEOUT;
    }
    if (h->last_sharded && info[9] != 0) {   // sharded update: another rank took part with an empty share and a status word
        h->ran = false;                            // (no rank commits: every rank returns an error)
        g_last_error = "sharded update: rank " + std::to_string(info[9] - 1) + " could not take part with its tracks (status " + std::to_string(info[10]) + ")";
        return ORCVIO_ERR_PEER;
    }
    {
        const int ro = feature_outcome(h, so, res->stats);
        if (ro != ORCVIO_OK) return ro;
    }
    if (h->last_sharded) {   // the joint update is applied whenever ANY rank stacked rows (the gathered status words)
        res->stats[1] = info[12] > 0 ? NA : 0;
        res->stats[3] = info[12] > 0 ? 1 : 0;
    }
    const bool want_thin = res->H_thin || res->r_thin;
    if (want_thin || res->K || res->G) {
        int rc = compute_optional(h, want_thin || res->K, res->K != nullptr, res->G != nullptr);
        if (rc != ORCVIO_OK) return rc;
        if (want_thin) {
            std::vector<double> La((size_t)NAP * NAP);
            HIPCHK(hipMemcpy(La.data(), h->d_La, sizeof(double) * La.size(), hipMemcpyDeviceToHost));
            if (res->H_thin) {
                std::memset(res->H_thin, 0, sizeof(double) * (size_t)NA * n);
                for (int i = 0; i < NA; ++i)
                    for (int k = i; k < NA; ++k) res->H_thin[(size_t)i * n + 15 + k] = La[(size_t)k * NAP + i];
            }
            if (res->r_thin)
                for (int i = 0; i < NA; ++i) res->r_thin[i] = La[(size_t)NA * NAP + i];
        }
        if (res->G) {
            std::vector<double> Gd((size_t)n * NP);
            HIPCHK(hipMemcpy(Gd.data(), h->d_KG, sizeof(double) * Gd.size(), hipMemcpyDeviceToHost));
            std::memset(res->G, 0, sizeof(double) * (size_t)n * n);
            for (int i = 0; i < n; ++i)
                for (int c = 0; c < NA; ++c) res->G[(size_t)i * n + 15 + c] = Gd[(size_t)i * NP + c];
        }
        if (res->K) {
            std::vector<double> Kd((size_t)n * NP);
            HIPCHK(hipMemcpy(Kd.data(), h->d_Y, sizeof(double) * Kd.size(), hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i)
                for (int c = 0; c < NA; ++c) res->K[(size_t)i * NA + c] = Kd[(size_t)i * NP + c];
        }
    }
    return ORCVIO_OK;
}

// ---- the zero-copy update (orcvio_msckf_io_begin / _io_update; the copying one-shot calls run on it too) ----------------------
// ONE graph launch per update: k_ingest (pinned arena -> HBM) -> the update's kernels -> k_epilogue (results -> host-coherent
// memory, [the commit of P+ and its square-root factor], then the flag).  No copy-engine transfer, no stream
// synchronisation: the calling thread spins on the flag.
static unsigned long long io_signature(const orcvio_msckf_handle* h, hipStream_t s, bool want_P, bool commit) {
    unsigned long long sig = launch_signature(h, s, h->h_stage_dev, (long)(0x100 | (want_P ? 1 : 0) | (commit ? 2 : 0)));
    auto mix = [&](unsigned long long v) { sig = (sig ^ v) * 1099511628211ull; };
    mix((unsigned long long)upload_bytes(h)); mix((unsigned long long)(size_t)h->d_Stmp); mix((unsigned long long)(size_t)h->d_Pres);
    mix(h->factor_opt); mix((unsigned long long)h->outs_small);
    return sig;
}

static int io_enqueue(orcvio_msckf_handle* h, hipStream_t s, bool want_P, bool commit) {
    const int n = h->n;
    // inputs: one pass over the arena, 16 bytes per lane, enough workgroups for the block to be one or two iterations
    int rc = launch_ingest(h, s, h->h_stage_dev, h->d_in, upload_bytes(h));
    if (rc == ORCVIO_OK) rc = enqueue_update(h, s);
    if (rc != ORCVIO_OK) return rc;
    // ONE launch behind the update: the results to host-coherent memory, the commit (refused on the device if the update is),
    // and the flag the caller waits on (k_epilogue)
    EpilogueArgs ea{};
    ea.small_src = reinterpret_cast<const u32x4*>(h->d_outs);
    ea.small_dst = reinterpret_cast<u32x4*>(h->h_stage_dev + h->in_cap);
    ea.small16 = h->outs_small / 16;
    ea.P_src = reinterpret_cast<const u32x4*>(h->d_outs + h->oo_Pout);
    ea.P_dst = reinterpret_cast<u32x4*>(h->h_stage_dev + h->in_cap + h->oo_Pout);
    ea.P16 = want_P ? (sizeof(double) * (size_t)n * n + 15) / 16 : 0;
    ea.nb_P = want_P ? 40 : 0;
    const bool fac = commit && h->factor_opt && h->n_nui == 0;
    ea.commit = commit ? (fac ? 2 : 1) : 0;
    ea.Pout = h->d_Pout; ea.Pres = h->d_Pres; ea.nn = (size_t)n * n;
    const PriorFactor pf = prior_factor(h);
    ea.Z = h->d_Z; ea.ldz = h->ldz; ea.kf = h->kf; ea.n = n; ea.sigma = h->flags.noise_feature;
    ea.prior = pf.base; ea.sLi = pf.sLi; ea.sLj = pf.sLj; ea.Sout = h->d_Stmp; ea.ldo = h->ldz;
    ea.dx = h->d_dx; ea.info = h->d_info;
    ea.counter = h->d_pubcnt; ea.seq = h->d_seq; ea.flag = h->h_flag_dev;
    hipLaunchKernelGGL(k_epilogue, dim3(1 + ea.nb_P + (commit ? 40 : 0)), dim3(256), 0, s, ea);
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

// Wait for the next publication: spin on the host-coherent flag (bounded), then fall back to a stream synchronisation.
static int io_wait(orcvio_msckf_handle* h, hipStream_t s) {
    const unsigned long long expected = h->pub_enqueued;   // (the latest publication on the stream: earlier ones nobody waited for are covered)
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    bool late = false;
    while (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) < expected) {
        _mm_pause();
        if ((++spins & 4095u) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > h->io_spin_seconds) { late = true; break; }
    }
    if (late) {
        HIPCHK(hipStreamSynchronize(s));   // (the in-launch waits of the kernels are bounded: the stream drains)
        const unsigned long long now = __atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE);
        if (now < expected) {
            h->pub_enqueued = now;   // (a launch that never ran does not publish later either)
            g_last_error = "io_update: the results were not published";
            return ORCVIO_ERR_TIMEOUT;
        }
    }
    return ORCVIO_OK;
}

// statistics and refusals of a finished feature update from the published small block (shared with orcvio_msckf_download)
static int feature_outcome(orcvio_msckf_handle* h, const char* so, int32_t* stats) {
    const int n = h->n, F = h->F;
    const double* dx = reinterpret_cast<const double*>(so + h->oo_dx);
    const int* acc = reinterpret_cast<const int*>(so + h->oo_accept);
    const int* info = reinterpret_cast<const int*>(so);
    int stacked = 0, nacc = 0;
    const int* row_ptr = h->h_row_ptr.data();
    for (int j = 0; j < F; ++j)
        if (acc[j]) { stacked += row_ptr[j + 1] - row_ptr[j]; ++nacc; }
    if (stats) {
        std::memset(stats, 0, sizeof(int32_t) * 8);
        stats[0] = stacked;
        stats[1] = stacked > 0 ? h->NA : 0;
        stats[2] = nacc;
        stats[3] = stacked > 0 ? 1 : 0;
        if (h->flags.discard_large_update) {
            const double nv = std::sqrt(dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
            const double np = std::sqrt(dx[6] * dx[6] + dx[7] * dx[7] + dx[8] * dx[8]);
            stats[4] = (nv > 1.0 || np > 1.5) ? 1 : 0;   // src/orcvio.cpp:4479-4494
        }
        stats[5] = info[0];   // zero-variance directions of the prior (dropped pivots of chol(P))
        stats[6] = info[1];   // pivots of chol(P) below -tol: the prior was not PSD
    }
    if (info[2] != 0 || info[3] != 0) {   // the device left P and x alone (k_finish_sqrt: P+ = P, dx = 0); the resident covariance is intact
        h->ran = false;                    // (nothing to commit)
        g_last_error = "M = s2 I + L^T A L is not positive definite (a prior beyond ~1e16 s2 in scale, or non-finite input): no update";
        return ORCVIO_ERR_NOT_SPD;
    }
    bool finite = true;
    for (int i = 0; i < n; ++i) finite = finite && std::isfinite(dx[i]);
    if (!finite) {   // NaN / Inf somewhere in the inputs (a NaN pivot does not show in the smallest pivot)
        h->ran = false;   // (cov_commit would make a non-finite P+ the resident covariance)
        g_last_error = "non-finite result (NaN / Inf in the prior, the poses or the noise): no update";
        return ORCVIO_ERR_NOT_SPD;
    }
    return ORCVIO_OK;
}

// the update on what stands in the arena (after upload_finalize): results in the pinned output block when this returns
static int io_run(orcvio_msckf_handle* h, bool want_P, bool commit, int32_t* stats) {
    if (h->pw_missing) { g_last_error = "update: tracks were uploaded without positions and have not been triangulated"; return ORCVIO_ERR_INVALID; }
    hipStream_t s = h->stream;
    h->last_stream = s;
    if (h->dl_pending) { HIPCHK(hipStreamSynchronize(h->dl_stream)); h->dl_pending = false; }
    static const bool timing = getenv("ORCVIO_TIMING") != nullptr;
    const auto tl0 = std::chrono::steady_clock::now();
    int rc = run_with_graph(h, h->g_io, io_signature(h, s, want_P, commit), s, [&](bool) { return io_enqueue(h, s, want_P, commit); });
    h->A_deferred = front_defers_assembly(h);   // (a replayed graph does not pass through enqueue_update)
    if (rc != ORCVIO_OK) return rc;
    h->pub_enqueued++;   // (one k_epilogue per launch, captured or not)
    h->ran = true; h->last_update_objects = false; h->last_run_kind = 0; h->last_sharded = false;
    const auto tl1 = std::chrono::steady_clock::now();
    rc = io_wait(h, s);
    if (timing) {
        static int calls = 0;
        if ((++calls % 64) == 0)
            fprintf(stderr, "io_run: launch %.1f us, wait %.1f us\n", std::chrono::duration<double, std::micro>(tl1 - tl0).count(),
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tl1).count());
    }
    if (rc != ORCVIO_OK) { h->ran = false; return rc; }
    const char* so = h->h_stage + h->in_cap;
    const int* info = reinterpret_cast<const int*>(so);
    if (info[8] != 0) {
        // a workgroup of k_front gave up at its device-wide counter (somebody else's kernel held compute units it needed), or a
        // solver wavefront of k_potrf_solve: run the update again on the forked path (plain launches, no in-launch device-wide
        // wait), inside this call; the inputs are in HBM already.  The commit kernels of the first attempt refused themselves.
        HIPCHK(hipMemset(h->d_info + 8, 0, sizeof(int)));
        HIPCHK(hipMemset(h->d_sync, 0, 256));
        h->front_retry_forked = true;
        h->front_fallbacks++;
        int rr = orcvio_msckf_run_update(h, s);
        if (rr == ORCVIO_OK) rr = download_enqueue(h, s, want_P);
        h->front_retry_forked = false;
        if (rr != ORCVIO_OK) { h->ran = false; return rr; }
        HIPCHK(hipStreamSynchronize(s));
        h->dl_pending = false;
        if (info[8] != 0) {   // (the copy has refreshed the pinned block)
            HIPCHK(hipMemset(h->d_info + 8, 0, sizeof(int)));
            h->ran = false;
            g_last_error = "k_potrf_solve / k_front: an in-launch hand-off timed out twice";
            return ORCVIO_ERR_TIMEOUT;
        }
        rr = feature_outcome(h, so, stats);
        if (rr == ORCVIO_OK && commit) rr = orcvio_msckf_cov_commit(h);
        return rr;
    }
    rc = feature_outcome(h, so, stats);
    if (rc != ORCVIO_OK) return rc;
    if (commit) {   // the kernels have written S+ into the spare factor buffer and P+ over the resident covariance
        if (h->factor_opt && h->n_nui == 0) {
            std::swap(h->d_Sres, h->d_Stmp);
            h->fac_n = h->n; h->fac_k = h->kf; h->fac_ld = h->ldz; h->fac_valid = true; h->fac_tail = h->tail;
        } else if (h->n_nui > 0) h->fac_valid = false;   // Schmidt: the nuisance block of P+ is the prior's, so P+ != s2 Z^T Z
        h->res_n = h->n;
    }
    return ORCVIO_OK;
}

int32_t orcvio_msckf_io_begin(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, int32_t n_clones, int32_t n_features,
                              int32_t n_observations, int32_t with_P, orcvio_msckf_io* io) {
    if (!h || !flags || !io) { g_last_error = "io_begin: null argument"; return ORCVIO_ERR_INVALID; }
    const int rc = upload_begin(h, flags, n_clones, n_features, n_observations, with_P != 0, true, "orcvio_msckf_io_begin");
    if (rc != ORCVIO_OK) return rc;
    char* st = h->h_stage;
    io->n = h->n;
    io->poses = reinterpret_cast<double*>(st + h->io_poses);
    io->obs_ptr = reinterpret_cast<int32_t*>(st + h->io_optr);
    io->p_w = reinterpret_cast<double*>(st + h->io_pw);
    io->obs_clone = reinterpret_cast<int32_t*>(st + h->io_oclone);
    io->obs_z = reinterpret_cast<double*>(st + h->io_z);
    io->obs_zvel = h->io_zvel != h->io_z ? reinterpret_cast<double*>(st + h->io_zvel) : nullptr;
    io->P = with_P ? reinterpret_cast<double*>(st + h->io_P) : nullptr;
    const char* so = st + h->in_cap;
    io->dx = reinterpret_cast<const double*>(so + h->oo_dx);
    io->gamma = reinterpret_cast<const double*>(so + h->oo_gamma);
    io->accept = reinterpret_cast<const int32_t*>(so + h->oo_accept);
    io->P_out = reinterpret_cast<const double*>(so + h->oo_Pout);
    h->io_open = true;
    return ORCVIO_OK;
}

int32_t orcvio_msckf_io_update(orcvio_msckf_handle* h, int32_t want_P, int32_t commit, int32_t* stats) {
    if (!h || !h->io_open) { g_last_error = "io_update: call orcvio_msckf_io_begin first"; return ORCVIO_ERR_INVALID; }
    static const bool timing = getenv("ORCVIO_TIMING") != nullptr;
    const auto tf0 = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(h->device));
    const auto tf1 = std::chrono::steady_clock::now();
    int rc = upload_finalize(h, "orcvio_msckf_io_update");
    if (timing) {
        static int calls = 0;
        if ((++calls % 64) == 0)
            fprintf(stderr, "io_update: set device %.1f us, finalize %.1f us\n", std::chrono::duration<double, std::micro>(tf1 - tf0).count(),
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tf1).count());
    }
    if (rc != ORCVIO_OK) return rc;   // (the arena keeps its layout: the caller may repair its arrays and call again)
    h->pw_missing = false;
    h->io_open = true;   // ... and may run the next update of the same shape without a new io_begin
    return io_run(h, want_P != 0, commit != 0, stats);
}

int32_t orcvio_msckf_update_features(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, const orcvio_msckf_window* window,
                                     const orcvio_msckf_tracks* tracks, const double* P, orcvio_msckf_result* result) {
    // the copying form of orcvio_msckf_io_begin / _io_update: the caller's arrays are copied into the arena, the results out of it
    static const bool timing = getenv("ORCVIO_TIMING") != nullptr;   // diagnostics: calls slower than 2 ms are broken down
    if (!result) { g_last_error = "update_features: null result"; return ORCVIO_ERR_INVALID; }
    const auto t0 = std::chrono::steady_clock::now();
    int rc = upload_to_arena(h, flags, window, tracks, P, "orcvio_msckf_update_features");
    if (rc != ORCVIO_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    const bool want_P = result->P_out != nullptr;
    rc = io_run(h, want_P, false, result->stats);
    const auto t3 = std::chrono::steady_clock::now();
    if (rc != ORCVIO_OK) return rc;
    const int n = h->n, F = h->F;
    const char* so = h->h_stage + h->in_cap;
    if (result->dx) std::memcpy(result->dx, so + h->oo_dx, sizeof(double) * n);
    if (result->P_out) std::memcpy(result->P_out, so + h->oo_Pout, sizeof(double) * (size_t)n * n);
    if (result->accept && F > 0) std::memcpy(result->accept, so + h->oo_accept, sizeof(int) * F);
    if (result->gamma && F > 0) std::memcpy(result->gamma, so + h->oo_gamma, sizeof(double) * F);
    if (result->H_thin || result->r_thin || result->K || result->G) {   // optional outputs: the staged download computes them
        orcvio_msckf_result opt = *result;
        opt.dx = nullptr; opt.P_out = nullptr; opt.accept = nullptr; opt.gamma = nullptr;
        rc = orcvio_msckf_download(h, &opt);
        if (rc != ORCVIO_OK) return rc;
    }
    const auto t4 = std::chrono::steady_clock::now();
    if (timing) {
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        static int calls = 0;
        if (us(t0, t4) > 2000.0 || (++calls % 64) == 0)
            fprintf(stderr, "update_features: staging %.0f us, launch + wait %.0f us, unpack %.0f us\n", us(t0, t1), us(t1, t3), us(t3, t4));
    }
    return rc;
}

// The gate alone: gatingTestFeature (src/orcvio.cpp:1953-1976) of featureJacobian_msckf for every listed track, against the
// prior -- what the reference asks of a feature before it lets it ENTER the state as a SLAM feature (:2361-2367) -- without
// an update.  gamma[F], accept[F].
int32_t orcvio_msckf_gate_tracks(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, const orcvio_msckf_window* window,
                                 const orcvio_msckf_tracks* tracks, const double* P, double* gamma, int32_t* accept) {
    if (!gamma || !accept) { g_last_error = "gate_tracks: null output"; return ORCVIO_ERR_INVALID; }
    int rc = orcvio_msckf_upload(h, flags, window, tracks, P);
    if (rc != ORCVIO_OK) return rc;
    if (h->pw_missing) { g_last_error = "gate_tracks: tracks without positions"; return ORCVIO_ERR_INVALID; }
    const int F = h->F;
    if (F == 0) return ORCVIO_OK;
    rc = launch_feature(h, h->stream);
    if (rc != ORCVIO_OK) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(gamma, h->d_gamma, sizeof(double) * F, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(accept, h->d_accept, sizeof(int) * F, hipMemcpyDeviceToHost));
    h->uploaded = false;   // nothing here is an update: the next update call uploads its own tracks
    return ORCVIO_OK;
}

// Object update: OrcVIO::removeLostObjects (src/orcvio.cpp:2154-2193).  Every object block is projected
// onto the left nullspace of its own Hf (SURVEY.md note N3: equal to the reference whenever one object
// arrives per call); the blocks are stacked, gated jointly with dof = sum(rows - cols) and applied in one
// update.  Objects with rows <= cols cannot be projected (math_utils.hpp:292) and are skipped.
// ---- per-stage device times (HIP events between the stages of an update; off unless switched on) -------------------
static void prof_begin(orcvio_msckf_handle* h, hipStream_t s) {
    if (!h->prof_on) return;
    h->prof_n = 0;
    h->prof_names.clear();
    if (h->prof_ev.empty()) { h->prof_ev.resize(24); for (auto& e : h->prof_ev) (void)hipEventCreate(&e); }
    (void)hipEventRecord(h->prof_ev[0], s);
    h->prof_n = 1;
}
static void prof_mark(orcvio_msckf_handle* h, hipStream_t s, const char* name) {
    if (!h->prof_on || h->prof_n == 0 || h->prof_n >= (int)h->prof_ev.size()) return;
    (void)hipEventRecord(h->prof_ev[h->prof_n++], s);
    h->prof_names.push_back(name);
}

// ---- object update: device scratch and host staging ------------------------------------------------------------------
// ONE input arena per object update, mirrored in pinned host memory (grown on demand) and copied with one asynchronous
// copy: [doubles ... | ints ...].  Device-side scratch (row arrays written by k_object_rows_batch, Cd, Sg, Gff, factors)
// is separate and never crosses PCIe.
struct ObjPlan {
    int nobj = 0, rows_tot = 0, no_max = 0, ngroups = 0;
    int NOP = 0, ldf = 0;
    // device pointers
    int *d_ridx = nullptr, *d_rowptr = nullptr, *d_clone = nullptr, *d_cols = nullptr;
    ObjGroup* d_groups = nullptr;
    double *d_hx = nullptr, *d_hf = nullptr, *d_res = nullptr;
    double *d_Cd = nullptr, *d_Sg = nullptr, *d_Gff = nullptr;
    // arrow structure of Hf (structured Householder QR instead of chol(Hf^T Hf)); arrow = false: the Gram route
    bool arrow = false;
    int Kmax = 0, rows_max = 0;
    ObjArrow* d_arrow = nullptr;
    int2* d_kp_range = nullptr;
    int* d_kp_rows = nullptr;
    double* d_Rarrow = nullptr;
    double *d_Hr = nullptr, *d_Hfr = nullptr;   // [nobj][N][NOP] per-clone parts of Hf^T r; [nobj][NOP + 1] their sums and |r|^2
    double* d_Bred = nullptr;   // [rows][9] the border of every row after the keypoint blocks have been eliminated
};
static int obj_stage_reserve(orcvio_msckf_handle* h, size_t bytes) {
    if (bytes <= h->obj_stage_cap) return ORCVIO_OK;
    HIPCHK(hipDeviceSynchronize());
    if (h->h_obj_stage) (void)hipHostFree(h->h_obj_stage);
    if (h->d_obj_in) (void)hipFree(h->d_obj_in);
    h->h_obj_stage = nullptr; h->d_obj_in = nullptr; h->obj_stage_cap = 0;
    const size_t cap = (bytes * 3 / 2 + 4095) & ~(size_t)4095;
    HIPCHK(hipHostMalloc(&h->h_obj_stage, cap, hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->h_obj_stage_dev), h->h_obj_stage, 0));
    HIPCHK(hipMalloc(&h->d_obj_in, cap));
    h->obj_stage_cap = cap;
    return ORCVIO_OK;
}
// device scratch of the compression: row arrays [Hx6 | HfR (ld ldf) | res | row_clone | row_cols] and [Cd | Sg | Gff]
static int objects_scratch(orcvio_msckf_handle* h, ObjPlan* pl) {
    const int NAP = h->NAP, N = h->N;
    pl->NOP = round_up(pl->no_max, 16);
    pl->ldf = round_up(pl->no_max + 1, 16);
    const size_t rows = (size_t)pl->rows_tot, nobj = (size_t)pl->nobj;
    int rc;
    if ((rc = grow(&h->d_obj_i, &h->cap_obj_i, 2 * rows + 16)) != ORCVIO_OK) return rc;
    if ((rc = grow(&h->d_objH, &h->cap_objH, rows * (6 + pl->ldf + 1) + 16)) != ORCVIO_OK) return rc;
    const size_t nCd = nobj * pl->NOP * NAP, nSg = nobj * N * 64, nHr = nobj * N * pl->NOP, nGff = nobj * pl->ldf * pl->ldf;
    const size_t nRa = nobj * (size_t)arrow_stride(pl->Kmax > 0 ? pl->Kmax : 1);
    if ((rc = grow(&h->d_Gobj, &h->cap_Gobj, nCd + nSg + nHr + nGff + nRa + rows * 9 + nobj * (pl->NOP + 1))) != ORCVIO_OK) return rc;
    if ((rc = grow(&h->d_RF, &h->cap_RF, nobj * ((size_t)pl->NOP * pl->NOP + 7 * 256))) != ORCVIO_OK) return rc;
    if ((rc = grow(&h->d_Yobj, &h->cap_Yobj, nobj * pl->NOP * NAP)) != ORCVIO_OK) return rc;
    pl->d_clone = h->d_obj_i;
    pl->d_cols = h->d_obj_i + rows;
    pl->d_hx = h->d_objH;
    pl->d_hf = pl->d_hx + rows * 6;
    pl->d_res = pl->d_hf + rows * pl->ldf;
    pl->d_Cd = h->d_Gobj;
    pl->d_Sg = pl->d_Cd + nCd;
    pl->d_Hr = pl->d_Sg + nSg;
    pl->d_Gff = pl->d_Hr + nHr;
    pl->d_Rarrow = pl->d_Gff + nGff;
    pl->d_Bred = pl->d_Rarrow + nRa;
    pl->d_Hfr = pl->d_Bred + rows * 9;
    return ORCVIO_OK;
}

// From the compact rows in device memory to this rank's block in dst (P already in d_P).  The prior's Cholesky factor is
// forked to the side stream (joined by objects_finish).
static int objects_pipeline(orcvio_msckf_handle* h, hipStream_t s, double* dst, const ObjPlan& pl, bool zeroed = false, bool forked = false) {
    // zeroed: k_object_rows_batch has cleared Cd / Sg / Hr and the pivot counters;  forked: the caller has forked the Cholesky of
    // the prior already (right behind the copy of P, before it staged the tracks)
    const int NA = h->NA, NAP = h->NAP, N = h->N, nobj = pl.nobj, NOP = pl.NOP, ldf = pl.ldf, no_max = pl.no_max;
    double* d_RF = h->d_RF;
    double* d_DinvF = h->d_RF + (size_t)nobj * NOP * NOP;
    // zero: C (clones an object does not see) and the clone tiles; for the Gram route also Y (padded rows) and R_F
    // (strictly-lower tiles) -- the arrow route writes every entry of Y it reads
    if (!pl.arrow) {
        HIPCHK(hipMemsetAsync(h->d_Yobj, 0, sizeof(double) * (size_t)nobj * NOP * NAP, s));
        HIPCHK(hipMemsetAsync(d_RF, 0, sizeof(double) * (size_t)nobj * NOP * NOP, s));
    }
    if (!zeroed) HIPCHK(hipMemsetAsync(pl.d_Cd, 0, sizeof(double) * ((size_t)nobj * NOP * NAP + (size_t)nobj * N * 64 + (size_t)nobj * N * NOP), s));   // Cd, Sg, Hr are adjacent
    int rc = forked ? ORCVIO_OK : launch_prior_fork(h, s);   // Cholesky of P on the side stream
    if (rc != ORCVIO_OK) return rc;
    prof_mark(h, s, "rows+copies");
    {   // cross products (they also give Hf^T r and |r|^2), the keypoint blocks of the structured QR (arrow route) or the compact Grams
        // [Hf | r]^T [Hf | r] (Gram route, which needs F = Hf^T Hf): one launch
        const int nbf = ldf / 16, gram_tiles = pl.arrow ? 0 : nbf * (nbf + 1) / 2, kp_blocks = pl.arrow ? (pl.Kmax + 1 + 7) / 8 : 0;
        const int blocks = (pl.ngroups + 7) / 8 + gram_tiles * nobj + kp_blocks * nobj;
        hipLaunchKernelGGL(k_obj_front, dim3(blocks), dim3(512), 0, s, pl.d_groups, pl.ngroups, pl.d_ridx, pl.d_hx, pl.d_hf, ldf, no_max,
                           h->flags.leg_dim - 15, NAP, NOP, N, pl.d_Cd, pl.d_Sg, pl.d_Hr, pl.d_rowptr, pl.d_Gff, nobj, gram_tiles, pl.d_arrow,
                           pl.d_kp_range, pl.d_kp_rows, pl.Kmax, pl.d_Rarrow, pl.d_Bred, kp_blocks);
    }
    prof_mark(h, s, "k_obj_front");
    const int solve_xblocks = (NAP + 255) / 256, nb_solve = pl.arrow ? solve_xblocks * nobj : 0;
    const bool fuse_border = pl.arrow && solve_xblocks == 1;   // border QR + Y + sum B in one launch (one solve workgroup per object)
    if (pl.arrow && !fuse_border) {
        // R of Hf by structured Householder QR (cond(Hf), not its square: msckf_kernels.hpp); the keypoint blocks are done, the border:
        if (!zeroed) HIPCHK(hipMemsetAsync(h->d_info + 4, 0, sizeof(int) * 2, s));
#define LAUNCH_BORDER(RPT) hipLaunchKernelGGL(k_obj_border_qr<RPT>, dim3(nobj), dim3(256), 0, s, pl.d_arrow, (const double*)pl.d_Bred, pl.Kmax, \
                                              pl.d_Rarrow, (const double*)pl.d_Hr, (const double*)pl.d_Sg, N, NOP, pl.d_Hfr)
        if (pl.rows_max <= 512) LAUNCH_BORDER(2);
        else if (pl.rows_max <= 1024) LAUNCH_BORDER(4);
        else LAUNCH_BORDER(8);
#undef LAUNCH_BORDER
        prof_mark(h, s, "k_obj_border_qr(Hf)");
    } else if (!pl.arrow) {
    // (Hf without the arrow structure of ObjectLM's state: the Gram route.)  F_o = Hf^T Hf (lower tiles of Gff) -> R_F ;
    // Y_o = L_F^-1 C_o, C_o = [Cd_o | Hf^T r]
    {
        const int nbf = NOP / 16, need = potrf_slots_needed(nbf);
        if (!zeroed) HIPCHK(hipMemsetAsync(h->d_info + 4, 0, sizeof(int) * 2, s));   // the batched factorisation ADDS its pivot counters
        const double tolF = (double)no_max * 2.220446049250313e-16;
        if (need <= 4)
            hipLaunchKernelGGL(k_potrf_reg<4>, dim3(nobj), dim3(512), 0, s, pl.d_Gff, ldf, no_max, tolF, d_RF, NOP, d_DinvF, h->d_info + 4,
                               (unsigned long long*)nullptr, (size_t)ldf * ldf, (size_t)NOP * NOP, (size_t)7 * 256, 1);
        else
            hipLaunchKernelGGL(k_potrf_reg<8>, dim3(nobj), dim3(512), 0, s, pl.d_Gff, ldf, no_max, tolF, d_RF, NOP, d_DinvF, h->d_info + 4,
                               (unsigned long long*)nullptr, (size_t)ldf * ldf, (size_t)NOP * NOP, (size_t)7 * 256, 1);
        prof_mark(h, s, "k_potrf_reg(F) batched");
        const int nwave = (NA + 1 + 15) / 16;
        hipLaunchKernelGGL(k_trsm_lds, dim3((nwave + 3) / 4, nobj), dim3(256), 0, s, d_RF, NOP, d_DinvF, no_max,
                           pl.d_Cd, (long)NAP, 1L, NA, pl.d_Gff + (size_t)no_max * ldf, 1L, h->d_Yobj, NAP,
                           (size_t)NOP * NOP, (size_t)7 * 256, (size_t)NOP * NAP, (size_t)NOP * NAP, (size_t)ldf * ldf);
        prof_mark(h, s, "k_trsm_lds(Y) batched");
    }
    }
    // Y_o = R^-T C_o (arrow route) and sum_o B_o in one launch; then A' = sum_o B_o - Y^T Y (Y = all Y_o stacked; padded rows are zero)
    if (fuse_border) {
        if (!zeroed) HIPCHK(hipMemsetAsync(h->d_info + 4, 0, sizeof(int) * 2, s));
        const dim3 grid(nobj + (NAP * NAP + 255) / 256);
        const size_t lds = sizeof(double) * arrow_stride(pl.Kmax > 0 ? pl.Kmax : 1);
#define LAUNCH_BSA(RPT) hipLaunchKernelGGL(k_obj_border_solve_assemble<RPT>, grid, dim3(256), lds, s, pl.d_arrow, (const double*)pl.d_Bred, pl.Kmax, \
                                           pl.d_Rarrow, (const double*)pl.d_Hr, (const double*)pl.d_Sg, N, NOP, pl.d_Hfr, (const double*)pl.d_Cd, NAP, NA, \
                                           h->d_Yobj, h->d_info + 4, nobj, h->flags.leg_dim - 15, h->d_Ab)
        if (pl.rows_max <= 512) LAUNCH_BSA(2);
        else if (pl.rows_max <= 1024) LAUNCH_BSA(4);
        else LAUNCH_BSA(8);
#undef LAUNCH_BSA
        prof_mark(h, s, "k_obj_border_solve_assemble");
    } else
    {   // |r|^2 per object: arrow route Hfr[o][NOP] (k_obj_border_qr), Gram route the corner of the compact Gram
        const double* rr = pl.arrow ? pl.d_Hfr + NOP : pl.d_Gff + (size_t)no_max * ldf + no_max;
        const size_t rr_stride = pl.arrow ? (size_t)NOP + 1 : (size_t)ldf * ldf;
        hipLaunchKernelGGL(k_obj_solve_assemble, dim3(nb_solve + (NAP * NAP + 255) / 256), dim3(256), sizeof(double) * arrow_stride(pl.Kmax > 0 ? pl.Kmax : 1), s,
                           nb_solve, solve_xblocks, pl.d_arrow, pl.d_Rarrow, pl.Kmax, pl.d_Cd, NOP, NAP, NA, (const double*)pl.d_Hfr, h->d_Yobj, h->d_info + 4,
                           pl.d_Sg, nobj, N, h->flags.leg_dim - 15, rr, rr_stride, h->d_Ab);
    }
    hipLaunchKernelGGL(k_gemm, dim3((NAP / 16) * (NAP / 16)), dim3(256), 0, s, h->d_Yobj, 1L, (long)NAP, h->d_Yobj, (long)NAP, 1L,
                       NAP, NAP, nobj * NOP, -1.0, 0.0, 0, dst, (long)NAP, 1L, h->d_Ab);
    HIPCHK(hipGetLastError());
    prof_mark(h, s, fuse_border ? "k_gemm(A')" : "k_obj_assemble_B+k_gemm(A')");
    return ORCVIO_OK;
}

// Prior of an object update: the caller's P staged through pinned memory (one asynchronous copy; the caller's buffer is
// free when the call returns), or the resident covariance in place (no copy at all).
static int objects_prior(orcvio_msckf_handle* h, hipStream_t s, const double* P, const char* who) {
    const int n = h->n;
    if (!P && h->res_n != n) { g_last_error = std::string(who) + ": P == NULL but the resident covariance does not match the window"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipStreamSynchronize(h->stream));   // the pinned staging of the previous call is free again
    if (s != h->stream) HIPCHK(hipStreamSynchronize(s));
    if (h->dl_pending) { HIPCHK(hipStreamSynchronize(h->dl_stream)); h->dl_pending = false; }
    layout_inputs(h, h->N, 0, 0, false, P != nullptr, n);
    layout_outputs(h, n, 1);
    if (P) {
        std::memcpy(h->h_stage + h->io_P, P, sizeof(double) * (size_t)n * n);
        const int ri = launch_ingest(h, s, h->h_stage_dev + h->io_P, h->d_P, sizeof(double) * (size_t)n * n, obj_ingest_kernel());
        if (ri != ORCVIO_OK) return ri;
    }
    return ORCVIO_OK;
}

// ORCVIO_OPT_REF_STACK_HF: System::processObjects stacks Hx, Hf and r of all objects VERTICALLY, Hf with its 45 columns
// shared (ros_wrapper/src/orcvio/src/System.cpp:684-702), and removeLostObjects projects the whole stack against that one Hf
// (src/orcvio.cpp:2154-2193): the objects become ONE block of rows.  Host arrays of the staged update are rewritten in
// place: one object, its rows regrouped by clone through the index list.
static void merge_objects_ref_stack(int N, int nobj, int rows_tot, int* ridx, int* rowptr, ObjGroup* groups, int* ng) {
    std::vector<int> old_ridx(ridx, ridx + rows_tot);
    std::vector<ObjGroup> old(groups, groups + *ng);
    int pos = 0, g = 0;
    for (int c = 0; c < N; ++c) {
        const int g0 = pos;
        for (const ObjGroup& q : old)
            if (q.clone == c)
                for (int k = q.r0; k < q.r1; ++k) ridx[pos++] = old_ridx[k];
        if (pos > g0) groups[g++] = ObjGroup{g0, pos, c, 0};
    }
    *ng = g;
    rowptr[0] = 0; rowptr[1] = rows_tot;
    (void)nobj;
}

// Arrow structure of the objects' Hf for the structured QR (k_obj_arrow_qr): rowkp[row] = keypoint block of the row (-1: a
// border-only row), Ks[o] = keypoint blocks of object o.  Fills arrows / ranges / kp_rows (host mirrors of the device arrays);
// returns false if some object does not fit the kernel's limits (<= 128 rows per keypoint, <= 2048 rows per object).
static bool build_arrow(const int* rowkp, const int* rowptr, const int* Ks, int nobj, ObjArrow* arrows, int2* ranges, int* kp_rows,
                        int* Kmax, int* rows_max) {
    // per object K + 1 ranges into kp_rows: the rows of keypoint block 0 .. K-1, then the border-only rows
    int off = 0, pos = 0;
    *Kmax = 0; *rows_max = 0;
    for (int o = 0; o < nobj; ++o) {
        const int r0 = rowptr[o], r1 = rowptr[o + 1], K = Ks[o];
        if (r1 - r0 > 2048 || K > 34) return false;
        int cnt[36] = {0};
        for (int r = r0; r < r1; ++r) cnt[rowkp[r] >= 0 ? rowkp[r] : K]++;
        int start[36];
        for (int k = 0; k <= K; ++k) {
            if (k < K && cnt[k] > 128) return false;
            start[k] = pos;
            ranges[off + k] = int2{pos, pos + cnt[k]};
            pos += cnt[k];
        }
        for (int r = r0; r < r1; ++r) kp_rows[start[rowkp[r] >= 0 ? rowkp[r] : K]++] = r;
        arrows[o] = ObjArrow{r0, r1 - r0, K, off};
        off += K + 1;
        if (K > *Kmax) *Kmax = K;
        if (r1 - r0 > *rows_max) *rows_max = r1 - r0;
    }
    return true;
}

// window-dependent sizes of an object update (no tracks)
static int objects_problem(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, int N, const double* P, const char* who) {
    if (flags->leg_dim != 22 && flags->leg_dim != 46) { g_last_error = std::string(who) + ": leg_dim must be 22 or 46"; return ORCVIO_ERR_INVALID; }
    if (N < 1 || N > h->maxN) { g_last_error = std::string(who) + ": window exceeds capacity"; return ORCVIO_ERR_CAPACITY; }
    h->flags = *flags;
    h->io_open = false;   // (the object update lays the arenas out for itself: a feature update needs a new io_begin)
    h->N = N; h->F = 0; h->nobs = 0;
    h->n = flags->leg_dim + 6 * N + h->n_extra;   // (n_extra: states behind the clones that no row of this update touches)
    h->NA = h->ekf_mode ? h->n - 15 : flags->leg_dim + 6 * N - 15;   // (EKF-SLAM rows reach into the extra states)
    h->ekf_F = 0; h->dense_rows = 0; h->new_F = 0;
    if (h->n > h->n_max) { g_last_error = "window + extra states exceed the handle's capacity"; return ORCVIO_ERR_CAPACITY; }
    h->NAP = round_up(h->NA + 1, 16);
    select_prior_factor(h, P != nullptr);
    h->NP = round_up(h->n > h->kf ? h->n : h->kf, 16);
    h->ldz = round_up(h->n + 1, 16);
    h->reg_path = (h->NP / 16) <= 14;
    select_tail(h);
    { const int rcl = factor_layout_clean(h); if (rcl != ORCVIO_OK) return rcl; }
    h->m_tot = 0; h->Mmax = 2; h->chunks = 1; h->rows_per_chunk = 8;
    h->h_row_ptr.assign(1, 0);
    return ORCVIO_OK;
}

// Local part of an object update: this rank's objects -> its compressed block [A' b'; b'^T c'] (NAP x NAP) in d_dst
// (the handle's own block if NULL), Cholesky of P forked on the side stream.
int32_t orcvio_msckf_objects_local(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, int32_t n_clones,
                                   const orcvio_msckf_object_rows* objs, int32_t n_objects, const double* P, double* d_dst,
                                   int32_t* dof_out, void* stream) {
    if (!h || !flags || n_objects < 0 || (n_objects > 0 && !objs)) { g_last_error = "objects_local: null argument"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    { const int rp = objects_problem(h, flags, n_clones, P, "objects_local"); if (rp != ORCVIO_OK) return rp; }
    const int N = n_clones, NAP = h->NAP;
    // usable objects, row offsets, widest object state
    std::vector<int> use;
    ObjPlan pl;
    int dof = 0;
    for (int o = 0; o < n_objects; ++o) {
        const orcvio_msckf_object_rows& ob = objs[o];
        if (ob.n_rows < 0 || ob.n_obj_cols < 1 || ob.n_obj_cols > 112) { g_last_error = "objects_local: bad block shape (object state columns must be 1..112)"; return ORCVIO_ERR_INVALID; }
        if (ob.n_rows > 0 && (!ob.row_clone || !ob.Hx6 || !ob.Hf || !ob.res)) { g_last_error = "objects_local: null block arrays"; return ORCVIO_ERR_INVALID; }
        if (h->ref_stack_hf ? ob.n_rows == 0 : ob.n_rows <= ob.n_obj_cols) continue;   // nullspace_project_inplace_svd returns false
        for (int r = 0; r < ob.n_rows; ++r)
            if (ob.row_clone[r] < 0 || ob.row_clone[r] >= N) { g_last_error = "objects_local: row_clone out of range"; return ORCVIO_ERR_INVALID; }
        if (h->ref_stack_hf && pl.no_max > 0 && ob.n_obj_cols != pl.no_max) { g_last_error = "objects_local: ORCVIO_OPT_REF_STACK_HF needs equal object state sizes"; return ORCVIO_ERR_INVALID; }
        use.push_back(o);
        pl.rows_tot += ob.n_rows;
        dof += ob.n_rows - ob.n_obj_cols;
        if (ob.n_obj_cols > pl.no_max) pl.no_max = ob.n_obj_cols;
    }
    if (h->ref_stack_hf) {   // one stacked block: projectable only if it has more rows than columns; dof = its rows - columns
        if (pl.rows_tot <= pl.no_max) { use.clear(); pl.rows_tot = 0; }
        dof = use.empty() ? 0 : pl.rows_tot - pl.no_max;
    }
    pl.nobj = (int)use.size();
    hipStream_t s = pick_stream(h, stream);
    h->last_stream = s;
    double* dst = d_dst ? d_dst : h->d_Ab;
    if (dof_out) *dof_out = dof;
    prof_begin(h, s);
    { const int rp = objects_prior(h, s, P, "objects_local"); if (rp != ORCVIO_OK) return rp; }
    h->uploaded = true;
    h->objects_mode = true;
    h->obj_dof = dof; h->obj_rows = pl.rows_tot; h->obj_count = pl.nobj;
    if (!h->d_obj_accept) { HIPCHK(hipMalloc(&h->d_obj_accept, sizeof(int) * 4)); HIPCHK(hipMalloc(&h->d_obj_gamma, sizeof(double) * 4)); }
    if (pl.nobj == 0) {   // nothing usable on this rank: a zero block
        HIPCHK(hipMemsetAsync(dst, 0, sizeof(double) * (size_t)NAP * NAP, s));
        return launch_prior_fork(h, s);
    }
    int rc = objects_scratch(h, &pl);
    if (rc != ORCVIO_OK) return rc;
    // staging arena: [Hx6 rows x 6 | HfR rows x ldf] doubles, then ints [ridx rows | rowptr nobj+1 | arrows 4 nobj | kp ranges
    // 2 x sum K | kp_rows rows | groups 4 x <= nobj N]
    const size_t rows = (size_t)pl.rows_tot, ldf = (size_t)pl.ldf;
    size_t sumK = 0;
    for (int o : use) sumK += (size_t)((objs[o].n_obj_cols - 9) / 3 > 0 ? (objs[o].n_obj_cols - 9) / 3 : 0);
    const size_t nd = rows * 6 + rows * ldf;
    const size_t o_ridx = 0, o_rowptr = o_ridx + rows, o_arrow = o_rowptr + pl.nobj + 1, o_range = o_arrow + (size_t)4 * pl.nobj,
                 o_kprows = o_range + 2 * (sumK + (size_t)pl.nobj), o_groups = o_kprows + rows, ni = o_groups + (size_t)4 * pl.nobj * N;
    if ((rc = obj_stage_reserve(h, nd * 8 + ni * 4)) != ORCVIO_OK) return rc;
    double* hd = reinterpret_cast<double*>(h->h_obj_stage);
    int* hi = reinterpret_cast<int*>(h->h_obj_stage + nd * 8);
    double* hx = hd;
    double* hf = hd + rows * 6;
    int* ridx = hi + o_ridx;
    int* rowptr = hi + o_rowptr;
    ObjGroup* groups = reinterpret_cast<ObjGroup*>(hi + o_groups);
    std::memset(hf, 0, sizeof(double) * rows * ldf);
    std::vector<int>& rowkp = h->obj_fnr;   // (scratch) keypoint block of every row, -1: border only, -2: no arrow structure
    rowkp.assign(rows, -1);
    std::vector<int> Ks(pl.nobj, 0);
    bool structured = true;
    int r0 = 0, ng = 0;
    rowptr[0] = 0;
    for (size_t ui = 0; ui < use.size(); ++ui) {
        const orcvio_msckf_object_rows& ob = objs[use[ui]];
        const int nc = ob.n_obj_cols;
        const bool shape_ok = nc >= 9 && (nc - 9) % 3 == 0;   // [pose 6 | shape 3 | 3 per keypoint], ObjectLM.h:117-123
        Ks[ui] = shape_ok ? (nc - 9) / 3 : 0;
        structured = structured && shape_ok;
        int cnt[ORCVIO_MAX_CLONES + 1] = {0};
        for (int r = 0; r < ob.n_rows; ++r) {
            std::memcpy(hx + (size_t)(r0 + r) * 6, ob.Hx6 + (size_t)r * 6, 6 * sizeof(double));
            double* row = hf + (size_t)(r0 + r) * ldf;
            const double* src = ob.Hf + (size_t)r * nc;
            std::memcpy(row, src, nc * sizeof(double));
            row[pl.no_max] = ob.res[r];
            cnt[ob.row_clone[r] + 1]++;
            if (structured) {   // the row's non-zeros behind the border must lie in ONE keypoint block
                int blk = -1;
                for (int c = 9; c < nc; ++c)
                    if (src[c] != 0.0) {
                        const int b = (c - 9) / 3;
                        if (blk >= 0 && b != blk) { structured = false; break; }
                        blk = b;
                    }
                rowkp[r0 + r] = blk;
            }
        }
        for (int c = 0; c < N; ++c) cnt[c + 1] += cnt[c];
        for (int c = 0; c < N; ++c)
            if (cnt[c + 1] > cnt[c]) groups[ng++] = ObjGroup{r0 + cnt[c], r0 + cnt[c + 1], c, (int)ui};
        int fill[ORCVIO_MAX_CLONES + 1];
        std::memcpy(fill, cnt, sizeof(int) * (N + 1));
        for (int r = 0; r < ob.n_rows; ++r) ridx[r0 + fill[ob.row_clone[r]]++] = r0 + r;   // rows grouped by clone (stable)
        r0 += ob.n_rows;
        rowptr[ui + 1] = r0;
    }
    if (h->ref_stack_hf && pl.nobj > 1) {
        merge_objects_ref_stack(N, pl.nobj, pl.rows_tot, ridx, rowptr, groups, &ng);
        pl.nobj = 1; h->obj_count = 1; structured = false;
        rc = objects_scratch(h, &pl);   // the scratch of ONE block: Cd | Sg | Hr adjacent again (the row arrays depend on the row count only)
        if (rc != ORCVIO_OK) return rc;
    }
    pl.ngroups = ng;
    pl.arrow = structured && h->arrow_opt &&
               build_arrow(rowkp.data(), rowptr, Ks.data(), pl.nobj, reinterpret_cast<ObjArrow*>(hi + o_arrow),
                           reinterpret_cast<int2*>(hi + o_range), hi + o_kprows, &pl.Kmax, &pl.rows_max);
    if (pl.arrow) { rc = objects_scratch(h, &pl); if (rc != ORCVIO_OK) return rc; }   // (room for the arrow factors)
    // device views of the arena
    double* dd = reinterpret_cast<double*>(h->d_obj_in);
    int* di = reinterpret_cast<int*>(h->d_obj_in + nd * 8);
    pl.d_hx = dd; pl.d_hf = dd + rows * 6;
    pl.d_ridx = di + o_ridx; pl.d_rowptr = di + o_rowptr; pl.d_groups = reinterpret_cast<ObjGroup*>(di + o_groups);
    pl.d_arrow = reinterpret_cast<ObjArrow*>(di + o_arrow); pl.d_kp_range = reinterpret_cast<int2*>(di + o_range); pl.d_kp_rows = di + o_kprows;
    { const int ri = launch_ingest(h, s, h->h_obj_stage_dev, h->d_obj_in, nd * 8 + (o_groups + (size_t)4 * ng) * 4, obj_ingest_kernel()); if (ri != ORCVIO_OK) return ri; }
    return objects_pipeline(h, s, dst, pl);
}

// The same from object TRACKS (state at the LM optimum + observations): the residual rows and Jacobians of SURVEY 8a rows
// 12-16 are evaluated on the device (k_object_rows) straight into the compact row arrays -- nothing but the tracks
// crosses PCIe.  Tracks whose in-window rows do not exceed their state columns are skipped (math_utils.hpp:292).
int32_t orcvio_msckf_objects_local_tracks(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, const orcvio_object_eval_flags* fl,
                                          int32_t n_clones, const orcvio_object_track* tracks, int32_t n_tracks, const double* P,
                                          double* d_dst, int32_t* dof_out, void* stream) {
    if (!h || !flags || !fl || n_tracks < 0 || (n_tracks > 0 && !tracks)) { g_last_error = "objects_local_tracks: null argument"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    static const bool timing = getenv("ORCVIO_TIMING") != nullptr;   // diagnostics: host wall time of the parts of this call
    const auto tt0 = std::chrono::steady_clock::now();
    { const int rp = objects_problem(h, flags, n_clones, P, "objects_local_tracks"); if (rp != ORCVIO_OK) return rp; }
    const int N = n_clones, NAP = h->NAP;
    typedef ObjUse Use;
    std::vector<Use>& use = h->obj_use;
    use.clear();
    ObjPlan pl;
    int dof = 0, Fmax = 1;
    size_t nd = 0, ni = 0;   // staged doubles / ints
    // pass 1: which tracks are usable, sizes
    for (int t = 0; t < n_tracks; ++t) {
        const orcvio_object_track& ob = tracks[t];
        if (!ob.wTo || !ob.shape || !ob.kps || !ob.frame_wTc || !ob.frame_zs || !ob.frame_bbox || !ob.frame_clone) { g_last_error = "objects_local_tracks: null track arrays"; return ORCVIO_ERR_INVALID; }
        const int K = ob.n_keypoints, F = ob.n_frames, ncol = 9 + 3 * K;
        if (K < 1 || K > 34 || F < 1) { g_last_error = "objects_local_tracks: 1..34 keypoints (object state <= 112 columns), >= 1 frame"; return ORCVIO_ERR_INVALID; }
        int rows = 0;
        for (int f = 0; f < F; ++f) {
            if (ob.frame_clone[f] >= N) { g_last_error = "objects_local_tracks: frame_clone out of the window"; return ORCVIO_ERR_INVALID; }
            if (ob.frame_clone[f] < 0) continue;
            int nv = 0;
            const double* zs = ob.frame_zs + (size_t)f * K * 2;
            for (int k = 0; k < K; ++k)
                if (std::isfinite(zs[2 * k]) && std::isfinite(zs[2 * k + 1])) ++nv;   // row finite test, ObjectLM.cpp:171-198
            rows += 2 * nv + 4;
        }
        if (h->ref_stack_hf ? rows == 0 : rows <= ncol) continue;   // nullspace_project_inplace_svd returns false
        if (h->ref_stack_hf && pl.no_max > 0 && ncol != pl.no_max) { g_last_error = "objects_local_tracks: ORCVIO_OPT_REF_STACK_HF needs equal object state sizes"; return ORCVIO_ERR_INVALID; }
        use.push_back(Use{t, pl.rows_tot, rows, ncol, nd, ni});
        nd += 16 + 3 + (size_t)3 * K + (size_t)F * (16 + 2 * K + 4);
        ni += (size_t)2 * F;
        pl.rows_tot += rows;
        dof += rows - ncol;
        if (ncol > pl.no_max) pl.no_max = ncol;
        if (F > Fmax) Fmax = F;
    }
    if (h->ref_stack_hf) {
        if (pl.rows_tot <= pl.no_max) { use.clear(); pl.rows_tot = 0; }
        dof = use.empty() ? 0 : pl.rows_tot - pl.no_max;
    }
    pl.nobj = (int)use.size();
    hipStream_t s = pick_stream(h, stream);
    h->last_stream = s;
    double* dst = d_dst ? d_dst : h->d_Ab;
    if (dof_out) *dof_out = dof;
    prof_begin(h, s);
    const auto tt1 = std::chrono::steady_clock::now();
    { const int rp = objects_prior(h, s, P, "objects_local_tracks"); if (rp != ORCVIO_OK) return rp; }
    const auto tt2 = std::chrono::steady_clock::now();
    h->uploaded = true;
    h->objects_mode = true;
    h->obj_dof = dof; h->obj_rows = pl.rows_tot; h->obj_count = pl.nobj;
    if (!h->d_obj_accept) { HIPCHK(hipMalloc(&h->d_obj_accept, sizeof(int) * 4)); HIPCHK(hipMalloc(&h->d_obj_gamma, sizeof(double) * 4)); }
    if (pl.nobj == 0) {
        HIPCHK(hipMemsetAsync(dst, 0, sizeof(double) * (size_t)NAP * NAP, s));
        return launch_prior_fork(h, s);
    }
    int rc = launch_prior_fork(h, s);   // the Cholesky of P runs on the side stream while the tracks are staged and their rows evaluated
    if (rc != ORCVIO_OK) return rc;
    rc = objects_scratch(h, &pl);
    if (rc != ORCVIO_OK) return rc;
    // staging arena: [track data (doubles) | kernel arguments (doubles)], then ints [frame_clone, frame_row0 per track | ridx |
    // rowptr | arrows | kp ranges | kp_rows | groups]
    static_assert(sizeof(ObjEvalArgs) % sizeof(double) == 0, "ObjEvalArgs is copied as doubles");
    const size_t arg_dbl = sizeof(ObjEvalArgs) / sizeof(double);
    const size_t args_off = nd;
    nd += arg_dbl * use.size();
    const size_t rows = (size_t)pl.rows_tot;
    size_t sumK = 0;
    for (const Use& u : use) sumK += (size_t)tracks[u.t].n_keypoints;
    const size_t o_ridx = ni, o_rowptr = o_ridx + rows, o_arrow = o_rowptr + pl.nobj + 1, o_range = o_arrow + (size_t)4 * pl.nobj,
                 o_kprows = o_range + 2 * (sumK + (size_t)pl.nobj), o_groups = o_kprows + rows, ni_tot = o_groups + (size_t)4 * pl.nobj * N;
    if ((rc = obj_stage_reserve(h, nd * 8 + ni_tot * 4)) != ORCVIO_OK) return rc;
    double* hd = reinterpret_cast<double*>(h->h_obj_stage);
    int* hi = reinterpret_cast<int*>(h->h_obj_stage + nd * 8);
    double* dd = reinterpret_cast<double*>(h->d_obj_in);
    int* di = reinterpret_cast<int*>(h->d_obj_in + nd * 8);
    int* ridx = hi + o_ridx;
    int* rowptr = hi + o_rowptr;
    ObjGroup* groups = reinterpret_cast<ObjGroup*>(hi + o_groups);
    // The arrow structure of the structured QR of Hf (per object K + 1 row lists: the rows of every keypoint block, then the
    // border-only rows) is written here, frame by frame, straight from the observation masks -- the order of build_arrow (rows
    // ascending within a list), without a pass over the 15 000 rows of a config-3 update per array.
    ObjArrow* arrows = reinterpret_cast<ObjArrow*>(hi + o_arrow);
    int2* kp_ranges = reinterpret_cast<int2*>(hi + o_range);
    int* kp_rows = hi + o_kprows;
    const bool want_arrow = h->arrow_opt && !(h->ref_stack_hf && pl.nobj > 1);
    bool arrow_ok = want_arrow;
    int arrow_off = 0, arrow_pos = 0, arrow_Kmax = 0, arrow_rows_max = 0;
    int ng = 0;
    rowptr[0] = 0;
    for (size_t ui = 0; ui < use.size(); ++ui) {
        const Use& u = use[ui];
        const orcvio_object_track& ob = tracks[u.t];
        const int K = ob.n_keypoints, F = ob.n_frames;
        double* q = hd + u.off_d;
        std::memcpy(q, ob.wTo, 16 * 8); q += 16;
        std::memcpy(q, ob.shape, 3 * 8); q += 3;
        std::memcpy(q, ob.kps, (size_t)3 * K * 8); q += 3 * K;
        std::memcpy(q, ob.frame_wTc, (size_t)16 * F * 8); q += (size_t)16 * F;
        std::memcpy(q, ob.frame_zs, (size_t)2 * K * F * 8); q += (size_t)2 * K * F;
        std::memcpy(q, ob.frame_bbox, (size_t)4 * F * 8);
        int* fc = hi + u.off_i;
        int* fr0 = fc + F;
        // rows of the frames in frame order (the reference's interleaved layout); groups by clone for the compression
        int first[ORCVIO_MAX_CLONES], count[ORCVIO_MAX_CLONES];
        for (int c = 0; c < N; ++c) { first[c] = -1; count[c] = 0; }
        int rr = u.row0;
        bool contiguous = true;   // every clone's rows are one contiguous run (frames map to distinct clones)
        std::vector<int>& fnr = h->obj_fnr;
        fnr.assign(F, 0);
        int kcnt[36] = {0};   // rows of every keypoint block, [K]: border-only rows
        std::vector<unsigned long long>& vmask = h->obj_vmask;   // observed keypoints of every frame of this track
        vmask.assign(F, 0ull);
        for (int f = 0; f < F; ++f) {
            fc[f] = ob.frame_clone[f];
            fr0[f] = 0;
            if (fc[f] < 0) continue;
            int nv = 0;
            unsigned long long mk = 0ull;
            const double* zs = ob.frame_zs + (size_t)f * K * 2;
            for (int k = 0; k < K; ++k)
                if (std::isfinite(zs[2 * k]) && std::isfinite(zs[2 * k + 1])) { mk |= 1ull << k; kcnt[k] += 2; ++nv; }
            vmask[f] = mk;
            kcnt[K] += 4;
            fr0[f] = rr;
            const int c = fc[f], nr = 2 * nv + 4;
            fnr[f] = nr;
            if (first[c] < 0) first[c] = rr; else if (first[c] + count[c] != rr) contiguous = false;
            count[c] += nr;
            rr += nr;
        }
        if (contiguous) {
            for (int r = u.row0; r < rr; ++r) ridx[r] = r;
            for (int c = 0; c < N; ++c)
                if (count[c] > 0) groups[ng++] = ObjGroup{first[c], first[c] + count[c], c, (int)ui};
        } else {   // two frames of the object share a clone: group the rows through the index list
            int pos = u.row0;
            for (int c = 0; c < N; ++c) {
                if (count[c] == 0) continue;
                const int g0 = pos;
                for (int f = 0; f < F; ++f)
                    if (fc[f] == c)
                        for (int r = 0; r < fnr[f]; ++r) ridx[pos++] = fr0[f] + r;
                groups[ng++] = ObjGroup{g0, pos, c, (int)ui};
            }
        }
        rowptr[ui + 1] = rr;
        if (arrow_ok) {
            if (rr - u.row0 > 2048) arrow_ok = false;
            int start[36];
            for (int k = 0; k <= K && arrow_ok; ++k) {
                if (k < K && kcnt[k] > 128) { arrow_ok = false; break; }
                start[k] = arrow_pos;
                kp_ranges[arrow_off + k] = int2{arrow_pos, arrow_pos + kcnt[k]};
                arrow_pos += kcnt[k];
            }
            if (arrow_ok) {
                for (int f = 0; f < F; ++f) {
                    if (fc[f] < 0) continue;
                    const unsigned long long mk = vmask[f];
                    int rj = fr0[f];
                    for (int k = 0; k < K; ++k)
                        if (mk >> k & 1ull) { int& st = start[k]; kp_rows[st] = rj; kp_rows[st + 1] = rj + 1; st += 2; rj += 2; }
                    int& sb = start[K];
                    kp_rows[sb] = rj; kp_rows[sb + 1] = rj + 1; kp_rows[sb + 2] = rj + 2; kp_rows[sb + 3] = rj + 3;
                    sb += 4;
                }
                arrows[ui] = ObjArrow{u.row0, rr - u.row0, K, arrow_off};
                arrow_off += K + 1;
                if (K > arrow_Kmax) arrow_Kmax = K;
                if (rr - u.row0 > arrow_rows_max) arrow_rows_max = rr - u.row0;
            }
        }
        ObjEvalArgs a;
        a.wTo = dd + u.off_d; a.shape = a.wTo + 16; a.kps = a.shape + 3; a.frame_wTc = a.kps + 3 * K;
        a.frame_zs = a.frame_wTc + (size_t)16 * F; a.frame_bbox = a.frame_zs + (size_t)2 * K * F;
        a.frame_clone = di + u.off_i; a.frame_row0 = a.frame_clone + F;
        a.K = K; a.F = F; a.ncol = u.ncol; a.ldhf = pl.ldf; a.rcol = pl.no_max; a.row_cols = pl.d_cols;
        a.obj_left = fl->use_left_perturbation; a.new_bbox = fl->use_new_bbox_residual; a.vio_left = fl->vio_use_left_perturbation;
        a.fix_D = fl->fix_dcampose_dimupose_to_identity;
        std::memcpy(a.R_b2c, fl->R_b2c, sizeof(a.R_b2c));
        std::memcpy(a.t_c_b, fl->t_c_b, sizeof(a.t_c_b));
        a.Hx6 = pl.d_hx; a.Hf = pl.d_hf; a.res = pl.d_res; a.row_clone = pl.d_clone;
        std::memcpy(hd + args_off + arg_dbl * ui, &a, sizeof(a));
    }
    const int n_eval = pl.nobj;   // (the row kernel is launched per evaluated track whatever happens to the blocks afterwards)
    bool stacked = false;
    if (h->ref_stack_hf && pl.nobj > 1) {
        merge_objects_ref_stack(N, pl.nobj, pl.rows_tot, ridx, rowptr, groups, &ng);
        pl.nobj = 1; h->obj_count = 1; stacked = true;
        rc = objects_scratch(h, &pl);   // the scratch of ONE block: Cd | Sg | Hr adjacent again (the row arrays depend on the row count only)
        if (rc != ORCVIO_OK) return rc;
    }
    pl.ngroups = ng;
    pl.arrow = !stacked && want_arrow && arrow_ok;
    if (pl.arrow) { pl.Kmax = arrow_Kmax; pl.rows_max = arrow_rows_max; }
    if (pl.arrow) {   // room for the arrow factors (the row arrays the kernel arguments point at do not move: same sizes)
        double* keep_objH = h->d_objH; int* keep_obj_i = h->d_obj_i;
        rc = objects_scratch(h, &pl);
        if (rc != ORCVIO_OK) return rc;
        if (h->d_objH != keep_objH || h->d_obj_i != keep_obj_i) { g_last_error = "objects_local_tracks: scratch moved"; return ORCVIO_ERR_HIP; }
    }
    pl.d_ridx = di + o_ridx; pl.d_rowptr = di + o_rowptr; pl.d_groups = reinterpret_cast<ObjGroup*>(di + o_groups);
    pl.d_arrow = reinterpret_cast<ObjArrow*>(di + o_arrow); pl.d_kp_range = reinterpret_cast<int2*>(di + o_range); pl.d_kp_rows = di + o_kprows;
    const auto tt3 = std::chrono::steady_clock::now();
    { const int ri = launch_ingest(h, s, h->h_obj_stage_dev, h->d_obj_in, nd * 8 + (o_groups + (size_t)4 * ng) * 4, obj_ingest_kernel()); if (ri != ORCVIO_OK) return ri; }
    (void)n_eval;
    // the rows, and in the same launch the zeroing of what the compression accumulates into (Cd, Sg, Hr: adjacent) and of the
    // two pivot counters.  After a merge for ORCVIO_OPT_REF_STACK_HF the scratch layout is another one: plain fills there.
    const bool fold_zero = !stacked;
    const size_t nzero = (size_t)pl.nobj * pl.NOP * NAP + (size_t)pl.nobj * N * 64 + (size_t)pl.nobj * N * pl.NOP;
    const unsigned zero_rows = fold_zero ? (unsigned)((nzero / 2 + (size_t)Fmax * 64 * 8 - 1) / ((size_t)Fmax * 64 * 8)) : 0u;   // ~8 double2 per thread
    hipLaunchKernelGGL(k_object_rows_batch, dim3(Fmax, (unsigned)use.size() + zero_rows), dim3(64), 0, s,
                       reinterpret_cast<const ObjEvalArgs*>(dd + args_off), (int)use.size(), pl.d_Cd, nzero, h->d_info + 4);
    HIPCHK(hipGetLastError());
    rc = objects_pipeline(h, s, dst, pl, fold_zero, true);   // (no synchronisation: everything staged lives in the handle's pinned arena)
    if (timing) {
        const auto tt4 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "objects_local_tracks: sizes %.1f us, prior (sync + P staging + copy) %.1f us, staging %.1f us, enqueue %.1f us\n", us(tt0, tt1),
                us(tt1, tt2), us(tt2, tt3), us(tt3, tt4));
    }
    return rc;
}

// Second part: rank-ordered sum of the gathered blocks, replicated solve, joint chi-square gate with the TOTAL degrees
// of freedom of all ranks' objects, gated write-back.
static int objects_finish_impl(orcvio_msckf_handle* h, const double* d_blocks, int n_blocks, size_t stride, const double* meta0, int dof_total,
                               hipStream_t s) {
    h->last_stream = s;
    if (h->dl_pending) { HIPCHK(hipStreamSynchronize(h->dl_stream)); h->dl_pending = false; }
    h->obj_dof = dof_total;
    h->A_deferred = false;
    int rc = ORCVIO_OK;
    if (!(n_blocks == 1 && d_blocks == h->d_A)) {   // (one-shot single-GPU calls compress into d_A: nothing to sum)
        rc = launch_reduce(h, s, d_blocks, n_blocks, h->d_A, stride, meta0);
        if (rc != ORCVIO_OK) return rc;
        prof_mark(h, s, "k_gram_reduce");
    } else {
        HIPCHK(hipMemsetAsync(h->d_info + 9, 0, sizeof(int) * 4, s));   // (no shard status words in this update)
    }
    // Kalman solve in square-root form, gate, gated write-back
    if (h->prior_forked) HIPCHK(hipStreamWaitEvent(s, h->ev_side, 0));
    prof_mark(h, s, "join chol(P) (side stream)");
    for (int st = ST_FORM_U; st <= ST_TRSM && rc == ORCVIO_OK; ++st) rc = launch_solve_stage(h, s, st);
    if (rc != ORCVIO_OK) return rc;
    prof_mark(h, s, "k_gemm(U)+k_gemm(M)+k_potrf_solve(M)");
    // table value below 500 dof, on the fly above (:1962-1968); dof 0 (no usable object anywhere) can never pass.  The gate is
    // decided inside k_finish_sqrt (ObjGate).
    h->obj_thr = dof_total > 0 ? orcvio_msckf_chi2_quantile(dof_total, h->flags.chi2_prob) : -1.0;
    rc = launch_solve_stage(h, s, ST_FINISH);
    prof_mark(h, s, "k_finish_sqrt (gate inside)");
    if (rc == ORCVIO_OK) { h->ran = true; h->last_update_objects = true; h->last_run_kind = 2; h->last_sharded = meta0 != nullptr; }
    return rc;
}

// Second part: rank-ordered sum of the gathered blocks, replicated solve, joint chi-square gate with the TOTAL degrees
// of freedom of all ranks' objects, gated write-back.
int32_t orcvio_msckf_objects_finish(orcvio_msckf_handle* h, const double* d_blocks, int32_t n_blocks, int32_t dof_total, void* stream) {
    if (!h || !h->uploaded || !h->objects_mode || !d_blocks || n_blocks < 1) { g_last_error = "objects_finish: no local object block"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    return objects_finish_impl(h, d_blocks, n_blocks, 0, nullptr, dof_total, pick_stream(h, stream));
}

// Results of an object update (after orcvio_msckf_objects_finish): accept[0], gamma[0], dx, P_out, stats, optional G.
int32_t orcvio_msckf_objects_download(orcvio_msckf_handle* h, orcvio_msckf_result* res) {
    if (!h || !res || !h->ran || !h->objects_mode) { g_last_error = "objects_download: no finished object update"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    const int n = h->n, NA = h->NA, dof = h->obj_dof, nobj = h->obj_count;
    const orcvio_msckf_flags* flags = &h->flags;
    int rc = ORCVIO_OK;
    // results: ONE copy of the outputs arena [info | dx | gamma | accept | (P+)] into its pinned mirror (enqueued behind the
    // update by the one-shot entry points; here for staged callers), one synchronisation
    const bool want_P = res->P_out != nullptr;
    if (h->pub_pending) {   // the one-shot calls: k_epilogue is pushing the results into the pinned block; wait on its flag
        h->pub_pending = false;
        const int rw = io_wait(h, h->last_stream ? h->last_stream : h->stream);
        if (rw != ORCVIO_OK) { h->ran = false; return rw; }
    } else {
        if (!(h->dl_pending && (h->dl_with_P || !want_P))) {
            hipStream_t sd = h->last_stream ? h->last_stream : h->stream;
            if (h->dl_pending) HIPCHK(hipStreamSynchronize(h->dl_stream));
            const int rq = download_enqueue(h, sd, want_P);
            if (rq != ORCVIO_OK) return rq;
        }
        HIPCHK(hipStreamSynchronize(h->dl_stream));
        h->dl_pending = false;
    }
    const char* so = h->h_stage + h->in_cap;
    const double* dx = reinterpret_cast<const double*>(so + h->oo_dx);
    const int acc = *reinterpret_cast<const int*>(so + h->oo_accept);
    double gam = *reinterpret_cast<const double*>(so + h->oo_gamma);
    int info[16] = {0};
    std::memcpy(info, so, sizeof(int) * 16);
    if (info[8] != 0) {   // a solver wavefront of k_potrf_solve gave up waiting for the factorisation workgroup: the solve ran
                          // through on stale bytes and nothing of this update may be committed
        HIPCHK(hipMemset(h->d_info + 8, 0, sizeof(int)));
        h->ran = false;
        g_last_error = "k_potrf_solve: an in-launch hand-off timed out";
        return ORCVIO_ERR_TIMEOUT;
    }
    if (h->last_sharded && info[9] != 0) {   // sharded update: another rank took part with an empty share and a status word
        h->ran = false;
        g_last_error = "sharded object update: rank " + std::to_string(info[9] - 1) + " could not take part with its tracks (status " + std::to_string(info[10]) + ")";
        return ORCVIO_ERR_PEER;
    }
    if (info[2] != 0 || info[3] != 0) {   // as orcvio_msckf_download: the device left P and x alone, nothing to commit
        h->ran = false;
        g_last_error = "M = s2 I + L^T A L is not positive definite (a prior beyond ~1e16 s2 in scale, or non-finite input): no update";
        return ORCVIO_ERR_NOT_SPD;
    }
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(dx[i])) {
            h->ran = false;   // (cov_commit would make a non-finite P+ the resident covariance)
            g_last_error = "non-finite result (NaN / Inf in the prior, the rows or the noise): no update";
            return ORCVIO_ERR_NOT_SPD;
        }
    if (res->dx) std::memcpy(res->dx, dx, sizeof(double) * n);
    if (res->P_out) std::memcpy(res->P_out, so + h->oo_Pout, sizeof(double) * (size_t)n * n);
    if (dof == 0) gam = NAN;   // no usable object on any rank (the reference returns before the gate, :2157)
    if (res->accept) res->accept[0] = acc;
    if (res->gamma) res->gamma[0] = gam;
    std::memset(res->stats, 0, sizeof(res->stats));
    res->stats[0] = acc ? dof : 0;
    res->stats[1] = acc ? NA : 0;
    res->stats[2] = acc ? nobj : 0;
    res->stats[3] = acc;
    if (flags->discard_large_update) {
        const double nv = std::sqrt(dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
        const double np = std::sqrt(dx[6] * dx[6] + dx[7] * dx[7] + dx[8] * dx[8]);
        res->stats[4] = (nv > 1.0 || np > 1.5) ? 1 : 0;
    }
    res->stats[5] = info[0];
    res->stats[6] = info[1];
    res->stats[7] = info[4];   // rank-deficient directions met in some Hf
    if (res->G) {   // basis-independent K*H of the applied update (zero if rejected)
        if (acc) {
            rc = compute_optional(h, false, false, true);
            if (rc != ORCVIO_OK) return rc;
            std::vector<double> Gd((size_t)n * h->NP);
            HIPCHK(hipMemcpy(Gd.data(), h->d_KG, sizeof(double) * Gd.size(), hipMemcpyDeviceToHost));
            std::memset(res->G, 0, sizeof(double) * (size_t)n * n);
            for (int i = 0; i < n; ++i)
                for (int c = 0; c < NA; ++c) res->G[(size_t)i * n + 15 + c] = Gd[(size_t)i * h->NP + c];
        } else {
            std::memset(res->G, 0, sizeof(double) * (size_t)n * n);
        }
    }
    return ORCVIO_OK;
}

// ORCVIO_OPT_OBJECT_DOF = 1: the gate's degrees of freedom are rows - rank(H_f) instead of the reference's rows - columns
// (src/orcvio.cpp:2172): the device projects onto the WHOLE left null space of a rank-deficient H_f (rows - rank directions,
// DESIGN.md 3.4), and this makes the threshold count what gamma sums.  The rank is what the structured QR found (dropped
// pivots, info[4]): one small copy and a synchronisation between the compression and the solve, in this mode only.
static int objects_rank_dof(orcvio_msckf_handle* h, hipStream_t s, int32_t* dof) {
    if (!h->obj_dof_rank || h->obj_count == 0) return ORCVIO_OK;
    int dropped = 0;
    HIPCHK(hipMemcpyAsync(&dropped, h->d_info + 4, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *dof += dropped;
    return ORCVIO_OK;
}

int32_t orcvio_msckf_update_objects(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, int32_t n_clones,
                                    const orcvio_msckf_object_rows* objs, int32_t n_objects, const double* P,
                                    orcvio_msckf_result* res) {
    if (!res) { g_last_error = "update_objects: null argument"; return ORCVIO_ERR_INVALID; }
    int32_t dof = 0;
    int rc = orcvio_msckf_objects_local(h, flags, n_clones, objs, n_objects, P, h->d_A, &dof, nullptr);
    if (rc == ORCVIO_OK) rc = objects_rank_dof(h, h->stream, &dof);
    if (rc != ORCVIO_OK) return rc;
    rc = orcvio_msckf_objects_finish(h, h->d_A, 1, dof, nullptr);
    if (rc == ORCVIO_OK) rc = obj_publish_kernel() ? publish_enqueue(h, h->stream, res->P_out != nullptr)   // results -> pinned block, then the flag
                                                   : download_enqueue(h, h->stream, res->P_out != nullptr);
    if (rc != ORCVIO_OK) return rc;
    rc = orcvio_msckf_objects_download(h, res);
    h->objects_mode = false;
    return rc;
}

int32_t orcvio_msckf_update_object_tracks(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, const orcvio_object_eval_flags* fl,
                                          int32_t n_clones, const orcvio_object_track* tracks, int32_t n_tracks, const double* P,
                                          orcvio_msckf_result* res) {
    if (!res) { g_last_error = "update_object_tracks: null argument"; return ORCVIO_ERR_INVALID; }
    int32_t dof = 0;
    static const bool timing = getenv("ORCVIO_TIMING") != nullptr;   // diagnostics: host wall time of the three parts
    const auto t0 = std::chrono::steady_clock::now();
    int rc = orcvio_msckf_objects_local_tracks(h, flags, fl, n_clones, tracks, n_tracks, P, h->d_A, &dof, nullptr);
    if (rc == ORCVIO_OK) rc = objects_rank_dof(h, h->stream, &dof);
    if (rc != ORCVIO_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    rc = orcvio_msckf_objects_finish(h, h->d_A, 1, dof, nullptr);
    if (rc == ORCVIO_OK) rc = obj_publish_kernel() ? publish_enqueue(h, h->stream, res->P_out != nullptr)   // results -> pinned block, then the flag
                                                   : download_enqueue(h, h->stream, res->P_out != nullptr);
    if (rc != ORCVIO_OK) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    rc = orcvio_msckf_objects_download(h, res);
    const auto t3 = std::chrono::steady_clock::now();
    if (timing) {
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "update_object_tracks: local %.1f us, finish (enqueue) %.1f us, download %.1f us\n", us(t0, t1), us(t1, t2), us(t2, t3));
    }
    h->objects_mode = false;
    return rc;
}


// ---- the object update from ObjectLM messages ----------------------------------------------------
// Sophus v1.0.0 SE3d::exp, tangent (upsilon, omega): R = exp(omega), t = V upsilon
static void se3_exp_wire(const double xi[6], double T[16]) {
    const double* u = xi;
    const double* w = xi + 3;
    double R[9];
    so3_exp_decl(w, R);
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
    const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
    double a, b;
    if (th < 1e-10) { a = 0.5; b = 1.0 / 6.0; }
    else { a = (1.0 - std::cos(th)) / th2; b = (th - std::sin(th)) / (th2 * th); }
    double V[9];
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * W[i] + b * W2[i];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = V[3 * i] * u[0] + V[3 * i + 1] * u[1] + V[3 * i + 2] * u[2];
    }
    T[12] = T[13] = T[14] = 0.0; T[15] = 1.0;
}

int32_t orcvio_msckf_update_object_lm_msgs(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, int32_t n_clones,
                                           const double* cur_window_timestamps, const double* R_b2c, const double* t_c_b,
                                           int32_t fix_D, int32_t wire_row_major, const orcvio_object_lm_msg* msgs, int32_t n_msgs,
                                           const double* P, orcvio_msckf_result* result) {
    if (!h || !flags || !cur_window_timestamps || !R_b2c || !t_c_b || n_msgs < 0 || (n_msgs > 0 && !msgs) || !result) {
        g_last_error = "update_object_lm_msgs: null argument"; return ORCVIO_ERR_INVALID;
    }
    // constructObjectResidualJacobians per message (host arithmetic on a few hundred rows), into compact row blocks
    struct Block { std::vector<int32_t> clone; std::vector<double> hx6, hf, res; int ncol = 0; };
    std::vector<Block> blocks;
    std::vector<orcvio_msckf_object_rows> rows;
    blocks.reserve(n_msgs);
    for (int q = 0; q < n_msgs; ++q) {
        const orcvio_object_lm_msg& m = msgs[q];
        if (m.n_rows < 0 || m.n_obj_cols < 1 || m.n_frames < 0 || (m.n_rows > 0 && (!m.residual || !m.jacobian_wrt_object_state || !m.jacobian_wrt_sensor_state)) ||
            (m.n_frames > 0 && (!m.valid_camera_pose_mat || !m.timestamps || !m.zs_num_wrt_timestamps))) {
            g_last_error = "update_object_lm_msgs: malformed message"; return ORCVIO_ERR_INVALID;
        }
        if (m.n_obj_cols > 112) { g_last_error = "update_object_lm_msgs: object state columns must be 1..112"; return ORCVIO_ERR_INVALID; }
        long sum_zs_l = 0;
        for (int f = 0; f < m.n_frames; ++f) {
            // counts arrive from the wire: a negative one would lower the sum and let a row range start outside the arrays
            if (m.zs_num_wrt_timestamps[f] < 0 || m.zs_num_wrt_timestamps[f] > (1 << 20)) { g_last_error = "update_object_lm_msgs: negative keypoint count"; return ORCVIO_ERR_INVALID; }
            sum_zs_l += 2L * m.zs_num_wrt_timestamps[f];
        }
        if (sum_zs_l + 4L * m.n_frames > (long)m.n_rows) { g_last_error = "update_object_lm_msgs: fewer rows than 2 x keypoints + 4 x frames"; return ORCVIO_ERR_INVALID; }
        const int sum_zs = (int)sum_zs_l;
        const int nr = m.n_rows, nc = m.n_obj_cols, nf = m.n_frames;
        // element (i, j) of a rows x cols wire matrix
        auto at = [&](const double* d, int rws, int cls, int i, int j) { return wire_row_major ? d[(size_t)i * cls + j] : d[(size_t)j * rws + i]; };
        Block b;
        b.ncol = nc;
        int src = 0;
        for (int f = 0; f < nf; ++f) {
            const int zf = 2 * m.zs_num_wrt_timestamps[f];
            int idx = -1;
            for (int c = 0; c < n_clones; ++c)
                if (cur_window_timestamps[c] == m.timestamps[f]) { idx = c; break; }   // exact match (std::find on doubles, :2073)
            if (idx >= 0) {
                double D[36];
                std::memset(D, 0, sizeof(D));
                if (fix_D) { for (int i = 0; i < 6; ++i) D[6 * i + i] = 1.0; }
                else {   // :2079-2093
                    double xi[6], wTc[16];
                    for (int i = 0; i < 6; ++i) xi[i] = at(m.valid_camera_pose_mat, 6, nf, i, f);
                    se3_exp_wire(xi, wTc);
                    double v[3], tbw[3];
                    for (int i = 0; i < 3; ++i) v[i] = -(R_b2c[3 * i] * t_c_b[0] + R_b2c[3 * i + 1] * t_c_b[1] + R_b2c[3 * i + 2] * t_c_b[2]);
                    for (int i = 0; i < 3; ++i) tbw[i] = wTc[4 * i] * v[0] + wTc[4 * i + 1] * v[1] + wTc[4 * i + 2] * v[2] + wTc[4 * i + 3];
                    if (flags->use_left_perturbation) {   // se3_ops.hpp:531-552, rows (upsilon, omega), cols (theta, p)
                        const double S[9] = {0, -tbw[2], tbw[1], tbw[2], 0, -tbw[0], -tbw[1], tbw[0], 0};
                        for (int i = 0; i < 3; ++i) {
                            for (int j = 0; j < 3; ++j) D[6 * i + j] = S[3 * i + j];
                            D[6 * (3 + i) + i] = 1.0;
                            D[6 * i + 3 + i] = 1.0;
                        }
                    } else {
                        const double S[9] = {0, -t_c_b[2], t_c_b[1], t_c_b[2], 0, -t_c_b[0], -t_c_b[1], t_c_b[0], 0};
                        for (int i = 0; i < 3; ++i)
                            for (int j = 0; j < 3; ++j) {
                                double sm = 0;
                                for (int k = 0; k < 3; ++k) sm += R_b2c[3 * i + k] * S[3 * k + j];
                                D[6 * i + j] = -sm;
                                D[6 * (3 + i) + j] = R_b2c[3 * i + j];
                                D[6 * i + 3 + j] = wTc[4 * j + i];   // R_w2c = R_c2w^T
                            }
                    }
                }
                auto emit = [&](int r) {
                    for (int c = 0; c < 6; ++c) {
                        double sm = 0;
                        for (int k = 0; k < 6; ++k) sm += at(m.jacobian_wrt_sensor_state, nr, 6, r, k) * D[6 * k + c];
                        b.hx6.push_back(sm);
                    }
                    for (int c = 0; c < nc; ++c) b.hf.push_back(at(m.jacobian_wrt_object_state, nr, nc, r, c));
                    b.res.push_back(m.residual[r]);
                    b.clone.push_back(idx);
                };
                for (int r = src; r < src + zf; ++r) emit(r);
                for (int r = sum_zs + 4 * f; r < sum_zs + 4 * f + 4; ++r) emit(r);
            }
            src += zf;
        }
        if (b.clone.empty()) continue;   // no pose of the object in the window (:2149): the message contributes nothing
        blocks.push_back(std::move(b));
    }
    for (const Block& b : blocks)
        rows.push_back(orcvio_msckf_object_rows{(int32_t)b.clone.size(), b.ncol, b.clone.data(), b.hx6.data(), b.hf.data(), b.res.data()});
    return orcvio_msckf_update_objects(h, flags, n_clones, rows.data(), (int32_t)rows.size(), P, result);
}

// ---- object residual rows ------------------------------------------------------
int32_t orcvio_msckf_object_rows_eval(orcvio_msckf_handle* h, const orcvio_object_eval_flags* fl, const orcvio_object_track* ob,
                                      int32_t cap_rows, int32_t* n_rows, int32_t* row_clone, double* Hx6, double* Hf, double* res) {
    if (!h || !fl || !ob || !n_rows || !ob->wTo || !ob->shape || !ob->kps || !ob->frame_wTc || !ob->frame_zs || !ob->frame_bbox ||
        !ob->frame_clone) { g_last_error = "object_rows_eval: null argument"; return ORCVIO_ERR_INVALID; }
    const int K = ob->n_keypoints, F = ob->n_frames;
    if (K < 1 || K > 60 || F < 1) { g_last_error = "object_rows_eval: 1..60 keypoints (one wavefront per frame), >= 1 frame"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    const int ncol = 9 + 3 * K;
    // row offsets of the in-window frames (valid keypoint rows first, then 4 bbox rows)
    std::vector<int> row0(F, 0);
    int rows = 0;
    for (int f = 0; f < F; ++f) {
        if (ob->frame_clone[f] < 0) continue;
        int nv = 0;
        for (int k = 0; k < K; ++k) {
            const double a = ob->frame_zs[((size_t)f * K + k) * 2], b = ob->frame_zs[((size_t)f * K + k) * 2 + 1];
            if (std::isfinite(a) && std::isfinite(b)) ++nv;   // row finite test, ObjectLM.cpp:171-198
        }
        row0[f] = rows;
        rows += 2 * nv + 4;
    }
    *n_rows = rows;
    if (rows == 0) return ORCVIO_OK;
    if (rows > cap_rows || !row_clone || !Hx6 || !Hf || !res) { g_last_error = "object_rows_eval: output buffers too small"; return ORCVIO_ERR_CAPACITY; }
    // pack inputs / outputs into the (growable) object scratch buffers
    const size_t in_d = 16 + 3 + (size_t)3 * K + (size_t)F * (16 + 2 * K + 4);
    const size_t out_d = (size_t)rows * (6 + ncol + 1);
    int rc;
    if ((rc = grow(&h->d_objH, &h->cap_objH, in_d + out_d)) != ORCVIO_OK) return rc;
    if ((rc = grow(&h->d_obj_i, &h->cap_obj_i, (size_t)2 * F + rows)) != ORCVIO_OK) return rc;
    std::vector<double> hin(in_d);
    double* q = hin.data();
    std::memcpy(q, ob->wTo, 16 * 8); q += 16;
    std::memcpy(q, ob->shape, 3 * 8); q += 3;
    std::memcpy(q, ob->kps, (size_t)3 * K * 8); q += 3 * K;
    std::memcpy(q, ob->frame_wTc, (size_t)16 * F * 8); q += (size_t)16 * F;
    std::memcpy(q, ob->frame_zs, (size_t)2 * K * F * 8); q += (size_t)2 * K * F;
    std::memcpy(q, ob->frame_bbox, (size_t)4 * F * 8);
    std::vector<int> hi(2 * F);
    for (int f = 0; f < F; ++f) { hi[f] = ob->frame_clone[f]; hi[F + f] = row0[f]; }
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->d_objH, hin.data(), sizeof(double) * in_d, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->d_obj_i, hi.data(), sizeof(int) * 2 * F, hipMemcpyHostToDevice, s));
    ObjEvalArgs a;
    a.wTo = h->d_objH; a.shape = a.wTo + 16; a.kps = a.shape + 3; a.frame_wTc = a.kps + 3 * K;
    a.frame_zs = a.frame_wTc + (size_t)16 * F; a.frame_bbox = a.frame_zs + (size_t)2 * K * F;
    a.frame_clone = h->d_obj_i; a.frame_row0 = h->d_obj_i + F;
    a.K = K; a.F = F; a.ncol = ncol; a.ldhf = ncol; a.rcol = -1; a.row_cols = nullptr;
    a.obj_left = fl->use_left_perturbation; a.new_bbox = fl->use_new_bbox_residual; a.vio_left = fl->vio_use_left_perturbation;
    a.fix_D = fl->fix_dcampose_dimupose_to_identity;
    std::memcpy(a.R_b2c, fl->R_b2c, sizeof(a.R_b2c));
    std::memcpy(a.t_c_b, fl->t_c_b, sizeof(a.t_c_b));
    double* outd = h->d_objH + in_d;
    a.Hx6 = outd; a.Hf = outd + (size_t)rows * 6; a.res = a.Hf + (size_t)rows * ncol;
    a.row_clone = h->d_obj_i + 2 * F;
    hipLaunchKernelGGL(k_object_rows, dim3(F), dim3(64), 0, s, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(Hx6, a.Hx6, sizeof(double) * (size_t)rows * 6, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(Hf, a.Hf, sizeof(double) * (size_t)rows * ncol, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(res, a.res, sizeof(double) * rows, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(row_clone, a.row_clone, sizeof(int) * rows, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ORCVIO_OK;
}

// ---- multi-GPU: RCCL communicator owned by the handle ---------------------------------------
// RCCL is loaded with dlopen on first use, so the library has no link-time dependency on it and a single-GPU caller
// never loads it.  In a process that already holds librccl.so.1 (PyTorch ships one) the same instance is reused.
// Every wait that another rank can strand is BOUNDED (ORCVIO_COMM_TIMEOUT_S, default 180 s): the creation of the
// communicator runs on a helper thread the caller stops waiting for, the streams that carry a collective are polled; a
// time-out aborts the communicator and returns ORCVIO_ERR_TIMEOUT -- never a hang.
namespace {
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;
double comm_timeout_seconds() {
    static const double t = [] { const char* e = getenv("ORCVIO_COMM_TIMEOUT_S"); const double v = e ? atof(e) : 180.0; return v > 0.0 ? v : 180.0; }();
    return t;
}
}  // namespace

static int rccl_load() {
    if (g_rccl.lib) return ORCVIO_OK;
    const char* env = getenv("ORCVIO_RCCL_LIB");
    const char* cands[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    std::string tried;
    for (const char* c : cands) {
        if (!c || !*c) continue;
        lib = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
        tried += std::string(c) + " ";
    }
    if (!lib) { g_last_error = "RCCL not found (tried " + tried + "; set ORCVIO_RCCL_LIB)"; return ORCVIO_ERR_NO_DEVICE; }
    RcclApi a;
    a.lib = lib;
#define RCCL_SYM(name) a.name = reinterpret_cast<decltype(a.name)>(dlsym(lib, "nccl" #name)); \
    if (!a.name) { g_last_error = "RCCL: symbol nccl" #name " missing"; dlclose(lib); return ORCVIO_ERR_NO_DEVICE; }
    RCCL_SYM(GetUniqueId) RCCL_SYM(CommInitRank) RCCL_SYM(CommDestroy) RCCL_SYM(CommAbort) RCCL_SYM(AllGather) RCCL_SYM(AllReduce)
    RCCL_SYM(GroupStart) RCCL_SYM(GroupEnd) RCCL_SYM(GetErrorString)
#undef RCCL_SYM
    g_rccl = a;
    return ORCVIO_OK;
}
#define RCCLCHK(expr)                                                                                       \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess) {                                                                            \
            g_last_error = std::string(#expr) + ": " + g_rccl.GetErrorString(_r);                           \
            return ORCVIO_ERR_HIP;                                                                          \
        }                                                                                                   \
    } while (0)

// A blocking RCCL call on a helper thread, waited for with a bound.  If the caller gives up, the thread is left to finish (or
// to sit) on its own and cleans up what it produced; the shared state outlives both.
namespace {
struct BoundedCall {
    std::mutex m;
    std::condition_variable cv;
    bool done = false, abandoned = false;
    ncclResult_t result = ncclSuccess;
    ncclComm_t comm = nullptr;
    ncclUniqueId id;
};
}  // namespace

int32_t orcvio_msckf_comm_unique_id(uint8_t* id) {
    if (!id) { g_last_error = "comm_unique_id: null"; return ORCVIO_ERR_INVALID; }
    { const int rl = rccl_load(); if (rl != ORCVIO_OK) return rl; }
    static_assert(sizeof(ncclUniqueId) == ORCVIO_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    auto st = std::make_shared<BoundedCall>();
    std::thread([st] {
        ncclUniqueId u;
        const ncclResult_t r = g_rccl.GetUniqueId(&u);
        std::lock_guard<std::mutex> lk(st->m);
        st->id = u; st->result = r; st->done = true;
        st->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(st->m);
    if (!st->cv.wait_for(lk, std::chrono::duration<double>(comm_timeout_seconds()), [&] { return st->done; })) {
        st->abandoned = true;
        g_last_error = "comm_unique_id: ncclGetUniqueId did not return within ORCVIO_COMM_TIMEOUT_S";
        return ORCVIO_ERR_TIMEOUT;
    }
    if (st->result != ncclSuccess) { g_last_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(st->result); return ORCVIO_ERR_HIP; }
    std::memcpy(id, &st->id, sizeof(ncclUniqueId));
    return ORCVIO_OK;
}

// a rank that never arrives leaves the others in a collective for ever: give the communicator up instead
static void comm_abort(orcvio_msckf_handle* h) {
    if (h->comm) { (void)g_rccl.CommAbort(h->comm); h->comm = nullptr; }
    h->comm_world = 0; h->comm_rank = 0;
    h->graph_epoch++;
}

// Wait for a stream that carries a collective: polled, bounded.  Spins for the first two milliseconds (an update takes a
// fraction of one), then naps between polls.
static int comm_stream_wait(orcvio_msckf_handle* h, hipStream_t s, const char* who) {
    if (!h->comm) { HIPCHK(hipStreamSynchronize(s)); return ORCVIO_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    const double limit = comm_timeout_seconds();
    for (unsigned it = 0;; ++it) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return ORCVIO_OK;
        if (e != hipErrorNotReady) { g_last_error = std::string(who) + ": " + hipGetErrorString(e); return ORCVIO_ERR_HIP; }
        if ((it & 63u) == 63u) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > limit) {
                comm_abort(h);
                g_last_error = std::string(who) + ": a rank did not arrive at the collective within ORCVIO_COMM_TIMEOUT_S; the communicator has been aborted";
                return ORCVIO_ERR_TIMEOUT;
            }
            if (el > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    }
}

int32_t orcvio_msckf_comm_destroy(orcvio_msckf_handle* h) {
    if (!h) return ORCVIO_ERR_INVALID;
    if (h->comm) {
        (void)hipSetDevice(h->device);
        (void)comm_stream_wait(h, h->stream, "comm_destroy");
        if (h->comm && h->comm_stream) (void)comm_stream_wait(h, h->comm_stream, "comm_destroy");
        if (h->comm) (void)g_rccl.CommDestroy(h->comm);
        h->comm = nullptr;
    }
    if (h->d_gather) { (void)hipFree(h->d_gather); h->d_gather = nullptr; }
    if (h->d_dofs) { (void)hipFree(h->d_dofs); h->d_dofs = nullptr; }
    if (h->h_dofs) { (void)hipHostFree(h->h_dofs); h->h_dofs = nullptr; }
    if (h->comm_stream) { (void)hipStreamDestroy(h->comm_stream); h->comm_stream = nullptr; }
    h->comm_world = 0; h->comm_rank = 0;
    h->graph_epoch++;
    return ORCVIO_OK;
}

int32_t orcvio_msckf_comm_init(orcvio_msckf_handle* h, const uint8_t* id, int32_t rank, int32_t world) {
    if (!h || !id || world < 1 || rank < 0 || rank >= world) { g_last_error = "comm_init: invalid rank / world"; return ORCVIO_ERR_INVALID; }
    { const int rl = rccl_load(); if (rl != ORCVIO_OK) return rl; }
    if (h->comm || h->d_gather) (void)orcvio_msckf_comm_destroy(h);
    HIPCHK(hipSetDevice(h->device));
    auto st = std::make_shared<BoundedCall>();
    std::memcpy(&st->id, id, sizeof(ncclUniqueId));
    const int device = h->device;
    std::thread([st, device, world, rank] {
        (void)hipSetDevice(device);
        ncclComm_t c = nullptr;
        const ncclResult_t r = g_rccl.CommInitRank(&c, world, st->id, rank);
        std::unique_lock<std::mutex> lk(st->m);
        st->result = r; st->comm = c; st->done = true;
        const bool orphan = st->abandoned;
        st->cv.notify_all();
        lk.unlock();
        if (orphan && r == ncclSuccess && c) (void)g_rccl.CommAbort(c);   // nobody is waiting for it any more
    }).detach();
    {
        std::unique_lock<std::mutex> lk(st->m);
        if (!st->cv.wait_for(lk, std::chrono::duration<double>(comm_timeout_seconds()), [&] { return st->done; })) {
            st->abandoned = true;
            g_last_error = "comm_init: ncclCommInitRank did not return within ORCVIO_COMM_TIMEOUT_S (a rank missing, or the bootstrap stuck)";
            return ORCVIO_ERR_TIMEOUT;
        }
        if (st->result != ncclSuccess) { g_last_error = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(st->result); return ORCVIO_ERR_HIP; }
        h->comm = st->comm;
    }
    h->comm_rank = rank; h->comm_world = world;
    const size_t slot = (size_t)h->NAP_max * h->NAP_max + ORCVIO_SHARD_META;
    HIPCHK(hipMalloc(&h->d_gather, sizeof(double) * slot * world));
    HIPCHK(hipMalloc(&h->d_dofs, sizeof(double) * 2 * (world + 8)));
    HIPCHK(hipHostMalloc(&h->h_dofs, sizeof(double) * 2 * (world + 8), hipHostMallocDefault));
    HIPCHK(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    h->graph_epoch++;
    return ORCVIO_OK;
}

int32_t orcvio_msckf_comm_info(orcvio_msckf_handle* h, int32_t* rank, int32_t* world) {
    if (!h) return ORCVIO_ERR_INVALID;
    if (rank) *rank = h->comm_rank;
    if (world) *world = h->comm_world;
    return ORCVIO_OK;
}

// max over the ranks of count <= 8 doubles (the bench contract's MAX over ranks; a caller's own consistency checks), through the
// handle's communicator on the handle's stream: everything enqueued there before is finished on every rank when it returns, so
// with count = 0 this is the barrier.
int32_t orcvio_msckf_comm_allreduce_max(orcvio_msckf_handle* h, double* values, int32_t count) {
    if (!h || count < 0 || count > 8 || (count > 0 && !values)) { g_last_error = "comm_allreduce_max: 0..8 values"; return ORCVIO_ERR_INVALID; }
    if (!h->comm) { g_last_error = "comm_allreduce_max: no communicator (orcvio_msckf_comm_init)"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    double* hb = h->h_dofs + 2 * h->comm_world;       // pinned scratch behind the dofs: [8]
    double* db = h->d_dofs + 2 * h->comm_world;
    for (int i = 0; i < 8; ++i) hb[i] = i < count ? values[i] : 0.0;
    HIPCHK(hipMemcpyAsync(db, hb, sizeof(double) * 8, hipMemcpyHostToDevice, s));
    RCCLCHK(g_rccl.AllReduce(db, db, 8, ncclDouble, ncclMax, h->comm, s));
    HIPCHK(hipMemcpyAsync(hb, db, sizeof(double) * 8, hipMemcpyDeviceToHost, s));
    { const int rw = comm_stream_wait(h, s, "comm_allreduce_max"); if (rw != ORCVIO_OK) return rw; }
    for (int i = 0; i < count; ++i) values[i] = hb[i];
    return ORCVIO_OK;
}
int32_t orcvio_msckf_comm_barrier(orcvio_msckf_handle* h) { return orcvio_msckf_comm_allreduce_max(h, nullptr, 0); }

// this rank's tracks -> its block, written straight into its slot of the gather buffer, a few status words behind it -> in-place
// all-gather (the one data-path collective; <= 295 KB per rank, latency-bound over xGMI) -> rank-ordered sum + replicated solve
static inline size_t shard_slot(const orcvio_msckf_handle* h) { return (size_t)h->NAP * h->NAP + ORCVIO_SHARD_META; }

int32_t orcvio_msckf_run_update_sharded(orcvio_msckf_handle* h, void* stream) {
    if (!h || !h->uploaded || h->pw_missing) { g_last_error = "run_update_sharded: nothing uploaded (or positions missing)"; return ORCVIO_ERR_INVALID; }
    if (!h->comm) { g_last_error = "run_update_sharded: no communicator (orcvio_msckf_comm_init)"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = pick_stream(h, stream);
    const size_t ne = (size_t)h->NAP * h->NAP, slot = shard_slot(h);
    double* mine = h->d_gather + slot * h->comm_rank;
    int rc = run_local_impl(h, s, mine);
    if (rc != ORCVIO_OK) return rc;
    hipLaunchKernelGGL(k_shard_meta, dim3(1), dim3(64), 0, s, mine + ne, h->shard_status, 0, h->F > 0 ? (const int*)h->d_accept : (const int*)nullptr,
                       (const int*)h->d_row_ptr, h->F);
    HIPCHK(hipGetLastError());
    RCCLCHK(g_rccl.AllGather(mine, h->d_gather, slot, ncclDouble, h->comm, s));
    rc = run_finish_impl(h, h->d_gather, h->comm_world, slot, h->d_gather + ne, s);
    h->last_sharded = rc == ORCVIO_OK;
    return rc;
}

// a status that only THIS rank's share can have caused (the window and the prior are replicated, so every other refusal hits
// all ranks alike, before the collective)
static inline bool per_rank_status(int rc) { return rc == ORCVIO_ERR_CAPACITY || rc == ORCVIO_ERR_TRACK_TOO_LONG || rc == ORCVIO_ERR_INVALID; }

int32_t orcvio_msckf_update_features_sharded(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, const orcvio_msckf_window* window,
                                             const orcvio_msckf_tracks* tracks, const double* P, orcvio_msckf_result* result) {
    if (!h || !result) { g_last_error = "update_features_sharded: null argument"; return ORCVIO_ERR_INVALID; }
    if (!h->comm) { g_last_error = "update_features_sharded: no communicator (orcvio_msckf_comm_init)"; return ORCVIO_ERR_INVALID; }
    int own = ORCVIO_OK;
    std::string own_error;
    int rc = upload_to_arena(h, flags, window, tracks, P, "orcvio_msckf_update_features_sharded");
    if (rc != ORCVIO_OK && per_rank_status(rc)) {
        // This rank's tracks were refused.  The other ranks are on their way into the all-gather: take part with an EMPTY share
        // and a status word, so that nobody waits for ever and every rank learns of it (ORCVIO_ERR_PEER).
        own = rc; own_error = g_last_error;
        const int32_t zero = 0;
        orcvio_msckf_tracks none{};
        none.n_features = 0; none.obs_ptr = &zero;
        rc = upload_to_arena(h, flags, window, &none, P, "orcvio_msckf_update_features_sharded");
        if (rc != ORCVIO_OK) { g_last_error = own_error; return own; }   // the window itself is unusable: every rank returns here alike
    }
    if (rc != ORCVIO_OK) return rc;
    HIPCHK(hipMemcpyAsync(h->d_in, h->h_stage, upload_bytes(h), hipMemcpyHostToDevice, h->stream));
    h->shard_status = own;
    rc = orcvio_msckf_run_update_sharded(h, nullptr);
    h->shard_status = 0;
    if (rc != ORCVIO_OK) return rc;
    rc = download_enqueue(h, h->stream, result->P_out != nullptr);
    if (rc != ORCVIO_OK) return rc;
    { const int rw = comm_stream_wait(h, h->stream, "update_features_sharded"); if (rw != ORCVIO_OK) { h->ran = false; h->dl_pending = false; return rw; } }
    rc = orcvio_msckf_download(h, result);
    if (own != ORCVIO_OK) { g_last_error = own_error; return own; }
    return rc;
}

int32_t orcvio_msckf_update_object_tracks_sharded(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags,
                                                  const orcvio_object_eval_flags* eval_flags, int32_t n_clones,
                                                  const orcvio_object_track* tracks, int32_t n_tracks, const double* P,
                                                  orcvio_msckf_result* res) {
    if (!h || !res) { g_last_error = "update_object_tracks_sharded: null argument"; return ORCVIO_ERR_INVALID; }
    if (!h->comm) { g_last_error = "update_object_tracks_sharded: no communicator (orcvio_msckf_comm_init)"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int world = h->comm_world, rank = h->comm_rank;
    // NAP is a function of the window only: known before the local part runs
    const int n = flags ? flags->leg_dim + 6 * n_clones + h->n_extra : 0;
    const int NA = h->ekf_mode ? n - 15 : n - h->n_extra - 15;
    const size_t ne = (size_t)round_up(NA + 1, 16) * round_up(NA + 1, 16), slot = ne + ORCVIO_SHARD_META;
    if (flags && (NA < 1 || round_up(NA + 1, 16) > h->NAP_max)) { g_last_error = "update_object_tracks_sharded: window exceeds capacity"; return ORCVIO_ERR_CAPACITY; }
    double* mine = h->d_gather + slot * rank;
    int32_t dof = 0;
    int own = ORCVIO_OK;
    std::string own_error;
    int rc = orcvio_msckf_objects_local_tracks(h, flags, eval_flags, n_clones, tracks, n_tracks, P, mine, &dof, nullptr);
    if (rc != ORCVIO_OK && per_rank_status(rc) && flags && eval_flags) {   // this rank's tracks were refused: an empty share + a status word
        own = rc; own_error = g_last_error;
        rc = orcvio_msckf_objects_local_tracks(h, flags, eval_flags, n_clones, nullptr, 0, P, mine, &dof, nullptr);
        if (rc != ORCVIO_OK) { g_last_error = own_error; return own; }   // the window / prior is unusable: every rank returns here alike
        dof = 0;
    }
    if (rc == ORCVIO_OK) rc = objects_rank_dof(h, s, &dof);
    if (rc != ORCVIO_OK) return rc;
    hipLaunchKernelGGL(k_shard_meta, dim3(1), dim3(64), 0, s, mine + ne, own, (int)dof, (const int*)nullptr, (const int*)nullptr, 0);
    HIPCHK(hipGetLastError());
    // The gate's threshold is the chi-square quantile of the TOTAL degrees of freedom, host arithmetic above 500 like the
    // reference's (src/orcvio.cpp:1962-1968), and every rank knows its own share of them before its kernels have run: the
    // dofs travel FIRST, on a stream of their own, and the host reads them while the device works on this rank's rows -- no
    // synchronisation in the middle of the update.
    h->h_dofs[world + rank] = (double)dof;
    HIPCHK(hipMemcpyAsync(h->d_dofs + rank, h->h_dofs + world + rank, sizeof(double), hipMemcpyHostToDevice, h->comm_stream));
    RCCLCHK(g_rccl.AllGather(h->d_dofs + rank, h->d_dofs, 1, ncclDouble, h->comm, h->comm_stream));
    HIPCHK(hipMemcpyAsync(h->h_dofs, h->d_dofs, sizeof(double) * world, hipMemcpyDeviceToHost, h->comm_stream));
    RCCLCHK(g_rccl.AllGather(mine, h->d_gather, slot, ncclDouble, h->comm, s));
    { const int rw = comm_stream_wait(h, h->comm_stream, "update_object_tracks_sharded"); if (rw != ORCVIO_OK) return rw; }
    int dof_total = 0;
    for (int r = 0; r < world; ++r) dof_total += (int)h->h_dofs[r];
    rc = objects_finish_impl(h, h->d_gather, world, slot, h->d_gather + ne, dof_total, s);
    if (rc != ORCVIO_OK) return rc;
    rc = download_enqueue(h, s, res->P_out != nullptr);
    if (rc != ORCVIO_OK) return rc;
    { const int rw = comm_stream_wait(h, s, "update_object_tracks_sharded"); if (rw != ORCVIO_OK) { h->ran = false; h->dl_pending = false; return rw; } }
    rc = orcvio_msckf_objects_download(h, res);
    h->objects_mode = false;
    if (own != ORCVIO_OK) { g_last_error = own_error; return own; }
    return rc;
}

int32_t orcvio_msckf_profile_stages(orcvio_msckf_handle* h, const char** names, double* ms, int32_t* count) {
    if (!h || !names || !ms || !count) { g_last_error = "profile_stages: null argument"; return ORCVIO_ERR_INVALID; }
    const int ns = h->prof_n > 0 ? h->prof_n - 1 : 0;
    if (*count < ns) { g_last_error = "profile_stages: output arrays too small"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    if (ns > 0) HIPCHK(hipEventSynchronize(h->prof_ev[ns]));
    for (int i = 0; i < ns; ++i) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, h->prof_ev[i], h->prof_ev[i + 1]));
        ms[i] = t;
        names[i] = h->prof_names[i];
    }
    *count = ns;
    return ORCVIO_OK;
}

// ---- per-kernel profile -------------------------------------------------------------------------
int32_t orcvio_msckf_profile_update(orcvio_msckf_handle* h, void* stream, int32_t reps, const char** names, double* ms,
                                    int32_t* count) {
    if (!h || !h->uploaded || !names || !ms || !count || reps < 1) { g_last_error = "profile_update: invalid"; return ORCVIO_ERR_INVALID; }
    static const char* kn[] = {"k_feature", "k_gram", "k_assemble", "k_potrf(P)", "k_gemm(U)", "k_gemm(M)", "k_potrf(M)", "k_trsm", "k_finish"};
    const int nk = 9;
    if (*count < nk) { g_last_error = "profile_update: need room for 9 entries"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = pick_stream(h, stream);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const bool front = front_fused_active(h);   // k_front = tracks + compression + chol(P) in one launch: reported as entry 0
    const bool defer = front_defers_assembly(h);
    for (int k = 0; k < nk; ++k) ms[k] = 0.0;
    std::vector<float> samples[9];   // per kernel: the MEDIAN over the repetitions is reported (one preempted launch is not the kernel's time)
    for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < nk; ++k) {
            if (front && k >= 1 && k <= 3) continue;
            HIPCHK(hipEventRecord(e0, s));
            int rc = ORCVIO_OK;
            h->A_deferred = defer;
            if (k == 0) rc = front ? launch_front(h, s, h->d_A, defer) : launch_feature(h, s);
            else if (k == 1) rc = launch_gram(h, s);
            else if (k == 2) rc = launch_assemble(h, s, h->d_A);
            else rc = launch_solve_stage(h, s, k - 3);
            if (rc != ORCVIO_OK) return rc;
            HIPCHK(hipEventRecord(e1, s));
            HIPCHK(hipEventSynchronize(e1));
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, e0, e1));
            samples[k].push_back(t);
        }
    }
    for (int k = 0; k < nk; ++k)
        if (!samples[k].empty()) {
            std::sort(samples[k].begin(), samples[k].end());
            const size_t m = samples[k].size();
            ms[k] = (m & 1) ? samples[k][m / 2] : 0.5 * (samples[k][m / 2 - 1] + samples[k][m / 2]);
        }
    int no = 0;
    for (int k = 0; k < nk; ++k) {
        if (k == 3 + ST_TRSM && fused_solve_active(h)) continue;   // nothing launched: part of k_potrf_solve(M)
        if (front && k >= 1 && k <= 3) continue;                   // part of k_front
        ms[no] = ms[k];
        names[no] = (k == 3 + ST_POTRF_M && fused_solve_active(h)) ? "k_potrf_solve(M)" : ((front && k == 0) ? "k_front" : kn[k]);
        ++no;
    }
    *count = no;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(hipMemsetAsync(h->d_info, 0, sizeof(int) * 8, s));
    HIPCHK(hipStreamSynchronize(s));
    h->ran = true;
    return ORCVIO_OK;
}

// ---- device-resident covariance --------------------------------------------------
int32_t orcvio_msckf_cov_set(orcvio_msckf_handle* h, int32_t n, const double* P) {
    if (!h || !P || n < 1 || n > h->n_max) { g_last_error = "cov_set: invalid"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(h->d_Pres, P, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice));
    h->res_n = n;
    h->fac_valid = false;   // a new covariance: its factor is not known
    return ORCVIO_OK;
}

int32_t orcvio_msckf_cov_get(orcvio_msckf_handle* h, int32_t* n_out, double* P_out) {
    if (!h || h->res_n == 0) { g_last_error = "cov_get: no resident covariance"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    if (n_out) *n_out = h->res_n;
    if (P_out) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(P_out, h->d_Pres, sizeof(double) * (size_t)h->res_n * h->res_n, hipMemcpyDeviceToHost));
    }
    return ORCVIO_OK;
}

int32_t orcvio_msckf_cov_propagate(orcvio_msckf_handle* h, int32_t leg, const double* Phi, const double* Q) {
    if (!h || !Phi || !Q || (leg != 22 && leg != 46) || h->res_n < leg) { g_last_error = "cov_propagate: invalid"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int n = h->res_n;
    double* dPhi = h->d_covT + (size_t)46 * h->n_max;   // behind the Phi P rows: Phi, then Q
    double* dQ = dPhi + 46 * 46;
    HIPCHK(hipMemcpyAsync(dPhi, Phi, sizeof(double) * leg * leg, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dQ, Q, sizeof(double) * leg * leg, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_cov_propagate_rows, dim3((leg * n + 255) / 256), dim3(256), 0, s, h->d_Pres, n, dPhi, leg, h->d_covT);
    hipLaunchKernelGGL(k_cov_propagate_finish, dim3((n * n + 255) / 256), dim3(256), 0, s, h->d_Pres, n, dPhi, dQ, leg, h->d_covT, h->d_Ptmp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));   // Phi and Q are caller memory
    std::swap(h->d_Pres, h->d_Ptmp);
    h->fac_valid = false;   // P_LL <- Phi P_LL Phi^T + Q: the factor of the sum is not a row operation on S
    return ORCVIO_OK;
}

int32_t orcvio_msckf_cov_augment(orcvio_msckf_handle* h) {
    if (!h || h->res_n < 9 || h->res_n + 6 > h->n_max) { g_last_error = "cov_augment: no resident covariance, or window full"; return ORCVIO_ERR_CAPACITY; }
    HIPCHK(hipSetDevice(h->device));
    const int n = h->res_n, m = n + 6;
    // the new clone goes BEHIND the clones and IN FRONT of the feature / nuisance states (rest_rows, src/orcvio.cpp:976-1003)
    if (h->n_extra > n - 15) { g_last_error = "cov_augment: more extra states than the resident covariance has"; return ORCVIO_ERR_INVALID; }
    hipLaunchKernelGGL(k_cov_augment, dim3((m * m + 255) / 256), dim3(256), 0, h->stream, h->d_Pres, n, n - h->n_extra, h->d_Ptmp);
    HIPCHK(hipGetLastError());
    std::swap(h->d_Pres, h->d_Ptmp);
    h->res_n = m;
    if (h->fac_valid && h->fac_n == n) {   // the new clone's rows of S are copies of the IMU's (theta, p) rows
        const int ldo = round_up(m + 1, 16);
        hipLaunchKernelGGL(k_fac_augment, dim3((h->fac_k * m + 255) / 256), dim3(256), 0, h->stream, h->d_Sres, h->fac_ld, h->fac_k, n,
                           n - h->n_extra, h->d_Stmp, ldo);
        HIPCHK(hipGetLastError());
        std::swap(h->d_Sres, h->d_Stmp);
        h->fac_n = m; h->fac_ld = ldo;
        h->fac_tail = 0;   // (the new clone's rows are copies of IMU rows: not zero in the trailing columns)
    } else h->fac_valid = false;
    return ORCVIO_OK;
}

int32_t orcvio_msckf_cov_remove_clones(orcvio_msckf_handle* h, int32_t leg, const int32_t* idx, int32_t count) {
    if (!h || (count > 0 && !idx) || count < 0 || (leg != 22 && leg != 46) || h->res_n < leg) { g_last_error = "cov_remove_clones: invalid"; return ORCVIO_ERR_INVALID; }
    if (count == 0) return ORCVIO_OK;
    HIPCHK(hipSetDevice(h->device));
    const int n = h->res_n, N = (n - leg) / 6;
    std::vector<char> drop(n, 0);
    for (int k = 0; k < count; ++k) {
        if (idx[k] < 0 || idx[k] >= N) { g_last_error = "cov_remove_clones: index out of the window"; return ORCVIO_ERR_INVALID; }
        for (int c = 0; c < 6; ++c) drop[leg + 6 * idx[k] + c] = 1;
    }
    std::vector<int> map;
    for (int i = 0; i < n; ++i)
        if (!drop[i]) map.push_back(i);
    const int m = (int)map.size();
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->d_covmap, map.data(), sizeof(int) * m, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_cov_remove, dim3((m * m + 255) / 256), dim3(256), 0, s, h->d_Pres, n, h->d_covmap, m, h->d_Ptmp);
    HIPCHK(hipGetLastError());
    if (h->fac_valid && h->fac_n == n) {   // deleting states deletes rows of S
        const int ldo = round_up(m + 1, 16);
        hipLaunchKernelGGL(k_fac_remove, dim3((h->fac_k * m + 255) / 256), dim3(256), 0, s, h->d_Sres, h->fac_ld, h->fac_k, h->d_covmap, m,
                           h->d_Stmp, ldo);
        HIPCHK(hipGetLastError());
        std::swap(h->d_Sres, h->d_Stmp);
        h->fac_n = m; h->fac_ld = ldo;
    } else h->fac_valid = false;
    HIPCHK(hipStreamSynchronize(s));   // map is a local
    std::swap(h->d_Pres, h->d_Ptmp);
    h->res_n = m;
    return ORCVIO_OK;
}

// Schmidt branch of pruneImuStateBuffer (src/orcvio.cpp:2881-2920): the listed clones leave the window but STAY in the covariance
// as nuisance states -- their 6 x 6 blocks and cross terms move to the end, one clone after the other in the listed order.
// A symmetric permutation: rows of the resident square-root factor move with it.
int32_t orcvio_msckf_cov_clones_to_nuisance(orcvio_msckf_handle* h, int32_t leg, const int32_t* idx, int32_t count) {
    if (!h || (count > 0 && !idx) || count < 0 || (leg != 22 && leg != 46) || h->res_n < leg) { g_last_error = "cov_clones_to_nuisance: invalid"; return ORCVIO_ERR_INVALID; }
    if (count == 0) return ORCVIO_OK;
    HIPCHK(hipSetDevice(h->device));
    const int n = h->res_n;
    std::vector<int> map(n);
    for (int i = 0; i < n; ++i) map[i] = i;
    std::vector<int> moved_before;   // window ranks are those BEFORE any of the listed clones has moved (ascending, as rm_imu_state_ids)
    for (int k = 0; k < count; ++k) {
        int shift = 0;
        for (int q : moved_before) if (q < idx[k]) ++shift;
        const int start = leg + 6 * (idx[k] - shift);
        if (idx[k] < 0 || start + 6 > n) { g_last_error = "cov_clones_to_nuisance: index out of the window"; return ORCVIO_ERR_INVALID; }
        std::rotate(map.begin() + start, map.begin() + start + 6, map.end());   // the block goes to the end, everything behind it moves up
        moved_before.push_back(idx[k]);
    }
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->d_covmap, map.data(), sizeof(int) * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_cov_remove, dim3((n * n + 255) / 256), dim3(256), 0, s, h->d_Pres, n, h->d_covmap, n, h->d_Ptmp);
    HIPCHK(hipGetLastError());
    if (h->fac_valid && h->fac_n == n) {
        hipLaunchKernelGGL(k_fac_remove, dim3((h->fac_k * n + 255) / 256), dim3(256), 0, s, h->d_Sres, h->fac_ld, h->fac_k, h->d_covmap, n,
                           h->d_Stmp, h->fac_ld);
        HIPCHK(hipGetLastError());
        std::swap(h->d_Sres, h->d_Stmp);
    } else h->fac_valid = false;
    HIPCHK(hipStreamSynchronize(s));   // map is a local
    std::swap(h->d_Pres, h->d_Ptmp);
    return ORCVIO_OK;
}

int32_t orcvio_msckf_cov_commit(orcvio_msckf_handle* h) {
    if (!h || !h->ran) { g_last_error = "cov_commit: no finished update"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->last_stream ? h->last_stream : h->stream;
    const int n = h->n, kf = h->kf;
    if (h->factor_opt && h->n_nui > 0) h->fac_valid = false;   // Schmidt: the nuisance block of P+ is the prior's, so P+ != s2 Z^T Z
    if (h->factor_opt && h->n_nui == 0) {   // S+ = sigma Z^T (or the prior's own factor if a gated object update was rejected): read before Pres changes
        const PriorFactor pf = prior_factor(h);
        hipLaunchKernelGGL(k_fac_commit, dim3((kf * n + 255) / 256), dim3(256), 0, s, h->d_Z, h->ldz, kf, n, h->flags.noise_feature,
                           h->last_update_objects ? h->d_obj_accept : (const int*)nullptr, pf.base, pf.sLi, pf.sLj, h->d_Stmp, h->ldz,
                           (const int*)(h->d_info + 2));
        HIPCHK(hipGetLastError());
        std::swap(h->d_Sres, h->d_Stmp);
        h->fac_n = n; h->fac_k = kf; h->fac_ld = h->ldz; h->fac_valid = true; h->fac_tail = h->tail;
    }
    HIPCHK(hipMemcpyAsync(h->d_Pres, h->d_Pout, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, s));
    if (s != h->stream) HIPCHK(hipStreamSynchronize(s));   // the other cov_* calls run on the handle's own stream
    h->res_n = n;
    return ORCVIO_OK;
}

// The tail of measurementUpdate_hybrid on the device (src/orcvio.cpp:1818-1821, :1904-1947): after an update that carried entering
// features (orcvio_msckf_upload_new_features), the resident covariance becomes the AUGMENTED one -- P+ of the update with the d k
// new feature states behind it (in front of the nuisance block under ORCVIO_OPT_SCHMIDT_STATES, :1920-1935) -- from the blocks
// H_1, H_2, r_1 that are still on the device; dx_new [d k] comes back.  Replaces cov_commit + cov_get + augment_state + cov_set for a
// caller that keeps the covariance in HBM.  The caller raises ORCVIO_OPT_EXTRA_STATES by d k for its next upload.
int32_t orcvio_msckf_cov_commit_new_features(orcvio_msckf_handle* h, double* dx_new) {
    if (!h || !h->ran || h->new_F <= 0 || !dx_new) { g_last_error = "cov_commit_new_features: no finished update with entering features"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    const int n = h->n, k = h->new_F, d = h->new_idp, sz = d * k, nt = n + sz, tail = 6 * h->n_nui;
    if (nt > h->n_max) { g_last_error = "cov_commit_new_features: the augmented state exceeds the handle's capacity"; return ORCVIO_ERR_CAPACITY; }
    hipStream_t s = h->last_stream ? h->last_stream : h->stream;
    // scratch behind the blocks: HH [sz][n + 1] | W [k][d][d] | nHHP [sz][n] | Q [sz][sz] | dx_new [sz] | flag
    const size_t need = ((size_t)sz * (n + 1) + (size_t)k * d * d + (size_t)sz * n + (size_t)sz * sz + sz + 2) * sizeof(double);
    if (need > h->aug_cap) {
        HIPCHK(hipDeviceSynchronize());
        if (h->d_aug) (void)hipFree(h->d_aug);
        HIPCHK(hipMalloc(&h->d_aug, need * 2));
        h->aug_cap = need * 2;
    }
    double* HH = reinterpret_cast<double*>(h->d_aug);
    double* W = HH + (size_t)sz * (n + 1);
    double* nHHP = W + (size_t)k * d * d;
    double* Q = nHHP + (size_t)sz * n;
    double* dxn = Q + (size_t)sz * sz;
    int* flag = reinterpret_cast<int*>(dxn + sz);
    const double* dout = reinterpret_cast<const double*>(h->d_new + h->new_out_off);
    const double* H1 = dout; const double* H2 = dout + (size_t)sz * n; const double* r1 = H2 + (size_t)k * d * d;
    const double s2 = h->flags.noise_feature * h->flags.noise_feature;
    HIPCHK(hipMemsetAsync(flag, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_aug_hh, dim3((n + 1 + 255) / 256, k), dim3(256), 0, s, H1, H2, r1, n, k, d, HH, W, flag, h->ref_h2_ldlt ? 1 : 0);
    hipLaunchKernelGGL(k_aug_dx, dim3(sz), dim3(64), 0, s, (const double*)HH, n, (const double*)h->d_dx, dxn);
    int rc = launch_gemm(s, HH, (long)(n + 1), 1L, h->d_Pout, (long)n, 1L, sz, n, n, -1.0, 0.0, 0, nHHP, (long)n, 1L);                 // nHHP = -HH P+
    if (rc == ORCVIO_OK) rc = launch_gemm(s, nHHP, (long)n, 1L, HH, 1L, (long)(n + 1), sz, sz, n, 1.0, 0.0, 0, Q, (long)sz, 1L);       // Q = nHHP HH^T
    if (rc != ORCVIO_OK) return rc;
    hipLaunchKernelGGL(k_aug_assemble, dim3((nt * nt + 255) / 256), dim3(256), 0, s, (const double*)h->d_Pout, n, sz, tail, d, (const double*)nHHP,
                       (const double*)Q, (const double*)W, s2, h->d_Ptmp);
    HIPCHK(hipGetLastError());
    int bad = 0;
    HIPCHK(hipMemcpyAsync(dx_new, dxn, sizeof(double) * sz, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (bad) { g_last_error = "cov_commit_new_features: singular H_2"; return ORCVIO_ERR_NOT_SPD; }
    std::swap(h->d_Pres, h->d_Ptmp);
    h->res_n = nt;
    h->fac_valid = false;   // (the factor of the augmented covariance would have d k more columns: not kept)
    h->new_F = 0;
    return ORCVIO_OK;
}

// Factor the resident covariance NOW (asynchronously, on the handle's stream): P = L L^T, L kept as the resident square-root
// factor.  processModel adds Q to the IMU block, after which no factor of P is known; a caller that propagates and augments when
// the image arrives and updates when the front end has finished tracking it (milliseconds later) takes the Cholesky of the
// prior -- the one part of the first update of a frame that does not depend on the tracks -- off the update's critical path.
int32_t orcvio_msckf_cov_prefactor(orcvio_msckf_handle* h) {
    if (!h || h->res_n == 0) { g_last_error = "cov_prefactor: no resident covariance"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    const int n = h->res_n, nb = (n + 15) / 16;
    if (!h->factor_opt || (h->fac_valid && h->fac_n == n)) return ORCVIO_OK;   // switched off, or the factor is known already
    if (nb > 14 || nb * 16 > h->NP_max) return ORCVIO_OK;   // no register-resident factorisation of this size: the update factors P itself
    const int ld = round_up(n + 1, 16);
    hipStream_t s = h->stream;
    // L(i, j) = R[j * ld + i] (k_potrf_reg writes the upper factor R, P = R^T R, full 16 x 16 tiles, zeros below the diagonal): the
    // layout of the resident factor S (S(i, j) = d_Sres[i + j * fac_ld]).  With the reversed factorisation (rev_prior_opt) the
    // factor comes out with its rows in reverse order: it is written to scratch and flipped into place, and its last 15 columns
    // are zero in the active rows (fac_tail)
    const double eps = 2.220446049250313e-16;
    const int need = potrf_slots_needed(nb);
    const bool rev = h->rev_prior_opt && h->fused_solve && n - 15 >= 16;
    double* dst = rev ? h->d_KG : h->d_Stmp;   // (d_KG: scratch of the optional outputs, free between updates)
#define LAUNCH_PF(NS) hipLaunchKernelGGL(k_potrf_reg<NS>, dim3(1), dim3(512), 0, s, (const double*)h->d_Pres, n, n, 8.0 * eps, dst, ld, h->d_DinvP, \
                                         h->d_info, (unsigned long long*)nullptr, (size_t)0, (size_t)0, (size_t)0, 0, 0, 1, rev ? 1 : 0)
    if (need <= 4) LAUNCH_PF(4);
    else if (need <= 8) LAUNCH_PF(8);
    else if (need <= 12) LAUNCH_PF(12);
    else LAUNCH_PF(16);
#undef LAUNCH_PF
    if (rev) hipLaunchKernelGGL(k_fac_flip, dim3((n * ld + 255) / 256), dim3(256), 0, s, (const double*)dst, ld, n, h->d_Stmp, ld);
    HIPCHK(hipGetLastError());
    std::swap(h->d_Sres, h->d_Stmp);
    h->fac_n = n; h->fac_k = n; h->fac_ld = ld; h->fac_valid = true; h->fac_tail = rev ? 15 : 0;
    return ORCVIO_OK;
}

// ---- feature triangulation --------------------------------------------------
void orcvio_msckf_triangulation_config_default(orcvio_triangulation_config* c) {
    if (!c) return;
    c->translation_threshold = 0.2; c->huber_epsilon = 0.01; c->estimation_precision = 5e-7; c->initial_damping = 1e-3;
    c->outer_loop_max_iteration = 10; c->inner_loop_max_iteration = 10; c->cost_threshold = 4.7673e-04;
    c->init_final_dist_threshold = 5.0;
}

static int launch_triangulate(orcvio_msckf_handle* h, const orcvio_triangulation_config* cfg, bool have_init, bool mark_skip, hipStream_t s) {
    if (h->F == 0) return ORCVIO_OK;
    TriArgs a;
    a.poses = h->d_poses; a.obs_ptr = h->d_obs_ptr; a.obs_clone = h->d_obs_clone; a.obs_z = h->d_obs_z;
    a.is_init = have_init ? h->d_tri_init : nullptr;
    a.p_w = h->d_pw; a.valid = h->d_tri_valid; a.flags = h->d_tri_flags; a.solution = h->d_tri_sol; a.cost = h->d_tri_cost;
    a.skip = mark_skip ? h->d_skip : nullptr;
    a.translation_threshold = cfg->translation_threshold; a.huber_epsilon = cfg->huber_epsilon;
    a.estimation_precision = cfg->estimation_precision; a.initial_damping = cfg->initial_damping;
    a.cost_threshold = cfg->cost_threshold; a.init_final_dist_threshold = cfg->init_final_dist_threshold;
    a.outer_max = cfg->outer_loop_max_iteration; a.inner_max = cfg->inner_loop_max_iteration;
    a.F = h->F;
    hipLaunchKernelGGL(k_triangulate, dim3(h->F), dim3(64), 0, s, a);
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

int32_t orcvio_msckf_triangulate_uploaded(orcvio_msckf_handle* h, const orcvio_triangulation_config* cfg,
                                          const int32_t* is_initialized, void* stream) {
    if (!h || !cfg || !h->uploaded || h->objects_mode) { g_last_error = "triangulate_uploaded: no uploaded tracks"; return ORCVIO_ERR_INVALID; }
    if (h->pw_missing && is_initialized) { g_last_error = "triangulate_uploaded: is_initialized needs uploaded positions"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = pick_stream(h, stream);
    if (is_initialized && h->F > 0) {
        HIPCHK(hipMemcpyAsync(h->d_tri_init, is_initialized, sizeof(int) * h->F, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));   // is_initialized is the caller's (pageable) memory: it may be reused as soon as this call returns
    }
    int rc = launch_triangulate(h, cfg, is_initialized != nullptr, true, s);
    if (rc != ORCVIO_OK) return rc;
    h->skip_active = true;
    h->pw_missing = false;
    return ORCVIO_OK;
}

int32_t orcvio_msckf_triangulate(orcvio_msckf_handle* h, const orcvio_triangulation_config* cfg, const orcvio_msckf_window* w,
                                 const orcvio_msckf_tracks* tr, const int32_t* is_initialized, orcvio_triangulation_result* res) {
    if (!h || !cfg || !w || !tr || !res || !w->R_b2w || !w->t_b_w || !w->R_b2c || !w->t_c_b || !tr->obs_ptr) {
        g_last_error = "orcvio_msckf_triangulate: null argument";
        return ORCVIO_ERR_INVALID;
    }
    const int N = w->n_clones, F = tr->n_features;
    if (N < 1 || F < 0) { g_last_error = "orcvio_msckf_triangulate: bad sizes"; return ORCVIO_ERR_INVALID; }
    if (N > h->maxN || F > h->maxF) { g_last_error = "orcvio_msckf_triangulate: exceeds handle capacity"; return ORCVIO_ERR_CAPACITY; }
    if (F > 0 && tr->obs_ptr[0] < 0) { g_last_error = "orcvio_msckf_triangulate: obs_ptr starts below zero"; return ORCVIO_ERR_INVALID; }
    const int nobs = F > 0 ? tr->obs_ptr[F] : 0;
    if (nobs < 0) { g_last_error = "orcvio_msckf_triangulate: obs_ptr not monotone"; return ORCVIO_ERR_INVALID; }
    if (nobs > h->maxObs) { g_last_error = "orcvio_msckf_triangulate: too many observations"; return ORCVIO_ERR_CAPACITY; }
    if (F > 0 && (!tr->obs_clone || !tr->obs_z)) { g_last_error = "orcvio_msckf_triangulate: null track arrays"; return ORCVIO_ERR_INVALID; }
    if (is_initialized && !tr->p_w) { g_last_error = "orcvio_msckf_triangulate: is_initialized needs tracks->p_w"; return ORCVIO_ERR_INVALID; }
    for (int j = 0; j < F; ++j) {
        const int M = tr->obs_ptr[j + 1] - tr->obs_ptr[j];
        if (M < 0) { g_last_error = "orcvio_msckf_triangulate: obs_ptr not monotone"; return ORCVIO_ERR_INVALID; }
        if (M > 64) { g_last_error = "orcvio_msckf_triangulate: track longer than 64 observations"; return ORCVIO_ERR_TRACK_TOO_LONG; }
    }
    for (int o = 0; o < nobs; ++o)
        if (tr->obs_clone[o] < 0 || tr->obs_clone[o] >= N) { g_last_error = "orcvio_msckf_triangulate: obs_clone out of range"; return ORCVIO_ERR_INVALID; }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    // this call owns the track buffers: whatever was uploaded for an update is gone
    h->uploaded = false; h->ran = false; h->skip_active = false; h->objects_mode = false; h->io_open = false;
    h->N = N; h->F = F; h->nobs = nobs;
    HIPCHK(hipStreamSynchronize(s));
    if (h->dl_pending) { HIPCHK(hipStreamSynchronize(h->dl_stream)); h->dl_pending = false; }
    layout_inputs(h, N, F, nobs, false, true, h->n_max);
    h->h_poses.assign((size_t)POSE_STRIDE * N, 0.0);
    const double* tfej = w->t_fej ? w->t_fej : w->t_b_w;
    for (int i = 0; i < N; ++i) {
        double* r = &h->h_poses[(size_t)POSE_STRIDE * i];
        std::memcpy(r + POSE_R_B2W, w->R_b2w + 9 * i, 9 * sizeof(double));
        std::memcpy(r + POSE_T_B_W, w->t_b_w + 3 * i, 3 * sizeof(double));
        std::memcpy(r + POSE_T_FEJ, tfej + 3 * i, 3 * sizeof(double));
        std::memcpy(r + POSE_R_B2C, w->R_b2c + 9 * i, 9 * sizeof(double));
        std::memcpy(r + POSE_T_C_B, w->t_c_b + 3 * i, 3 * sizeof(double));
    }
    HIPCHK(hipMemcpyAsync(h->d_poses, h->h_poses.data(), sizeof(double) * POSE_STRIDE * N, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->d_obs_ptr, tr->obs_ptr, sizeof(int) * (F + 1), hipMemcpyHostToDevice, s));
    if (F > 0) {
        if (tr->p_w) HIPCHK(hipMemcpyAsync(h->d_pw, tr->p_w, sizeof(double) * 3 * F, hipMemcpyHostToDevice, s));
        if (is_initialized) HIPCHK(hipMemcpyAsync(h->d_tri_init, is_initialized, sizeof(int) * F, hipMemcpyHostToDevice, s));
        if (nobs > 0) {
            HIPCHK(hipMemcpyAsync(h->d_obs_clone, tr->obs_clone, sizeof(int) * nobs, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(h->d_obs_z, tr->obs_z, sizeof(double) * 2 * nobs, hipMemcpyHostToDevice, s));
        }
    }
    int rc = launch_triangulate(h, cfg, is_initialized != nullptr, false, s);
    if (rc != ORCVIO_OK) return rc;
    if (F > 0) {
        if (res->valid) HIPCHK(hipMemcpyAsync(res->valid, h->d_tri_valid, sizeof(int) * F, hipMemcpyDeviceToHost, s));
        if (res->flags) HIPCHK(hipMemcpyAsync(res->flags, h->d_tri_flags, sizeof(int) * F, hipMemcpyDeviceToHost, s));
        if (res->p_w) HIPCHK(hipMemcpyAsync(res->p_w, h->d_pw, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, s));
        if (res->inv_param) HIPCHK(hipMemcpyAsync(res->inv_param, h->d_tri_sol, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, s));
        if (res->cost) HIPCHK(hipMemcpyAsync(res->cost, h->d_tri_cost, sizeof(double) * F, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    return ORCVIO_OK;
}

// ---- incrementState_IMUCam (src/orcvio.cpp:4468-4567): host arithmetic -------------------------
static void so3_exp(const double w[3], double R[9]);
static void so3_exp_decl(const double w[3], double R[9]) { so3_exp(w, R); }
static void so3_exp(const double w[3], double R[9]) {
    // Sophus v1.0.0 SO3d::exp: unit quaternion from the rotation vector, then to a matrix
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double th = std::sqrt(th2);
    double imag, real;
    if (th < 1e-10) {
        const double th4 = th2 * th2;
        imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
        real = 1.0 - th2 / 8.0 + th4 / 384.0;
    } else {
        imag = std::sin(0.5 * th) / th;
        real = std::cos(0.5 * th);
    }
    const double x = imag * w[0], y = imag * w[1], z = imag * w[2], q = real;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * q); R[2] = 2 * (x * z + y * q);
    R[3] = 2 * (x * y + z * q); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * q);
    R[6] = 2 * (x * z - y * q); R[7] = 2 * (y * z + x * q); R[8] = 1 - 2 * (x * x + y * y);
}
static void mat3_mul(const double* A, const double* B, double* C) {
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    std::memcpy(C, T, sizeof(T));
}

int32_t orcvio_msckf_increment_state(const orcvio_msckf_flags* f, const double* dx, orcvio_msckf_state* st) {
    if (!f || !dx || !st || !st->clone_R_b2w || !st->clone_t_b_w) { g_last_error = "increment_state: null argument"; return -1; }
    const int leg = f->leg_dim;
    const double nv = std::sqrt(dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5]);
    const double np = std::sqrt(dx[6] * dx[6] + dx[7] * dx[7] + dx[8] * dx[8]);
    if ((nv > 1.0 || np > 1.5) && f->discard_large_update) return 0;   // :4479-4494
    const bool left = f->use_larvio || f->use_left_perturbation;          // :4498, :4543
    double Rt[9];
    so3_exp(dx, Rt);
    if (left) mat3_mul(Rt, st->R_b2w_imu, st->R_b2w_imu); else mat3_mul(st->R_b2w_imu, Rt, st->R_b2w_imu);
    for (int i = 0; i < 3; ++i) {
        st->v[i] += dx[3 + i];
        st->p[i] += dx[6 + i];
        st->bg[i] += dx[9 + i];
        st->ba[i] += dx[12 + i];
    }
    {   // extrinsic: R_b2c <- R_b2c * R(smallAngleQuaternion(dtheta))^T  (:4512-4516, math_utils.hpp:104-121)
        double q[4] = {0.5 * dx[15], 0.5 * dx[16], 0.5 * dx[17], 0.0};
        const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
        if (n2 <= 1.0) q[3] = std::sqrt(1.0 - n2);
        else {
            q[3] = 1.0;
            const double s = 1.0 / std::sqrt(1.0 + n2);
            for (double& v : q) v *= s;
        }
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        // Eigen Quaterniond(w,x,y,z).toRotationMatrix() -- no normalisation, as Eigen does
        const double Rq[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                              2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                              2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
        double RqT[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) RqT[i * 3 + j] = Rq[j * 3 + i];
        mat3_mul(st->R_b2c, RqT, st->R_b2c);
        for (int i = 0; i < 3; ++i) st->t_c_b[i] += dx[18 + i];
    }
    st->td += dx[21];
    if (leg == 46) for (int i = 0; i < 24; ++i) st->imu_intrinsics[i] += dx[22 + i];   // :4522-4533
    for (int c = 0; c < st->n_clones; ++c) {
        const double* da = dx + leg + 6 * c;
        double* R = st->clone_R_b2w + 9 * c;
        double* t = st->clone_t_b_w + 3 * c;
        so3_exp(da, Rt);
        if (left) mat3_mul(Rt, R, R); else mat3_mul(R, Rt, R);
        for (int i = 0; i < 3; ++i) t[i] += da[3 + i];
        if (st->clone_R_c2w) {   // orientation_cam = R_b2w * R_b2c^T (:4555-4561)
            double RbcT[9];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) RbcT[i * 3 + j] = st->R_b2c[j * 3 + i];
            mat3_mul(R, RbcT, st->clone_R_c2w + 9 * c);
        }
        if (st->clone_t_c_w)
            for (int i = 0; i < 3; ++i)
                st->clone_t_c_w[3 * c + i] = t[i] + R[i * 3] * st->t_c_b[0] + R[i * 3 + 1] * st->t_c_b[1] + R[i * 3 + 2] * st->t_c_b[2];
    }
    return 1;
}

// ---- diagnostics: test hooks and ablation timers.  NOT part of the product ABI: compiled only into the diagnostics build
//      (orcvio_amd/lib/liborcvio_msckf_dbg.so, -DORCVIO_DEBUG_HOOKS), which the tests that need them load explicitly ----------
#ifdef ORCVIO_DEBUG_HOOKS
// ---- debug access to intermediate device buffers (tests only; not part of the public header) ---
// which: 0 Hs [m_tot x NAP], 1 Ab, 2 A (summed block), 3 RP, 4 M, 5 RM, 6 Z, 8 U, 7 dims -> int32[8]
int32_t orcvio_msckf_debug_read(orcvio_msckf_handle* h, int32_t which, void* dst, int64_t max_bytes) {
    if (!h || !dst) return ORCVIO_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    const size_t pp = (size_t)h->NAP * h->NAP * sizeof(double);
    const size_t np2 = (size_t)h->NP * h->NP * sizeof(double);
    const void* src = nullptr;
    size_t bytes = 0;
    switch (which) {
        case 0:
            if (!h->materialize) { g_last_error = "debug_read: stack not materialised (ORCVIO_OPT_MATERIALIZE_STACK)"; return ORCVIO_ERR_INVALID; }
            src = h->d_Hs; bytes = (size_t)h->m_tot * h->NAP * sizeof(double); break;
        case 1: src = h->d_Ab; bytes = pp; break;
        case 2: {
            const int ra = assemble_deferred(h, h->stream);
            if (ra != ORCVIO_OK) return ra;
            HIPCHK(hipDeviceSynchronize());
            src = h->d_A; bytes = pp; break;
        }
        case 3: src = h->d_RP; bytes = np2; break;
        case 4: src = h->d_M; bytes = np2; break;
        case 5: src = h->d_RM; bytes = np2; break;
        case 6: src = h->d_Z; bytes = (size_t)h->n * h->ldz * sizeof(double); break;
        case 8: src = h->d_U; bytes = np2; break;
        case 9: src = h->d_sync; bytes = 256; break;   // k_front: counter + diagnostic time stamps
        case 10: {   // how often an update was re-run on the forked path because the fused front end lost a hand-off
            if ((size_t)max_bytes < sizeof(int32_t)) return ORCVIO_ERR_INVALID;
            *reinterpret_cast<int32_t*>(dst) = h->front_fallbacks;
            return ORCVIO_OK;
        }
        case 7: {
            int32_t dims[8] = {h->n, h->NA, h->NAP, h->NP, h->m_tot, h->Mmax, h->ldz, h->reg_path ? 1 : 0};
            if ((size_t)max_bytes < sizeof(dims)) return ORCVIO_ERR_INVALID;
            std::memcpy(dst, dims, sizeof(dims));
            return ORCVIO_OK;
        }
        default: return ORCVIO_ERR_INVALID;
    }
    if ((size_t)max_bytes < bytes) { g_last_error = "debug_read: buffer too small"; return ORCVIO_ERR_INVALID; }
    if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return ORCVIO_OK;
}

// Test hook: hold `n_cus` compute units for `ms` milliseconds with a spinning kernel on the handle's SIDE stream (returns at once).
int32_t orcvio_msckf_debug_occupy(orcvio_msckf_handle* h, int32_t n_cus, double ms) {
    if (!h || n_cus < 1 || ms <= 0.0) return ORCVIO_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    static bool attr = false;
    if (!attr) { HIPCHK(hipFuncSetAttribute((const void*)k_debug_occupy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL(k_debug_occupy, dim3(n_cus), dim3(64), (size_t)150 * 1024, h->side, (unsigned long long)(ms * 1e5), h->d_info + 40);
    HIPCHK(hipGetLastError());
    return ORCVIO_OK;
}

// Test hook: factor an arbitrary symmetric n x n host matrix with the same kernels the update uses.
// Out: L (n x n lower, row-major, host), Dinv ([nb][16][16]), info[2] (dropped / negative pivots).
int32_t orcvio_msckf_debug_potrf(orcvio_msckf_handle* h, const double* X, int32_t n, double tol_rel, int32_t force_lds_path,
                                 double* L_out, double* Dinv_out, int32_t* info_out) {
    if (!h || !X || n < 1 || n > h->n_max) return ORCVIO_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    const int saveNP = h->NP;
    const bool save_path = h->reg_path;
    h->NP = round_up(n, 16);
    h->reg_path = !force_lds_path && (h->NP / 16) <= 14;
    const int NP = h->NP;
    HIPCHK(hipMemcpy(h->d_M, X, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(h->d_info, 0, sizeof(int) * 8));
    HIPCHK(hipMemset(h->d_RM, 0, sizeof(double) * (size_t)h->NP_max * h->NP_max));   // lower tiles: zero for any layout
    int rc = launch_potrf(h, h->stream, h->d_M, n, n, tol_rel, h->d_RM, h->d_DinvM, h->d_info);
    if (rc == ORCVIO_OK) {
        HIPCHK(hipStreamSynchronize(h->stream));
        std::vector<double> buf((size_t)NP * NP);
        HIPCHK(hipMemcpy(buf.data(), h->d_RM, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));
        long sLi, sLj;
        factor_strides(h, sLi, sLj);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) L_out[(size_t)i * n + j] = buf[(size_t)i * sLi + (size_t)j * sLj];
        if (Dinv_out) HIPCHK(hipMemcpy(Dinv_out, h->d_DinvM, sizeof(double) * 256 * ((n + 15) / 16), hipMemcpyDeviceToHost));
        if (info_out) HIPCHK(hipMemcpy(info_out, h->d_info, sizeof(int) * 2, hipMemcpyDeviceToHost));
        HIPCHK(hipMemset(h->d_info, 0, sizeof(int) * 8));
        // d_RM lower tiles may now hold data of another leading dimension: restore the invariant
        HIPCHK(hipMemset(h->d_RM, 0, sizeof(double) * (size_t)h->NP_max * h->NP_max));
    }
    h->NP = saveNP;
    h->reg_path = save_path;
    return rc;
}

// Test hook: Z = L^-1 B with the factor left in the handle by orcvio_msckf_debug_potrf_keep (same call
// with keep = 1 semantics: call debug_potrf first, then this before anything else).
int32_t orcvio_msckf_debug_trsm(orcvio_msckf_handle* h, const double* X, int32_t n, const double* B, int32_t nrhs, double* Z_out) {
    if (!h || !X || !B || n < 1 || n > h->n_max || nrhs < 1 || nrhs > h->NP_max) return ORCVIO_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    const int saveNP = h->NP, saveldz = h->ldz;
    const bool save_path = h->reg_path;
    h->NP = round_up(n, 16);
    h->reg_path = (h->NP / 16) <= 14;
    const int NP = h->NP;
    HIPCHK(hipMemcpy(h->d_M, X, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_U, B, sizeof(double) * (size_t)n * nrhs, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(h->d_RM, 0, sizeof(double) * (size_t)h->NP_max * h->NP_max));
    int rc = launch_potrf(h, h->stream, h->d_M, n, n, 0.0, h->d_RM, h->d_DinvM, h->d_info);
    if (rc == ORCVIO_OK) rc = launch_trsm(h, h->stream, h->d_RM, h->d_DinvM, n, h->d_U, nrhs, 1, nrhs, nullptr, 0, h->d_Z, h->NP_max);
    if (rc == ORCVIO_OK) {
        HIPCHK(hipStreamSynchronize(h->stream));
        std::vector<double> buf((size_t)n * h->NP_max);
        HIPCHK(hipMemcpy(buf.data(), h->d_Z, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < nrhs; ++c) Z_out[(size_t)i * nrhs + c] = buf[(size_t)i * h->NP_max + c];
        HIPCHK(hipMemset(h->d_info, 0, sizeof(int) * 8));
        HIPCHK(hipMemset(h->d_RM, 0, sizeof(double) * (size_t)h->NP_max * h->NP_max));
    }
    (void)NP;
    h->NP = saveNP; h->ldz = saveldz; h->reg_path = save_path;
    return rc;
}

// Diagnostic: time k_potrf_reg on the handle's current P with parts of the algorithm switched off
// (results are garbage for ablate != 0).  Returns the average kernel time in microseconds.
int32_t orcvio_msckf_debug_potrf_ablate(orcvio_msckf_handle* h, int32_t ablate, int32_t reps, double* us_out) {
    if (!h || !h->uploaded || !h->reg_path || !us_out) return ORCVIO_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    const int n = h->n, NP = h->NP;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL(k_potrf_reg<16>, dim3(1), dim3(512), 0, h->stream, h->d_P, n, n, 1.8e-15, h->d_RP, NP, h->d_DinvP,
                           h->d_info + 6, (unsigned long long*)nullptr, (size_t)0, (size_t)0, (size_t)0, 0, ablate);
    HIPCHK(hipEventRecord(e0, h->stream));
    for (int rep = 0; rep < reps; ++rep)
        hipLaunchKernelGGL(k_potrf_reg<16>, dim3(1), dim3(512), 0, h->stream, h->d_P, n, n, 1.8e-15, h->d_RP, NP, h->d_DinvP,
                           h->d_info + 6, (unsigned long long*)nullptr, (size_t)0, (size_t)0, (size_t)0, 0, ablate);
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *us_out = 1e3 * ms / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ORCVIO_OK;
}

// Diagnostic: core-clock stamps of one k_potrf_reg<12> run on the handle's current P (256 values; layout in potrf_reg_body)
int32_t orcvio_msckf_debug_potrf_stamps(orcvio_msckf_handle* h, unsigned long long* out256) {
    if (!h || !h->uploaded || !h->reg_path || !out256) return ORCVIO_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc(&d, sizeof(unsigned long long) * 256));
    HIPCHK(hipMemset(d, 0, sizeof(unsigned long long) * 256));
    const int n = h->n, NP = h->NP;
    const char* ab = getenv("ORCVIO_POTRF_ABLATE");   // diagnostic only: phases switched off (results are garbage)
    const int ablate = ab ? atoi(ab) : 0;
    // ORCVIO_POTRF_COLD (after a finished update): factor M instead, each time right behind the k_gemm that writes it from all
    // XCDs -- the conditions of the replayed graph (cold L2 of the factorising CU's XCD) instead of a warm repetition
    const bool cold = getenv("ORCVIO_POTRF_COLD") != nullptr && h->ran;
    for (int rep = 0; rep < 4; ++rep) {
        if (cold) {
            const int rcf = launch_solve_stage(h, h->stream, ST_FORM_M);
            if (rcf != ORCVIO_OK) return rcf;
            hipLaunchKernelGGL(k_potrf_reg<16>, dim3(1), dim3(512), 0, h->stream, h->d_M, NP, h->kf, 0.0, h->d_RM, NP, h->d_DinvM,
                               h->d_info + 6, d, (size_t)0, (size_t)0, (size_t)0, 0, ablate, 0);
        } else
            hipLaunchKernelGGL(k_potrf_reg<16>, dim3(1), dim3(512), 0, h->stream, h->d_P, n, n, 1.8e-15, h->d_RP, NP, h->d_DinvP,
                               h->d_info + 6, d, (size_t)0, (size_t)0, (size_t)0, 0, ablate);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out256, d, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return ORCVIO_OK;
}

// Diagnostic: average time of k_feature with phases switched off (outputs are garbage for ablate != 0).
int32_t orcvio_msckf_debug_feature_ablate(orcvio_msckf_handle* h, int32_t ablate, int32_t reps, double* us_out) {
    if (!h || !h->uploaded || !us_out) return ORCVIO_ERR_INVALID;
    HIPCHK(hipSetDevice(h->device));
    h->feat_ablate = ablate;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) launch_feature(h, h->stream);
    HIPCHK(hipEventRecord(e0, h->stream));
    for (int r = 0; r < reps; ++r) launch_feature(h, h->stream);
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *us_out = 1e3 * ms / reps;
    h->feat_ablate = 0;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ORCVIO_OK;
}

#endif  // ORCVIO_DEBUG_HOOKS

}  // extern "C"

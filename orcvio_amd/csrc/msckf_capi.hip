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
#include "feature_split.hpp"
#include "triangulate.hpp"
#include "cov_ops.hpp"
#include "ekf_rows.hpp"
#include "object_rows.hpp"
#include "object_fused.hpp"
#include "io_ops.hpp"
#include "frame_ops.hpp"
#include <immintrin.h>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

using namespace orcvio_amd;

static thread_local std::string g_last_error;

// Environment switches.  The PRODUCT library reads ten, all documented in include/orcvio_msckf.h ("Environment"): the communicator's
// transport and bounds, the bounds of the in-launch waits, the two numerics-relevant forms of the frame call.  Every ablation, stamp and
// measured-slower path of the lab notes is read through dbg_getenv, which the product build compiles to "unset" (VERDICT r5 #6): those
// switches exist in the diagnostics build only (liborcvio_msckf_dbg.so, -DORCVIO_DEBUG_HOOKS; tests/test_abi.py checks the strings).
#ifdef ORCVIO_DEBUG_HOOKS
static inline const char* dbg_getenv(const char* name) { return getenv(name); }
#else
static inline const char* dbg_getenv(const char*) { return nullptr; }
#endif

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

struct ObjUse {   // a usable object track of the current object update
    int t, row0, rows, ncol;
    size_t off_d, off_i;
    int fin;        // in-window frames
    int distinct;   // 1: every in-window frame on a clone of its own (the one-launch compression's condition)
    size_t off_nv;  // first entry of the track's per-frame keypoint counts in the handle's obj_nv (-1 entries: frame not in the window)
};

struct IpcComm;
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
    bool io_submitted = false, io_sub_P = false, io_sub_commit = false;   // orcvio_msckf_io_submit: launched, not yet collected
    bool io_open = false, io_with_P = false;  // orcvio_msckf_io_begin has laid the arena out and the caller is filling it
    double io_spin_seconds = 2.0;             // bound of the host's wait on h_flag (then: stream synchronisation, then ERR_TIMEOUT)
    bool last_sharded = false;                // the last finished update went through the handle's all-gather (status words in info[9..12])
    hipEvent_t* shard_ev = nullptr;           // orcvio_msckf_profile_sharded: four events the next run_update_sharded records between its parts
    int ranks_seen = 0;                       // ranks whose block was present in the last sharded update looked at (orcvio_msckf_comm_details)
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
    int obj_refine_mode = 1;            // ORCVIO_OPT_OBJECT_REFINE: 0 never, 1 objects with cond_F(R) above 3e6 (default), 2 every object
    int obj_refined = 0;                // objects of the last downloaded object update that took the explicit-basis projection (k_obj_refine)
    size_t obj_fused_budget = 0;        // dynamic LDS a workgroup of k_obj_fused may take on this device (0: not asked yet)
    int obj_fused_opt = 1;              // ORCVIO_OBJ_FUSED=0: object tracks always through the three-launch pipeline (rows materialised)
    unsigned* d_obj_done = nullptr;     // completion counter of k_gemm_objA (cumulative; never reset)
    unsigned obj_done_total = 0u;       // what that counter reads once every compression enqueued so far has finished
    const unsigned* join_wait = nullptr;   // the next ST_FORM_U product polls this counter (frame call) ...
    unsigned join_expect = 0u;          // ... until it reaches this value
    int early_rows_tot = 0, early_no_max = 0, early_dof = 0, early_Fmax = 1; size_t early_nd = 0, early_ni = 0;   // ... and what its scan of the tracks found
    bool obj_early_done = false;        // orcvio_msckf_io_update_frame has enqueued this update's compression already (objects_local_tracks skips it)
    bool obj_last_fused = false;        // the last object update from tracks took the one-launch compression
    size_t obj_lds_budget = 0;          // dynamic LDS the border launch of the object update may take on THIS handle's device (0: not asked yet)
    bool arrow_opt = true;              // ORCVIO_OPT_OBJECT_QR: structured Householder QR of Hf (0: chol(Hf^T Hf), round 1's route)
    bool use_factor = false;            // the current upload's prior comes with its factor: no Cholesky of P
    // The prior's Cholesky is taken of the REVERSED matrix (potrf_reg_body, rev): P = S S^T with S(i, c) = L'(n-1-i, c), whose
    // rows 15 .. n-1 (the states measurement rows touch) are zero in the last 15 columns -- M = s2 I + L_a^T A L_a is
    // diag(M', s2 I_15) and only M' (kf - 15) is formed and factored.  `tail` = those trailing columns of the factor in use
    // (15, or 0: LDS-panel path, unfused solve, a resident factor that has been through an augmentation); fac_tail = the same
    // for the resident factor (kept by commit / remove_clones / clones_to_nuisance, dropped by augment).
    bool rev_prior_opt = true;          // ORCVIO_REV_PRIOR=0: the plain Cholesky (tail 0), same results
    int tail = 0, fac_tail = 0;
    // orcvio_msckf_io_update_frame: the object tracks' compression runs on a stream of its own beside the feature update's solve
    hipStream_t obj_stream = nullptr;
    hipEvent_t ev_obj = nullptr;
    bool frame_mode = false;            // (objects_prior: no stream synchronisation, nothing of the prior is staged)
    size_t out_shift = 0;               // host mirror of the outputs arena: offset of the block the next download lands in / is read from
    // small host -> device transfers (SLAM feature records, Phi / Q, nuisance poses) go through a ring of pinned bounce buffers:
    // a host memcpy and an asynchronous copy each, no stream synchronisation (aux_copies)
    char* h_fetch = nullptr;            // pinned block behind fetch_copies (small device arrays -> the caller's pageable memory)
    size_t fetch_cap = 0;
    hipEvent_t ev_arena = nullptr;      // behind the last device read of the pinned input arena that was enqueued (upload_begin waits
    bool arena_busy = false;            //  for THAT, not for whatever else the stream carries: a prefactorisation, a commit ..)
    char* h_aux[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t aux_cap[4] = {0, 0, 0, 0};
    hipEvent_t ev_aux[4] = {nullptr, nullptr, nullptr, nullptr};
    bool aux_busy[4] = {false, false, false, false};
    int aux_next = 0;
    bool obj_status_cleared = false;    // k_object_rows_batch of the current object update has zeroed the shard status words (info[9..12])
    int last_run_kind = 0;              // 0: run_update (single GPU), 1: run_local / run_finish (staged or sharded), 2: objects
    bool front_retry_forked = false;    // (download: the fused front end lost a hand-off; re-running on the forked path)
    int front_spin_limit = 1 << 19;    // polls before a workgroup of k_front gives up at the device-wide counter (tens of ms)
    int front_fallbacks = 0;            // how often that happened (orcvio_msckf_counters [0])
    long long cnt_graph_captures = 0, cnt_graph_replays = 0, cnt_plain_runs = 0;   // launch sequences captured / replayed from a graph / enqueued as plain launches
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
    bool front_blocked_by_comm = false; // ipc transport, ranks sharing this device: k_front's co-residency cannot be had (capi_ipc.inc); the option above is not touched
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
    double* d_split = nullptr;          // scratch of the two-kernel front end for many tracks (feature_split.hpp): E and the gate's right-hand sides per track
    size_t split_cap = 0;
    int split_min_tracks = 0;           // ORCVIO_SPLIT_TRACKS: track count from which k_feature_e + k_feature_gate replace k_feature (0 = never: the
                                        // default since round 5 -- at 2 000 tracks the two-launch form measured 0.208 ms against 0.211 for k_feature
                                        // with E in LDS, inside the run-to-run spread, and moves 85 MB of E scratch through HBM per update for it)
    int* d_sync = nullptr;              // device-wide counter of k_front (own allocation, zero between launches)
    int* d_la_rdy = nullptr;            // k_potrf_solve_la: one word per block row a far workgroup brings forward (own 128-byte line, cleared by the k_gemm ahead of it); [16..31] / [32] the same and the step counter for chol(P) inside k_front; [48] the finish workgroups' counter (LaFin)
    int la_solve = 3;                   // look-ahead depth of the fused solve (0: k_potrf_solve, one workgroup holds the whole trailing matrix; 2 / 3: k_potrf_solve_la)
    size_t chain_fin_oo = 0;            // ... and where in the second outputs arena it put P++ (checked against the object half's layout)
    bool front_u = false;               // ORCVIO_FRONT_U=1: U = [A; b^T] L_a by the feature workgroups of k_front, behind the Grams (FrontUArgs) instead of the
                                        // k_gemm_asmA launch -- bit-identical, tested, and SLOWER (k_front 39.5 -> 54.9 us for a launch of 8.3 + gap: a
                                        // second device-wide barrier, and 144 strips of A read past the caches): a measured negative, kept opt-in
    bool front_did_U = false;           // ... the launch_front of this update did
    double chain_fin_thr = 0.0;         // ... and the threshold it used
    std::shared_ptr<void> prestage;     // ObjPrestage (capi_objects.inc): what orcvio_msckf_io_stage_object_tracks left for the next frame call
    unsigned long long cnt_prestaged = 0;   // frame calls that found their object tracks staged ahead
    bool prestage_valid = false;        // ... still stands (no other scan of object tracks since)
    int chain_fin_dof = -1;             // degrees of freedom the chained object solve's in-launch finish gated with (-1: it did not)
    int fuse_finish = 1;                // P+ / dx (and an object update's gate) by finish workgroups of k_potrf_solve_la (LaFin) instead of a k_finish_sqrt
                                        // launch behind it: 1 = in the chained frame call (both halves: 8-9 us off the frame), 2 = every update whose solve
                                        // takes the look-ahead form (measured 2 us SLOWER per queued update: the in-launch hand-off costs what the launch
                                        // boundary does, and the longer launch delays the next replay), 0 = never (ORCVIO_FUSE_FINISH)
    bool fin_frame = false;             // the chained frame call is enqueueing its feature half
    int la_spin = 1 << 22;              // polls (~1 us each) before a wait inside k_potrf_solve_la gives up: the update is then run again through k_potrf_solve
    // second solve context of the frame call's chained object solve (capi_frame.inc, ORCVIO_FRAME_CHAIN): allocated on first use
    double *d_U2 = nullptr, *d_M2 = nullptr, *d_RM2 = nullptr, *d_DinvM2 = nullptr, *d_Z2 = nullptr;
    char* d_outs2 = nullptr;
    unsigned* d_chain_words = nullptr;  // [0] feature half: M formed, [1] feature half: committed (cumulative values); [32] step counter, [48..63] block-row words of the second solve, [64] its finish workgroups' counter (LaFin)
    unsigned chain_seq = 0u;
    long long cnt_chained = 0;          // frames whose object solve ran chained (orcvio_msckf_counters [6])
    bool frame_chain = true;            // ORCVIO_FRAME_CHAIN (read at create, default 1): orcvio_msckf_io_update_frame runs the object solve chained (capi_frame.inc)
    unsigned* mark_M_word = nullptr;    // launch_solve_stage(ST_FORM_M) stores mark_M_val there from a launch of its own behind the product (once)
    unsigned mark_M_val = 0u;
    bool blk2_opt = true;               // ORCVIO_BLK2 (read at create): windows of 15 .. 26 block steps factor by 2 x 2 blocks out of the register kernels
    bool la_attr = false;               // the dynamic-LDS opt-in of k_potrf_solve_la is set for this handle's device
    // multi-GPU: RCCL communicator of this handle (orcvio_msckf_comm_init), the all-gather buffer [world][NAP_max^2] and
    // the gathered degrees of freedom of a sharded object update
    ncclComm_t comm = nullptr;
    struct IpcComm* ipc = nullptr;      // the second transport (ORCVIO_COMM_TRANSPORT=ipc, capi_ipc.inc): shared memory + HIP IPC, ranks of one node
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
    std::vector<signed char> obj_nv;    // (objects_tracks_scan) detected keypoints of every frame of every usable track: the ONE pass over the observations
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
    std::vector<unsigned char> track_is_run;   // upload_finalize: the track's clones are a contiguous run
    std::vector<int> frame_row_ptr;     // orcvio_msckf_io_update_frame: the feature half's row offsets while the object half uses h_row_ptr
    // orcvio_msckf_io_step_frame (capi_step.inc): the second update of a frame goes through an arena pair of its own, swapped in and out
    char *d_in2 = nullptr, *h_stage2 = nullptr, *h_stage2_dev = nullptr;
    bool arena_swapped = false;
    int* epi_info_keep = nullptr;       // io_enqueue hands it to k_epilogue (EpilogueArgs.info_keep); set by orcvio_msckf_io_update_frame around its feature half
    int* d_step_words = nullptr;        // [0..15] status words of the frame's first update, kept for the second update's commit (info_also)
    std::vector<int> step_row_ptr;      // the second update's row offsets
    long long cnt_step_frames = 0, cnt_step_repairs = 0;   // frames through orcvio_msckf_io_step_frame; updates of such frames run again after a lost hand-off
    bool step_ingested = false;         // k_frame_head has pulled the arena of the update being enqueued (io_enqueue skips its ingest launch)
    bool ekf_one_launch = true;         // the in-state features' evaluation, fill and gate in ONE launch (k_ekf_evalgate) instead of three (ORCVIO_EKF_ONE_LAUNCH=0, diagnostics)
    bool finpub_opt = true;             // a feature update's k_finish_sqrt + k_epilogue as ONE launch (k_finish_pub) on the in-place paths (ORCVIO_FINISH_PUB=0, diagnostics)
    FinishPubArgs* fin_pub = nullptr;   // set by io_enqueue around enqueue_update: launch_solve_stage(ST_FINISH) launches k_finish_pub with it
    bool last_finpub_commit = false;    // the update enqueued last committed by writing P+ into the spare covariance buffer: the host swaps d_Pres / d_Ptmp
    bool thin_opt = true;               // updates of at most THIN_MAX_ROWS projected rows on the in-place paths take the reference's direct form (k_thin_gain / k_thin_apply;
                                        // ORCVIO_THIN_UPDATE=0, diagnostics): pruneImuStateBuffer's update is a handful of rows
    double *d_Hthin = nullptr, *d_Vthin = nullptr, *d_uthin = nullptr;   // its projected rows [THIN_MAX_ROWS][NAP_max], V [n_max][THIN_MAX_ROWS], u
    bool thin_blocked = false;          // orcvio_msckf_update_features with K / G / H_thin requested: those are derived from the general path's factors
    bool last_update_thin = false;      // the update enqueued last took it: its commit leaves NO square-root factor
    bool step_fused = true;             // ORCVIO_STEP_FUSED=0 (diagnostics build): the frame's small steps as the separate launches and copies of the round-5 calls
    // the frame call's in-state rows beside k_front (enqueue_update): k_wait_word -> k_ekf_evalgate -> k_gram -> k_obj_done on `side`, joined
    // by two polled words (d_step_words[32]: k_front has started = the frame head is complete; [33]: the rows' Gram is complete, k_gemm_asmA_w)
    bool ekf_side_opt = true;           // ORCVIO_STEP_EKF_SIDE=0 (diagnostics build): the rows on the update's own stream, in front of k_front
    bool ekf_side_now = false;          // set by orcvio_msckf_io_step_frame around its first update
    bool ekf_side_used = false;         // the frame being enqueued has work on `side` (a repair drains it)
    long long ekf_side_skip_until = 0;  // a frame whose side-stream join gave up was repaired: the next 4 096 frames keep the rows on the update's own stream
                                        // (a profiler that serialises dispatches, another tenant: the join must not cost its bound on every frame)
    unsigned ekf_side_seq = 0;
    unsigned* front_mark = nullptr; unsigned front_mark_val = 0;       // launch_front: FrontGramArgs.started
    const unsigned* asm_wait = nullptr; unsigned asm_wait_val = 0;     // launch_solve_stage(ST_FORM_U): k_gemm_asmA_w
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

#include "capi_handle.inc"   // chi-square quantile, the arenas, create / destroy, options
#include "capi_update.inc"   // upload, the launches of one update, the launch-graph cache, run_local / run_finish
#include "capi_hybrid.inc"   // EKF-SLAM rows of the hybrid filter, features entering the state, the H_1 / H_2 tail
#include "capi_io.inc"   // download, the zero-copy update (io_begin / io_update), the copying one-shot call, gate_tracks
#include "capi_objects.inc"   // the object update: staging, compression pipeline, finish / download, ObjectLM messages, row evaluation
#include "capi_ipc.inc"   // the second transport of the communicator: HIP IPC + shared memory (several ranks per device possible)
#include "capi_comm.inc"   // the handle's RCCL communicator, bounded waits, the sharded updates
#include "capi_cov.inc"   // per-kernel profile, the device-resident covariance and its square-root factor
#include "capi_frame.inc"   // one frame in one call: feature update + object update, the objects' compression beside the features' solve
#include "capi_step.inc"   // one FILTER frame in one call: propagate, augment, update, prune update, marginalise on the resident covariance
#include "capi_state.inc"   // triangulation, incrementState_IMUCam (host arithmetic)
#include "capi_debug.inc"   // diagnostics build only: test hooks and ablation timers

}  // extern "C"

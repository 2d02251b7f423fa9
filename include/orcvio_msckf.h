/*
 * orcvio_msckf.h -- C-ABI of the MI355X-native MSCKF measurement-update path.
 *
 * This is the drop-in boundary for ONE hot path of shanmo/OrcVIO: everything the
 * reference does between "a set of feature tracks / object residual rows is ready"
 * and "delta_x and the updated covariance are available" (SURVEY.md section 8).
 * The reference has no FFI layer; the seam is cut inside class OrcVIO
 * (reference include/orcvio/orcvio.h:200-214,393-396).  Each entry point below
 * names the reference code it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - plain C, caller-owned buffers, no exceptions, int status codes;
 *   - all floating point is FP64, ids / indices are int32;
 *   - matrices handed over by the host are dense.  P is symmetric so row- and
 *     column-major coincide; every other matrix states its layout;
 *   - error-state order is the reference's (src/orcvio.cpp:202-225,4497-4533):
 *       theta(0:3) v(3:6) p(6:9) bg(9:12) ba(12:15) theta_ext(15:18) t_ext(18:21)
 *       td(21) [IMU intrinsics 22:46 if leg_dim==46] | clone i: theta,p at leg_dim+6i
 *   - one update in flight per handle; calls return after the result is in the
 *     caller's buffers unless the name says _async / _device.
 *
 * Environment.  The library reads TEN environment variables, all of them here (tests/test_abi.py checks the built
 * library against this list); every ablation / stamp / experiment switch of docs/LAB_NOTES.md exists in the diagnostics
 * build only (liborcvio_msckf_dbg.so, -DORCVIO_DEBUG_HOOKS):
 *   ORCVIO_COMM_TRANSPORT    "ipc": the communicator uses HIP IPC + shared memory instead of RCCL (ranks of one node)
 *   ORCVIO_COMM_TIMEOUT_S    bound of every wait another rank can strand, seconds (default 60)
 *   ORCVIO_IPC_WAIT_S        bound of the ipc transport's device-side wait for a peer's block, seconds (default 20)
 *   ORCVIO_IPC_XDEV          "1": allow the ipc transport between ranks on DIFFERENT devices (unverified: refused otherwise)
 *   ORCVIO_RCCL_LIB          path of the RCCL library to dlopen (default librccl.so.1 / librccl.so)
 *   ORCVIO_IO_SPIN_SECONDS   how long the calling thread spins on the result flag before it falls back to a stream
 *                            synchronisation (default 2)
 *   ORCVIO_LA_SPIN           polls (~1 us each) before a wait inside the look-ahead factorisation gives up and the update is
 *                            run again in separate launches (default 1 << 22)
 *   ORCVIO_FRONT_SPIN        the same for the device-wide counter of the fused front end (default 1 << 19)
 *   ORCVIO_FRAME_CHAIN       "0": orcvio_msckf_io_update_frame solves the object update behind the feature update's commit
 *                            (bit-identical to the two calls) instead of chained to its factor (equal to rounding; default 1)
 *   ORCVIO_FRAME_OVERLAP     "0": orcvio_msckf_io_update_frame runs its two halves one behind the other
 */
#ifndef ORCVIO_MSCKF_H
#define ORCVIO_MSCKF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORCVIO_MSCKF_ABI_VERSION 1

/* status codes */
enum {
    ORCVIO_OK = 0,
    ORCVIO_ERR_INVALID = 1,        /* bad argument / null pointer / size mismatch      */
    ORCVIO_ERR_NO_DEVICE = 2,      /* no gfx950 device or HIP runtime failure at create */
    ORCVIO_ERR_CAPACITY = 3,       /* window / tracks exceed the handle's capacity      */
    ORCVIO_ERR_TRACK_TOO_LONG = 4, /* a track has more than ORCVIO_MAX_TRACK observations */
    ORCVIO_ERR_HIP = 5,            /* HIP runtime error during the call (see last_error) */
    ORCVIO_ERR_NOT_SPD = 6,        /* S = H P H^T + sigma^2 I not positive definite in double (sigma^2 lost beside H P H^T: a
                                      prior beyond ~1e16 sigma^2 in scale, sigma = 0), or a non-finite result (NaN / Inf in the
                                      prior, the poses or the noise).  NO UPDATE: the device leaves P and x alone (P+ = P,
                                      dx = 0), the resident covariance and its factor are untouched, and cov_commit refuses
                                      until the next successful update.  (The reference has no such guard: its LDLT
                                      returns whatever comes out, src/orcvio.cpp:1690-1697.) */
    ORCVIO_ERR_TIMEOUT = 7,        /* a bounded wait gave up: an in-launch hand-off (k_front's device-wide counter, a solver
                                      wavefront of k_potrf_solve) after its retry, the creation of the communicator, or a rank
                                      that never arrived at a collective.  NO UPDATE, nothing to commit (cov_commit refuses);
                                      after a time-out inside a collective the communicator has been aborted (comm_info
                                      reports world = 0) and must be created again */
    ORCVIO_ERR_PEER = 8            /* sharded calls: ANOTHER rank could not take part with its share (its tracks were refused:
                                      capacity, a track too long, an index out of range).  Every rank still went through the
                                      collective (the failing rank with an empty share), so nobody hangs; every rank returns an
                                      error (the failing rank its own status, the others this one) and NO rank has an update
                                      to commit */
};

#define ORCVIO_MAX_TRACK 32   /* observations per feature track handled by the wave kernel */
#define ORCVIO_MAX_CLONES 60  /* sliding-window clones per handle                          */
#define ORCVIO_CHI2_TABLE 500 /* reference chi_squre_num_threshold, src/orcvio.cpp:483      */

/* Flags that change hot-path arithmetic (reference YAML keys; src/orcvio.cpp:62-415). */
typedef struct orcvio_msckf_flags {
    int32_t leg_dim;               /* 22, or 46 with calib_imu_instrinsic (src/orcvio.cpp:196-199) */
    int32_t use_larvio;            /* use_larvio_flag                                      */
    int32_t use_left_perturbation; /* use_left_perturbation_flag                           */
    int32_t if_fej;                /* if_FEJ: position_FEJ in the Jacobians (:1104)         */
    int32_t estimate_td;           /* estimate_td: column 21 = observations_vel (:1211)     */
    int32_t discard_large_update;  /* discard_large_update_flag (:4479-4494), see stats     */
    double noise_feature;          /* noise_feature (sigma); squared inside (:106,113)      */
    double chi2_prob;              /* chi_square_threshold_feat, e.g. 0.95                  */
} orcvio_msckf_flags;

/* Sliding window of augmented IMU states (reference struct IMUState_Aug,
 * include/orcvio/imu_state.h:103-148; std::map order = index order here). */
typedef struct orcvio_msckf_window {
    int32_t n_clones;
    const double* R_b2w; /* [N][9] row-major: orientation (body -> world)          */
    const double* t_b_w; /* [N][3] position                                        */
    const double* t_fej; /* [N][3] position_FEJ (may alias t_b_w when !if_fej)      */
    const double* R_b2c; /* [N][9] row-major: R_imu_cam0 (per-clone copy)           */
    const double* t_c_b; /* [N][3] t_cam0_imu                                       */
} orcvio_msckf_window;

/* Feature tracks to be used in this update, CSR over observations (reference struct
 * Feature / MapServer, include/orcvio/feat/feature.hpp:34-269).  The caller lists
 * only the observations that take part: all of them for removeLostFeatures
 * (src/orcvio.cpp:2503-2519), only those of the clones being removed for
 * pruneImuStateBuffer (:2810-2845).  Tracks with fewer than 2 listed observations
 * are skipped (accept = 0, gamma = NaN). */
typedef struct orcvio_msckf_tracks {
    int32_t n_features;
    const double* p_w;        /* [F][3] Feature::position (world)                          */
    const int32_t* obs_ptr;   /* [F+1]                                                     */
    const int32_t* obs_clone; /* [nobs] window index of the observing clone, ascending     */
    const double* obs_z;      /* [nobs][2] Feature::observations (normalised coordinates)   */
    const double* obs_zvel;   /* [nobs][2] Feature::observations_vel; may be NULL if !estimate_td */
} orcvio_msckf_tracks;

/* Pre-evaluated object residual rows in window coordinates: the output of
 * OrcVIO::constructObjectResidualJacobians (src/orcvio.cpp:2017-2151) for one object,
 * i.e. the arguments of OrcVIO::removeLostObjects (:2154-2193), in compact form:
 * every row touches exactly one clone. */
typedef struct orcvio_msckf_object_rows {
    int32_t n_rows;           /* rows of Hx / Hf / res                                     */
    int32_t n_obj_cols;       /* columns of Hf (object state dim, 45 for a 12-keypoint car) */
    const int32_t* row_clone; /* [n_rows] window index of the clone the row belongs to     */
    const double* Hx6;        /* [n_rows][6] the 6 non-zeros of the Hx row (theta, p of that clone) */
    const double* Hf;         /* [n_rows][n_obj_cols] row-major                            */
    const double* res;        /* [n_rows]                                                  */
} orcvio_msckf_object_rows;

/* Caller-allocated outputs.  NULL pointers are skipped. */
typedef struct orcvio_msckf_result {
    double* dx;       /* [n]    delta_x = K r                      (src/orcvio.cpp:1697,1824) */
    double* P_out;    /* [n*n]  (I-KH)P, symmetrised              (:1741-1753)               */
    int32_t* accept;  /* [F]    chi-square gate result             (:1953-1976)               */
    double* gamma;    /* [F]    Mahalanobis distance of each block                            */
    double* H_thin;   /* [(n-15)*n] row-major compressed Jacobian: upper-triangular R placed in state
                         columns 15..n-1 with R^T R = H^T H (+1e-11 max(diag) I: the stacked Jacobian is
                         rank deficient in every update, the shift avoids a rank decision)           */
    double* r_thin;   /* [n-15]                                                              */
    double* K;        /* [n*(n-15)] row-major Kalman gain w.r.t. H_thin                       */
    double* G;        /* [n*n] row-major K*H_thin (basis independent, SURVEY.md note N1)      */
    /* dx, P_out, gamma, accept and G are the results, accurate to ~1e-12 against the double restatement (bar: 1e-6).  H_thin,
       r_thin and K are DIAGNOSTIC outputs: the update is computed in square-root form from the Gram block and never
       needs them; they are derived afterwards from a Cholesky factor of the singular Gram with the shift above and reproduce
       K * H_thin = G and K * r_thin = dx to ~1e-5 relative only (tests/test_gpu_parity.py).  K is basis dependent in any case
       (SURVEY.md note N1): compare G, not K, across implementations. */
    int32_t stats[8]; /* [0] stacked rows (accepted)  [1] rows of H_thin  [2] accepted blocks
                         [3] 1 if an update was applied to P  [4] 1 if the reference's
                         large-update test (:4479) would discard delta_x  [5] zero-variance
                         directions of the prior (dropped pivots of chol(P))  [6] != 0: some pivot of chol(P)
                         fell below -tol (the prior was not PSD)  [7] objects: rank-deficient H_f columns */
} orcvio_msckf_result;

typedef struct orcvio_msckf_handle orcvio_msckf_handle;

/* Version / capability probes (no device needed). */
int32_t orcvio_msckf_abi_version(void);
const char* orcvio_msckf_last_error(void);

/* Chi-square quantile used for the gating tables: replaces boost::math::quantile
 * (src/orcvio.cpp:486-494, 1962-1968).  Host-side, no device needed. */
double orcvio_msckf_chi2_quantile(int32_t dof, double prob);

/* Create / destroy a handle that owns device buffers, streams and the cache of captured launch
 * graphs.  device = HIP device ordinal.  Capacity bounds what later calls may pass.
 * A handle is NOT thread-safe: one thread at a time per handle (the filter thread of the reference is the only caller of these
 * call sites); different handles may be used from different threads.  orcvio_msckf_last_error() is thread-local. */
int32_t orcvio_msckf_create(int32_t device, int32_t max_clones, int32_t max_features,
                            int32_t max_observations, orcvio_msckf_handle** out);
void orcvio_msckf_destroy(orcvio_msckf_handle* h);

/* Options.  ORCVIO_OPT_MATERIALIZE_STACK (default 0): also write the stacked projected blocks
 * [H' | r'] (what the reference builds in H_msckf, src/orcvio.cpp:2497-2527) to device memory.  The
 * update itself never needs them; tests and callers that want H' switch it on. */
/* ORCVIO_OPT_FUSED_SOLVE (default 1): factor M and solve for Z in one launch (solver workgroups trail the
 * factorisation block step by block step); 0 = two launches (k_potrf_reg, k_trsm_lds).  Same arithmetic. */
/* ORCVIO_OPT_LOOKAHEAD_SOLVE (default 3; with ORCVIO_OPT_FUSED_SOLVE on, from six block steps = 81 active states): the one-launch solve
 * spreads the trailing update of chol(M) over one "far" workgroup per block row (k_potrf_solve_la, look-ahead depth 3: a far workgroup
 * applies the published panels to its row and hands it to the factorising workgroup three block steps before it is due); 0 = the whole
 * trailing matrix in the registers of ONE workgroup (k_potrf_solve).  Same arithmetic in the same order: bit-identical results; 32 against
 * 38 us at 30 clones.  The far workgroups are eight to ten more workgroups that must become resident while the launch runs: every wait is
 * bounded (seconds), and an update whose hand-off was lost is run again through k_potrf_solve inside the same call (counted in
 * orcvio_msckf_counters [0]); object updates report ORCVIO_ERR_TIMEOUT instead, as they do for k_potrf_solve's own hand-off. */
/* ORCVIO_OPT_FUSED_FRONT (default 1): the Cholesky of the prior runs as workgroup 0 of the feature launch (k_front)
 * whenever the whole front end is co-resident (n <= 224, 1 + ceil(F/2) workgroups <= compute units); 0 = always fork
 * it to the handle's side stream around k_feature.  Same kernels' bodies, same arithmetic. */
/* ORCVIO_OPT_EXTRA_STATES (default 0): number of state columns BEHIND the clones that no row of the update touches --
 * the EKF-SLAM feature states (and Schmidt nuisance states) the reference keeps at the end of state_cov when
 * max_features_in_one_grid > 0 (src/orcvio.cpp:1495-1510): its H_msckf then has zero columns there
 * (featureJacobian_msckf builds its rows state_cov.cols() wide, :1191-1192) and the update still moves those states and
 * their covariance through the cross terms.  With value k, P / P_out are (LEG + 6N + k)^2 and dx has that length. */
/* ORCVIO_OPT_EKF_ROWS (default 0): the next uploads may be followed by orcvio_msckf_upload_ekf_rows -- the extra states
 * are then active columns of the compressed block (the rows of the SLAM features reach into them). */
/* ORCVIO_OPT_STAGE_PROFILE (default 0): record HIP events between the stages of the object update (rows, compression, the
 * batched factorisation of F, Y, A', the solve, the gate); orcvio_msckf_profile_stages returns the per-stage device times
 * of the last object update (SURVEY.md 8d "device-only time per stage from hipEvents"). */
/* ORCVIO_OPT_RESIDENT_FACTOR (default 1): orcvio_msckf_cov_commit also keeps a square-root factor of the committed
 * covariance (S+ = sigma Z^T, a by-product of the solve), and an update whose prior is the resident covariance (P == NULL)
 * uses it instead of a Cholesky factorisation of P: the second and third update of a frame (pruneImuStateBuffer,
 * processObjects; src/orcvio.cpp:591-594, System.cpp:551-555) skip that 45 us chain.  cov_augment / cov_remove_clones carry
 * the factor along, cov_set / cov_propagate drop it.  0 = always factor P. */
/* ORCVIO_OPT_OBJECT_QR (default 1): the left-nullspace projection of an object's rows against Hf (math_utils.hpp:287-312,
 * the reference takes a full-U SVD) needs the triangular factor of Hf.  1: a structured Householder QR that uses the arrow
 * shape of ObjectLM's state ([pose 6 | shape 3 | 3 per keypoint], include/orcvio/obj/ObjectLM.h:117-123; accuracy
 * cond(Hf) eps -- Hf of a real object has cond ~ 1e8: the keypoint rows have a gauge that only the bbox rows break).  0, and
 * for rows whose Hf does not have that shape: chol(Hf^T Hf) (cond^2: directions below ~1e-8 of the largest are treated as
 * null; stats[7] counts them). */
/* ORCVIO_OPT_SCHMIDT_STATES (default 0) = k: the Schmidt-EKF of the reference (use_schmidt, src/orcvio.cpp:1730-1751, :1893-1935,
 * :2881-2920): the LAST 6 k of the extra states are nuisance states -- clones that left the window but stay in state_cov.  The
 * update leaves the 6k x 6k nuisance block of the covariance as it was (:1740-1751); SLAM features may
 * be anchored at them: anchor index N + j in orcvio_msckf_slam_features / _ekf_rows / _new_features addresses nuisance state j,
 * whose pose comes from orcvio_msckf_upload_nuisance_poses and whose Jacobian block lands in its columns of the nuisance block
 * (:1591-1606).  N + k <= max_clones.  See also orcvio_msckf_cov_clones_to_nuisance, orcvio_msckf_augment_state_nuisance. */
/* ORCVIO_OPT_REF_STACK_HF (default 0): compatibility with the reference's stacking of SEVERAL objects in one call.
 * System::processObjects concatenates Hx, Hf and r of all objects vertically with Hf's 45 columns SHARED
 * (ros_wrapper/src/orcvio/src/System.cpp:684-702) and removeLostObjects projects the whole stack against that single Hf
 * (src/orcvio.cpp:2154-2193) -- the objects are treated as one object observed many times (SURVEY note N3; right for one
 * object per call only).  0: every object is projected against its own Hf (block-diagonal Hf; equal to the reference whenever
 * one object arrives).  1: the literal shared-Hf stack, dof = total rows - columns; all objects must have the same state size. */
/* ORCVIO_OPT_OBJECT_DOF (default 0): degrees of freedom of the object gate (src/orcvio.cpp:2172-2176).  0: rows - columns of H_f,
 * the reference's count.  1: rows - rank(H_f).  They differ only for a rank-deficient H_f (a keypoint never seen inside the window
 * has three zero columns, one seen in a single frame two rows for three columns): the reference keeps rows - columns directions of
 * the left null space -- which ones is decided by rounding noise in its SVD -- while this library projects onto the whole null
 * space (rows - rank directions), so that with the reference's count gamma sums more directions than its threshold
 * counts (biased towards rejection); 1 makes the threshold count what gamma sums.  (One more host synchronisation per object
 * update in that mode: the rank is what the structured QR finds on the device.) */
/* ORCVIO_OPT_REF_H2_LDLT (default 0): the tail of measurementUpdate_hybrid for features ENTERING the state.  The reference solves
 * with `H_2.ldlt()` (src/orcvio.cpp:1826-1827) where H_2 is the upper-triangular R of the new features' H_f (SPQR, :2421-2436):
 * Eigen's LDLT reads the LOWER triangle only, so the reference divides by diag(H_2).  0 (default): the triangular system is
 * solved (what the algebra of :1818-1821 means; identical to the reference for feature_idp_dim = 1, every shipped configuration).
 * 1: the reference's literal arithmetic (affects orcvio_msckf_augment_new_features and orcvio_msckf_cov_commit_new_features; the
 * handle-less orcvio_msckf_augment_state has the twin orcvio_msckf_augment_state_ref_ldlt).  P22 uses (H_2^T H_2)^-1 either way,
 * as the reference does (:1907-1908). */
/* ORCVIO_OPT_OBJECT_REFINE (default 1): how an object whose H_f is ill conditioned is projected (structured-QR route only).  The
 * fast route takes Y = Q_1^T [H_x | r] from the semi-normal equations R^-T (H_f^T [H_x | r]), accurate to cond(H_f) eps; the
 * reference's full-U SVD (math_utils.hpp:287-312) is backward stable.  1: objects whose triangular factor R has |R|_F |R^-1|_F above 3e6
 * (an estimate of cond(H_f) from above, formed on the device; every real car: cond(H_f) ~ 3e8 on the reference's one_car frames) form the basis explicitly, row by row (q_i R = h_i), take
 * Y from it and correct for its loss of orthonormality -- delta_x within 1e-9 of a 50-digit evaluation where the fast
 * route has 1.4e-6.  0: never.  2: every object.  orcvio_msckf_objects_refined reports how many objects of the last update took it.
 * Object TRACKS that qualify for the one-launch compression (orcvio_msckf_counters [5]) take the explicit basis for every object in
 * modes 1 and 2 alike -- there it is the only route, with the orthonormalisation applied to the rows of the basis (to first order,
 * like the correction above); mode 0 sends the tracks through the three-launch pipeline and its fast route. */
enum { ORCVIO_OPT_MATERIALIZE_STACK = 1, ORCVIO_OPT_FUSED_SOLVE = 2, ORCVIO_OPT_FUSED_FRONT = 3, ORCVIO_OPT_EXTRA_STATES = 4,
       ORCVIO_OPT_EKF_ROWS = 5, ORCVIO_OPT_STAGE_PROFILE = 6, ORCVIO_OPT_RESIDENT_FACTOR = 7, ORCVIO_OPT_OBJECT_QR = 8,
       ORCVIO_OPT_REF_STACK_HF = 9, ORCVIO_OPT_SCHMIDT_STATES = 10, ORCVIO_OPT_OBJECT_DOF = 11, ORCVIO_OPT_REF_H2_LDLT = 12,
       ORCVIO_OPT_OBJECT_REFINE = 13, ORCVIO_OPT_LOOKAHEAD_SOLVE = 14 };
int32_t orcvio_msckf_set_option(orcvio_msckf_handle* h, int32_t option, int32_t value);

/* EKF-SLAM rows of the hybrid filter (existing SLAM features).  For every SLAM feature the current
 * state observes, the reference evaluates featureJacobian_ekf (src/orcvio.cpp:1575-1651: measurementJacobian_ekf_3didp
 * :1229-1353 or _1didp :1356-1478), gates the row pair with 2 degrees of freedom (:2457) and stacks what passed under
 * the MSCKF rows for ONE update (measurementUpdate_hybrid, :1766-1950).  The rows come in the compact form of those
 * functions' outputs; gate and stacking happen on the device.  Call order: set_option(ORCVIO_OPT_EXTRA_STATES, k),
 * set_option(ORCVIO_OPT_EKF_ROWS, 1), upload(...) [P is (LEG+6N+k)^2], upload_ekf_rows, run_update, download,
 * download_ekf.  The rows belong to the upload they follow.  New features (the H_1 / H_2 initialisation) stay with the
 * caller. */
typedef struct orcvio_msckf_ekf_rows {
    int32_t n_features;
    int32_t idp_dim;         /* feature_idp_dim: 3 (invParam) or 1 (invDepth)                               */
    const int32_t* anchor;   /* [F] anchor clone (index in the window, Feature::id_anchor)                  */
    const int32_t* state;    /* [F] observing clone (state_server.imu_state.id)                             */
    const int32_t* slot;     /* [F] position in feature_states: columns LEG + 6N + idp_dim*slot ...         */
    const double* H_e;       /* [F][2][6]  -> columns 15..20 (:1641)                                        */
    const double* H_a;       /* [F][2][6]  -> the anchor clone (:1639)                                      */
    const double* H_x;       /* [F][2][6]  -> the observing clone (:1640)                                   */
    const double* H_f;       /* [F][2][idp_dim] -> the feature's own columns (:1630, :1636)                 */
    const double* z_vel;     /* [F][2] observations_vel -> column 21 under estimate_td (:1642), else unused  */
    const double* r;         /* [F][2]                                                                      */
} orcvio_msckf_ekf_rows;
int32_t orcvio_msckf_upload_ekf_rows(orcvio_msckf_handle* h, const orcvio_msckf_ekf_rows* rows);
/* ... or the SLAM features themselves: the four blocks are then evaluated on the device (measurementJacobian_ekf_3didp,
 * src/orcvio.cpp:1229-1353, / _1didp, :1356-1478) from the window poses already uploaded -- observations in, delta_x out.
 * Same call order, this call instead of orcvio_msckf_upload_ekf_rows. */
typedef struct orcvio_msckf_slam_features {
    int32_t n_features;
    int32_t idp_dim;          /* feature_idp_dim: 3 or 1                                                     */
    const int32_t* anchor;    /* [F] Feature::id_anchor as a window index                                    */
    const int32_t* state;     /* [F] the observing clone (state_server.imu_state.id)                         */
    const int32_t* slot;      /* [F] position in feature_states                                              */
    const double* param;      /* [F][3] idp 3: Feature::invParam; idp 1: Feature::obs_anchor                 */
    const double* inv_depth;  /* [F]    idp 1: Feature::invDepth (NULL for idp 3)                            */
    const double* p_w;        /* [F][3] Feature::position                                                    */
    const double* p_fej;      /* [F][3] Feature::position_FEJ, read under if_fej (may be NULL otherwise)      */
    const double* z;          /* [F][2] observations[imu_state.id]                                           */
    const double* z_vel;      /* [F][2] observations_vel[imu_state.id], read under estimate_td               */
} orcvio_msckf_slam_features;
int32_t orcvio_msckf_upload_slam_features(orcvio_msckf_handle* h, const orcvio_msckf_slam_features* feats);
/* The gate alone, no update: gatingTestFeature (src/orcvio.cpp:1953-1976) of featureJacobian_msckf for every listed
 * track against the prior P -- the test the reference applies to a feature before it may enter the state as a SLAM
 * feature (:2361-2367).  gamma [F], accept [F]. */
int32_t orcvio_msckf_gate_tracks(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, const orcvio_msckf_window* window,
                                 const orcvio_msckf_tracks* tracks, const double* P, double* gamma, int32_t* accept);

/* Rows the caller has projected and gated itself, stacked under everything else as they are: H [n_rows][n] over the
 * WHOLE state (n = LEG + 6N + extra states; the first 15 columns must be zero, as for every feature row), r [n_rows].
 * This is how a frame that initialises NEW SLAM features keeps the heavy update on the device: the caller evaluates
 * featureJacobian_ekf_new and the W = [V | U] split of src/orcvio.cpp:2337-2436 (a few features, small), hands the
 * V-part rows (zero in the new features' columns) over here, and applies the reference's own H_1 / H_2 algebra
 * (:1811-1947) to the downloaded delta_x and covariance.  The rows belong to the upload they follow. */
int32_t orcvio_msckf_upload_dense_rows(orcvio_msckf_handle* h, int32_t n_rows, const double* H, const double* r);
/* New SLAM features in the 3-parameter form need no special rows when the MSCKF rows are LARVIO's (use_larvio = 1: the SLAM
 * rows' own error-state convention) and if_FEJ = 0: list them among the tracks (the V part of their rows is their MSCKF
 * block).  After the update, this call returns their correction and the augmented
 * covariance -- measurementUpdate_hybrid, src/orcvio.cpp:1811-1821 and :1904-1947, without nuisance states -- from what the
 * feature kernel left on the device for those tracks (host arithmetic on a few 3 x n blocks):
 *   track[j]      index of new feature j among the tracks of the last upload (it must have been accepted)
 *   anchor[j]     its anchor clone (window index);  inv_param [j][3] = Feature::invParam
 *   dx, P_upd     delta_x [n] and covariance [n][n] of the update just downloaded
 *   dx_new [3 n_new];  P_aug [(n + 3 n_new)^2] = [[P_upd, (-HH P)^T], [-HH P, P22]] */
int32_t orcvio_msckf_augment_new_features(orcvio_msckf_handle* h, const orcvio_msckf_window* window, int32_t n_new,
                                          const int32_t* track, const int32_t* anchor, const double* inv_param,
                                          const double* dx, const double* P_upd, double* dx_new, double* P_aug);
/* Host arithmetic for features entering the state, either parametrisation (no device involved):
 * orcvio_msckf_new_feature_rows -- featureJacobian_ekf_new (src/orcvio.cpp:1481-1572) for the listed features (CSR of ALL
 *   their observations; the 1-parameter form drops the anchor's own, :1494-1496) and the rotation by W = [V | U]
 *   (:2416-2436), one small Householder QR per feature.  H_top [*rows_top][n_cols], r_top: the V parts, for
 *   orcvio_msckf_upload_dense_rows (capacity: 2 x observations rows).  H_1 [d n_new][n_cols], H_2 [n_new][d][d], r_1.
 *   param: invParam (d = 3) or obs_anchor (d = 1); inv_depth: d = 1 only.  n_cols = state_cov.cols() before the update.
 * orcvio_msckf_augment_state -- the tail of measurementUpdate_hybrid (:1818-1821, :1904-1947, no nuisance states) from
 *   the delta_x [n] and covariance [n][n] of the update: dx_new [d n_new], P_aug [(n + d n_new)^2]. */
int32_t orcvio_msckf_new_feature_rows(const orcvio_msckf_flags* flags, const orcvio_msckf_window* window, int32_t idp_dim, int32_t n_cols,
                                      int32_t n_new, const int32_t* anchor, const double* param, const double* inv_depth,
                                      const double* p_w, const double* p_fej, const int32_t* obs_ptr, const int32_t* obs_clone,
                                      const double* obs_z, const double* obs_zvel, int32_t* rows_top, double* H_top, double* r_top,
                                      double* H_1, double* H_2, double* r_1);
int32_t orcvio_msckf_augment_state(int32_t n, int32_t n_new, int32_t idp_dim, const double* H_1, const double* H_2, const double* r_1,
                                   double sigma2, const double* dx, const double* P_upd, double* dx_new, double* P_aug);
/* ... or on the device (either parametrisation, FEJ, td): featureJacobian_ekf_new (src/orcvio.cpp:1481-1572) and the W = [V | U]
 * split (:2416-2436) for the features that enter the state, evaluated from the window poses of the upload they follow.  The V
 * parts are appended to the dense rows of that upload (behind those of orcvio_msckf_upload_dense_rows, if any) and take part in
 * the update; the U parts stay on the device until orcvio_msckf_download_new_feature_blocks fetches H_1 [d k][n], H_2 [k][d][d],
 * r_1 [d k] for orcvio_msckf_augment_state.  param: invParam (d = 3) or obs_anchor (d = 1); CSR of ALL observations of each
 * feature (the 1-parameter form drops the anchor's own, :1494-1496).  Call order: upload, [upload_slam_features],
 * [upload_dense_rows], upload_new_features, run_update, download, download_new_feature_blocks, augment_state. */
typedef struct orcvio_msckf_new_features {
    int32_t n_features;
    int32_t idp_dim;          /* feature_idp_dim: 3 or 1                                      */
    const int32_t* anchor;    /* [k] Feature::id_anchor as a window index                     */
    const double* param;      /* [k][3] idp 3: invParam; idp 1: obs_anchor                    */
    const double* inv_depth;  /* [k]    idp 1: invDepth (NULL for idp 3)                      */
    const double* p_w;        /* [k][3] Feature::position                                     */
    const double* p_fej;      /* [k][3] Feature::position_FEJ, read under if_fej              */
    const int32_t* obs_ptr;   /* [k+1] CSR over the features' observations                    */
    const int32_t* obs_clone; /* [nobs] window index of the observing state                   */
    const double* obs_z;      /* [nobs][2]                                                    */
    const double* obs_zvel;   /* [nobs][2], read under estimate_td                            */
} orcvio_msckf_new_features;
int32_t orcvio_msckf_upload_new_features(orcvio_msckf_handle* h, const orcvio_msckf_new_features* feats);
int32_t orcvio_msckf_download_new_feature_blocks(orcvio_msckf_handle* h, double* H_1, double* H_2, double* r_1);
/* Schmidt-EKF (ORCVIO_OPT_SCHMIDT_STATES): poses of the nuisance states (state_server.nui_imu_states), after orcvio_msckf_upload;
 * nui->n_clones must equal the option's value. */
int32_t orcvio_msckf_upload_nuisance_poses(orcvio_msckf_handle* h, const orcvio_msckf_window* nui);
/* orcvio_msckf_augment_state with nui_rows nuisance rows at the end of the state: the new feature states are inserted in front
 * of them (src/orcvio.cpp:1920-1935); P_aug in the order [old | new | nuisance]. */
int32_t orcvio_msckf_augment_state_nuisance(int32_t n, int32_t n_new, int32_t idp_dim, int32_t nui_rows, const double* H_1, const double* H_2,
                                            const double* r_1, double sigma2, const double* dx, const double* P_upd, double* dx_new,
                                            double* P_aug);
/* orcvio_msckf_augment_state_nuisance (nui_rows may be 0) with the reference's LITERAL `H_2.ldlt().solve(..)` for HH and dx_new
 * (src/orcvio.cpp:1826-1827: an LDLT of the lower triangle of the upper-triangular H_2, i.e. division by its diagonal); see
 * ORCVIO_OPT_REF_H2_LDLT.  Differs from the corrected form for feature_idp_dim = 3 only. */
int32_t orcvio_msckf_augment_state_ref_ldlt(int32_t n, int32_t n_new, int32_t idp_dim, int32_t nui_rows, const double* H_1, const double* H_2,
                                            const double* r_1, double sigma2, const double* dx, const double* P_upd, double* dx_new,
                                            double* P_aug);
/* gamma[F], accept[F] of the SLAM features of the last update (either may be NULL) */
int32_t orcvio_msckf_download_ekf(orcvio_msckf_handle* h, double* gamma, int32_t* accept);

/* Feature update: replaces the loop + compression + update of
 * OrcVIO::removeLostFeatures (src/orcvio.cpp:2497-2560) and of
 * OrcVIO::pruneImuStateBuffer (:2803-2851): featureJacobian_msckf ->
 * nullspace_project_inplace_svd -> gatingTestFeature -> stack -> QR compression ->
 * measurementUpdate_{hybrid(pure MSCKF case),msckf}.  Host buffers in, host buffers out;
 * P is n x n with n = leg_dim + 6*n_clones. */
int32_t orcvio_msckf_update_features(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags,
                                     const orcvio_msckf_window* window,
                                     const orcvio_msckf_tracks* tracks, const double* P,
                                     orcvio_msckf_result* result);

/* ---- The same update without the copies: the handle's pinned arena, written and read in place ------------------------------
 * SURVEY.md 8d measures the update from "flat inputs in host memory" to "delta_x, P+ in host memory".  orcvio_msckf_update_features
 * copies the caller's arrays into the handle's pinned input arena and the results out of its pinned output block; a caller that
 * flattens its containers (StateServer / MapServer, src/orcvio.cpp:2497-2527, :2803-2848) STRAIGHT INTO the arena and reads the
 * results where they land saves both copies (326 KB of P each way at 30 clones).
 *   io_begin   lays the arena out for (n_clones, n_features, n_observations; with_P = 0: the prior is the resident covariance; 2: ... as the propagation and
 *              augmentation of orcvio_msckf_io_step_frame will leave it, for that call)
 *              and returns the pointers.  Inputs, to be written by the caller: poses [N][ORCVIO_POSE_STRIDE] (one record per
 *              clone: R_b2w 9 row-major | t_b_w 3 | t_fej 3 | R_b2c 9 | t_c_b 3 | 1 unused), p_w [F][3], obs_ptr [F+1] (CSR,
 *              obs_ptr[F] == n_observations), obs_clone, obs_z [nobs][2], obs_zvel [nobs][2] (NULL unless flags->estimate_td),
 *              P [n][n] (NULL when with_P == 0).  Outputs, valid after io_update returns ORCVIO_OK and until the next call on
 *              the handle that takes tracks: dx [n], gamma [F], accept [F], P_out [n][n] (written only with want_P).
 *   io_update  validates what stands in the arena (the index arrays, as orcvio_msckf_upload does), and runs the update: ONE
 *              launch of a captured graph whose first kernel pulls the arena into HBM and whose last kernel pushes the results
 *              into the (host-coherent) output block and raises a flag word there; the calling thread spins on that word -- no
 *              copy-engine transfer, no stream synchronisation.  commit != 0: orcvio_msckf_cov_commit is part of the same launch
 *              (P+ and its square-root factor become resident; refused on the device if the update is).  stats as in
 *              orcvio_msckf_result.  The same arena may be updated again (next frame, same sizes) without a new io_begin.
 * Status codes as orcvio_msckf_update_features; ORCVIO_ERR_TIMEOUT if the results were never published. */
#define ORCVIO_POSE_STRIDE 28
typedef struct orcvio_msckf_io {
    int32_t n;            /* leg_dim + 6 n_clones (+ extra states) */
    double* poses;
    double* p_w;
    int32_t* obs_ptr;
    int32_t* obs_clone;
    double* obs_z;
    double* obs_zvel;
    double* P;
    const double* dx;
    const double* gamma;
    const int32_t* accept;
    const double* P_out;
} orcvio_msckf_io;
int32_t orcvio_msckf_io_begin(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, int32_t n_clones, int32_t n_features,
                              int32_t n_observations, int32_t with_P, orcvio_msckf_io* io);
int32_t orcvio_msckf_io_update(orcvio_msckf_handle* h, int32_t want_P, int32_t commit, int32_t* stats /* [8] or NULL */);
/* The same in two halves, for a caller with work of its own to do meanwhile (the next image, the object mapper's reply):
 *   io_submit   validates the arena and LAUNCHES the update (returns after ~20 us of host time; nothing is waited for)
 *   io_collect  waits for the results (the flag word) and reports the outcome exactly as io_update does
 * Between the two calls the handle must not be used for anything else, and the arena must not be written.
 * Hybrid filter: the rows of the in-state features (orcvio_msckf_upload_slam_features / _upload_ekf_rows, with ORCVIO_OPT_EKF_ROWS and
 * ORCVIO_OPT_EXTRA_STATES set) may be handed over between io_begin and io_update / io_submit; they ride in the same launch. */
int32_t orcvio_msckf_io_submit(orcvio_msckf_handle* h, int32_t want_P, int32_t commit);
int32_t orcvio_msckf_io_collect(orcvio_msckf_handle* h, int32_t* stats /* [8] or NULL */);


/* Object update: replaces OrcVIO::removeLostObjects (src/orcvio.cpp:2154-2193) for one
 * object block (nullspace projection against Hf -> gate with dof = rows -> NaN check ->
 * measurementUpdate_msckf).  result->accept/gamma have length 1. */
int32_t orcvio_msckf_update_objects(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags,
                                    int32_t n_clones, const orcvio_msckf_object_rows* rows,
                                    int32_t n_objects, const double* P,
                                    orcvio_msckf_result* result);

/* Object residual rows at a fixed state: the CameraLM / ObjectLM functor evaluation exported by
 * ObjectFeatureInitializer::single_levenberg_marquardt (src/obj/ObjectFeatureInitializer.cpp:394-434; functors
 * src/obj/ObjectResJacCam.cpp:153-519, src/obj/ObjectLM.cpp:250-632) followed by
 * OrcVIO::constructObjectResidualJacobians (src/orcvio.cpp:2017-2151), evaluated on the device.  Output rows are
 * interleaved per in-window frame [keypoint rows ; 4 bbox rows] and can be passed to
 * orcvio_msckf_update_objects as they are.  The LM *solver* is not part of this library: the caller supplies
 * the state at which the rows are evaluated.
 * The rows are evaluated with UNIT residual weights and the Huber loss OFF: the functors' `residual_weight` and
 * `use_valid` / huber arguments (src/obj/ObjectResJacCam.cpp:535, 567-576; src/obj/ObjectLM.cpp:775-811) are ones / infinity
 * at every shipped call site (include/orcvio/obj/ObjectFeatureInitializer.h:40, ObjectLM.h:301), so no entry point exposes them. */
typedef struct orcvio_object_eval_flags {
    int32_t use_left_perturbation;      /* the object mapper's flag (ObjectInitNode.cpp:140)             */
    int32_t use_new_bbox_residual;      /* use_new_bbox_residual_flag (:206).  1: the reference's rows LITERALLY -- its Jacobians take the
                                           plane in the world frame (src/obj/ObjectResJacCam.cpp:405,446) although the residual uses the
                                           object frame, and the shape derivative lacks -sign(b4) (src/obj/ObjectLM.cpp:602-603): SURVEY
                                           note N8, the parity target.  2 (opt-in): the same residual with CORRECTED Jacobians (plane in
                                           the object frame, sign restored), which match central differences to 1e-9 */
    int32_t vio_use_left_perturbation;  /* the filter's flag: selects D (src/orcvio.cpp:2092)            */
    int32_t fix_dcampose_dimupose_to_identity; /* OrcVIO::fixDcamposeDimuposeToI (orcvio.h:115-119)      */
    double R_b2c[9];                    /* current extrinsics: state_server.imu_state.R_imu_cam0         */
    double t_c_b[3];                    /*                      state_server.imu_state.t_cam0_imu         */
} orcvio_object_eval_flags;

typedef struct orcvio_object_track {
    int32_t n_keypoints;        /* K (12 for the car class, config/object_feat_unity.yaml).  K = 0: a BBOX-ONLY track -- object state
                                   [pose 6 | shape 3], four bbox rows per in-window frame (src/obj/ObjectResJacCam.cpp:308-519 alone), H_f
                                   has 9 columns; kps / frame_zs are not read.  An EXTENSION (BASELINE config 5's "bbox-only OrcVIO-lite"
                                   read literally): the reference's lite mode sends no residuals at all (SURVEY note N4) */
    int32_t n_frames;
    const double* wTo;          /* [16] row-major object -> world                                        */
    const double* shape;        /* [3]  ellipsoid semi-axes                                              */
    const double* kps;          /* [K][3] keypoints in the object frame                                  */
    const double* frame_wTc;    /* [F][16] camera -> world of every frame (exp of valid_camera_pose_mat) */
    const double* frame_zs;     /* [F][K][2] normalised keypoint detections, NaN = not detected          */
    const double* frame_bbox;   /* [F][4] xmin, ymin, xmax, ymax (normalised)                            */
    const int32_t* frame_clone; /* [F] window index of the clone with that exact timestamp, or -1 (:2073) */
} orcvio_object_track;

/* Returns ORCVIO_OK and *n_rows = 0 if no frame of the object is in the window (the reference returns false,
 * :2149).  Hf has 9 + 3K columns.  cap_rows bounds the caller's buffers. */
int32_t orcvio_msckf_object_rows_eval(orcvio_msckf_handle* h, const orcvio_object_eval_flags* flags,
                                      const orcvio_object_track* obj, int32_t cap_rows, int32_t* n_rows,
                                      int32_t* row_clone, double* Hx6, double* Hf, double* res);

/* ---- staged, device-resident form (what bench.py times; also the multi-GPU path) ----
 * upload:      copy window / tracks / P to the handle's device buffers (host -> HBM);
 * run_local:   Jacobians -> nullspace -> gate -> stacked [H'|r'] -> Gram compression.
 *              Leaves this rank's compressed block [A | b] ((n-14) x (n-14), symmetric,
 *              row-major, padded to a multiple of 16) in device memory;
 * block_ptr:   device pointer / element count of that block (for an RCCL all-gather);
 * run_finish:  sums `n_blocks` compressed blocks found at d_blocks (device pointer,
 *              consecutive, same size) and performs the Kalman solve: dx, P+ on device;
 * run_update:  run_local + run_finish on the handle's own block (single GPU);
 * download:    copy results to host buffers.
 * `stream` is a hipStream_t passed as void* (NULL = the handle's own stream). */
int32_t orcvio_msckf_upload(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags,
                            const orcvio_msckf_window* window, const orcvio_msckf_tracks* tracks,
                            const double* P);
int32_t orcvio_msckf_run_local(orcvio_msckf_handle* h, void* stream);
int32_t orcvio_msckf_block_ptr(orcvio_msckf_handle* h, double** d_block, int64_t* n_elems);
/* run_local writing the block straight into caller-owned device memory (e.g. the send buffer of the
 * all-gather), n_elems doubles as reported by block_ptr. */
int32_t orcvio_msckf_run_local_to(orcvio_msckf_handle* h, double* d_dst, void* stream);
int32_t orcvio_msckf_run_finish(orcvio_msckf_handle* h, const double* d_blocks, int32_t n_blocks,
                                void* stream);
int32_t orcvio_msckf_run_update(orcvio_msckf_handle* h, void* stream);
int32_t orcvio_msckf_sync(orcvio_msckf_handle* h, void* stream);
int32_t orcvio_msckf_download(orcvio_msckf_handle* h, orcvio_msckf_result* result);

/* Per-kernel device time of the last run_update, measured with HIP events on the launch
 * stream.  names[i] / ms[i] for i < *count (count in: capacity, out: filled). */
int32_t orcvio_msckf_profile_update(orcvio_msckf_handle* h, void* stream, int32_t reps,
                                    const char** names, double* ms, int32_t* count);

/* Per-stage device times of the last object update (ORCVIO_OPT_STAGE_PROFILE must be on): names[i] / ms[i] for
 * i < *count (count in: capacity, out: filled). */
int32_t orcvio_msckf_profile_stages(orcvio_msckf_handle* h, const char** names, double* ms, int32_t* count);

/* State correction: replaces OrcVIO::incrementState_IMUCam (src/orcvio.cpp:4468-4567).
 * Pure host arithmetic (15 + 6N small updates); returns 1 if the correction was applied,
 * 0 if the reference's large-update test discarded it. */
typedef struct orcvio_msckf_state {
    double R_b2w_imu[9]; /* current IMU orientation (row-major) */
    double v[3], p[3], bg[3], ba[3];
    double R_b2c[9], t_c_b[3]; /* extrinsics of the current IMU state */
    double td;
    double imu_intrinsics[24]; /* T1..M2 when leg_dim == 46 */
    int32_t n_clones;
    double* clone_R_b2w; /* [N][9] in/out */
    double* clone_t_b_w; /* [N][3] in/out */
    double* clone_R_c2w; /* [N][9] out: orientation_cam */
    double* clone_t_c_w; /* [N][3] out: position_cam    */
} orcvio_msckf_state;
int32_t orcvio_msckf_increment_state(const orcvio_msckf_flags* flags, const double* dx,
                                     orcvio_msckf_state* state);

/* ---- Object update in two halves (the multi-GPU form of orcvio_msckf_update_objects) -----------------
 * Objects are dealt across the ranks; P and the window are replicated.  objects_local leaves this rank's compressed
 * block [A' b'; b'^T c'] (NAP x NAP doubles, NAP = 16*ceil((6*n_clones + leg_dim - 14)/16)) in d_dst (device memory;
 * NULL = the handle's own block, orcvio_msckf_block_ptr) and reports the degrees of freedom of its usable objects;
 * the caller all-gathers the blocks and sums the dofs; objects_finish sums the blocks in rank order, solves, gates
 * jointly with dof_total (src/orcvio.cpp:2172-2176) and applies or discards the update; objects_download returns
 * accept[0], gamma[0], dx, P_out (and G on request). */
int32_t orcvio_msckf_objects_local(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, int32_t n_clones,
                                   const orcvio_msckf_object_rows* objects, int32_t n_objects, const double* P,
                                   double* d_dst, int32_t* dof_out, void* stream);
int32_t orcvio_msckf_objects_finish(orcvio_msckf_handle* h, const double* d_blocks, int32_t n_blocks,
                                    int32_t dof_total, void* stream);
int32_t orcvio_msckf_objects_download(orcvio_msckf_handle* h, orcvio_msckf_result* result);
/* Cumulative counters of the handle since orcvio_msckf_create (count <= ORCVIO_COUNTERS values are written):
 *   [0] front_fallbacks   updates whose fused front end (k_front, ORCVIO_OPT_FUSED_FRONT) or look-ahead solve (ORCVIO_OPT_LOOKAHEAD_SOLVE) lost
 *                         an in-launch hand-off: for k_front its co-residency bet -- a workgroup gave
 *                         up at the launch's device-wide counter because another tenant of the device (a second handle, another
 *                         stream's long kernel, another process) held compute units -- and were re-run on the forked seven-launch
 *                         path inside the same call.  The results are the same; each such update costs the bounded wait (tens of ms)
 *                         plus a second run.  A caller that sees this grow shares the device: set ORCVIO_OPT_FUSED_FRONT = 0
 *                         
 *   [1] launch sequences captured into a hipGraph   [2] replayed from one   [3] enqueued as plain launches
 *   [4] 1 while the communicator blocks the fused front end (ipc transport with ranks sharing this device)
 *   [5] 1 if the last object update from tracks compressed its objects in ONE launch (k_obj_fused: rows evaluated into LDS, never
 *       materialised; every object through the explicit basis), 0 if it took the three-launch pipeline over materialised rows
 *       (an object with two frames on one clone, more than 32 in-window frames or 16 keypoints, rows beyond the LDS staging,
 *       ORCVIO_OPT_OBJECT_REFINE = 0, ORCVIO_OPT_REF_STACK_HF, ORCVIO_OPT_OBJECT_QR = 0)
 *   [6] frames (orcvio_msckf_io_update_frame) whose object solve ran chained to the feature update's factor (ORCVIO_FRAME_CHAIN)
 *   [7] frame calls that found their object tracks staged ahead (orcvio_msckf_io_stage_object_tracks)
 *   [8] filter frames through orcvio_msckf_io_step_frame   [9] updates of such frames run again after a lost in-launch hand-off */
#define ORCVIO_COUNTERS 10
int32_t orcvio_msckf_counters(orcvio_msckf_handle* h, int64_t* counters, int32_t count);

/* Objects of the last downloaded object update (any entry point) whose projection against H_f went through the explicit basis
 * (ORCVIO_OPT_OBJECT_REFINE; math_utils.hpp:287-312 is the step it stands for): this rank's objects only. */
int32_t orcvio_msckf_objects_refined(orcvio_msckf_handle* h, int32_t* count);

/* The object update straight from object TRACKS (state at the LM optimum + observations; the structs of
 * orcvio_msckf_object_rows_eval): the residual rows and Jacobians of CameraLM::Error{Feature,BBox}Quadric /
 * ObjectLM::df and constructObjectResidualJacobians are evaluated on the device into the compact row arrays, so only
 * the tracks cross PCIe (a 12-keypoint car seen in 30 frames is 9 KB of input against 360 KB of rows).
 * objects_local_tracks is the sharded first half (followed by orcvio_msckf_objects_finish / _objects_download). */
int32_t orcvio_msckf_update_object_tracks(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags,
                                          const orcvio_object_eval_flags* eval_flags, int32_t n_clones,
                                          const orcvio_object_track* tracks, int32_t n_tracks, const double* P,
                                          orcvio_msckf_result* result);
int32_t orcvio_msckf_objects_local_tracks(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags,
                                          const orcvio_object_eval_flags* eval_flags, int32_t n_clones,
                                          const orcvio_object_track* tracks, int32_t n_tracks, const double* P,
                                          double* d_dst, int32_t* dof_out, void* stream);

/* The object tracks of the NEXT orcvio_msckf_io_update_frame call, scanned and staged ahead of it.  The caller has them before the
 * frame's feature update starts -- they come from the object mapper (ros_wrapper/src/orcvio/src/System.cpp:622-708), not from the
 * image -- so their packing into the handle's pinned staging arena (~14 us of host time for twenty objects) need not sit inside the
 * frame call, between the tracks' launch and the compression's: it is done here, as the feature tracks are written into the arena by
 * the caller before the call.  Call between orcvio_msckf_io_begin (with_P = 0; the window's clone count is taken from it) and
 * orcvio_msckf_io_update_frame, and hand the SAME `tracks` pointer, count, flags and eval flags to the frame call: anything else -- or
 * any other object call in between -- and the frame call stages for itself, as without this call (orcvio_msckf_counters [7] counts the
 * frames that found their tracks staged).  The track data are COPIED here: later changes to them are not seen by the frame call.
 * Results are the frame call's, bit for bit, either way. */
int32_t orcvio_msckf_io_stage_object_tracks(orcvio_msckf_handle* h, const orcvio_msckf_flags* object_flags,
                                            const orcvio_object_eval_flags* eval_flags, const orcvio_object_track* tracks, int32_t n_tracks);

/* ---- One frame in one call: the feature update, then the object update on the covariance it leaves -------------------------
 * System::imageCallback runs Estimator->processFeatures (its last update: OrcVIO::removeLostFeatures / pruneImuStateBuffer) and
 * then processObjects -> OrcVIO::removeLostObjects on the same state (ros_wrapper/src/orcvio/src/System.cpp:548-554).  This call
 * is orcvio_msckf_io_update(h, 0, commit = 1, ..) on what stands in the arena (orcvio_msckf_io_begin with with_P = 0: the prior
 * is the resident covariance) followed by orcvio_msckf_update_object_tracks(.., P = NULL, ..) -- same kernels, same results --
 * except that the object tracks' COMPRESSION (rows, structured QR, A': it depends on neither the prior nor the feature update)
 * runs on a stream of its own beside the feature update's solve, where the device is otherwise idle; only the object solve
 * waits for the feature update's commit.
 *   features   dx [n], gamma [F], accept [F] (caller-owned, each may be NULL) and stats; P_out / K / G / thin outputs must be NULL
 *   objects    as orcvio_msckf_update_object_tracks (dx, gamma[0], accept[0], stats; P_out / G are served by the two calls in
 *              sequence); commit_objects != 0: orcvio_msckf_cov_commit after an accepted object update
 * Status: a refused feature update (ORCVIO_ERR_NOT_SPD ..) returns its code and NOTHING of the frame is applied; a failure of the
 * object half returns its code with the feature update applied and committed (features->stats[3] says so).  After the call the
 * arena belongs to the object update: the next frame starts with orcvio_msckf_io_begin.  ORCVIO_FRAME_OVERLAP=0 (environment)
 * runs the two halves one behind the other. */
int32_t orcvio_msckf_io_update_frame(orcvio_msckf_handle* h, orcvio_msckf_result* features, const orcvio_msckf_flags* object_flags,
                                     const orcvio_object_eval_flags* eval_flags, const orcvio_object_track* tracks, int32_t n_tracks,
                                     int32_t commit_objects, orcvio_msckf_result* objects);

/* ---- One FILTER frame in one call: propagate, augment, update, prune update, marginalise -- on the resident covariance ---------
 * What OrcVIO::processFeatures does to state_cov per image (src/orcvio.cpp:567-594): batchImuProcessing -> processModel's
 * covariance propagation (:800-816), stateAugmentation (:962-1010), removeLostFeatures' update (:2497-2560), pruneImuStateBuffer's
 * update on the clones that leave (:2803-2851) and their marginalisation (:2874-2956).  As separate calls these are
 * cov_propagate, cov_augment, io_begin + upload_slam_features + io_update(commit), io_begin + io_update(commit), cov_remove_clones:
 * ~30 launches and copies and two host round trips per frame.  Here everything is enqueued at once on the handle's stream -- the
 * covariance never leaves HBM, the calling thread waits ONCE, for the flag word behind the last update -- with the small steps
 * folded into a few launches (frame_ops.hpp).  Same arithmetic in the same order as the separate calls: bit-identical results
 * (tests/test_gpu_stream.py).
 * Call order:  io_begin(flags, N, F, nobs, with_P = 2)   N = clones AFTER this frame's augmentation; with_P = 2: "the prior is the
 *                                                         resident covariance as this frame's propagation + augmentation leave it"
 *              the caller writes poses / tracks of the first update into the arena (as for io_update)
 *              io_step_frame(step, result)
 *   Phi, Q           [leg][leg] accumulated state transition and process noise of the IMU block (NULL, NULL: no propagation)
 *   augment          != 0: a clone is appended behind the window's clones (in front of the extra states)
 *   slam_features    hybrid filter (ORCVIO_OPT_EKF_ROWS): the in-state features the current state observes; NULL: none
 *   prune_tracks     the second update of the frame: observations of the clones that leave only (CSR, window indices, p_w given);
 *                    NULL: none.  Its window is the first update's, its prior what the first update commits.
 *   prune_apply_dx   != 0: the window poses of the second update are the arena's poses incremented by the first update's dx ON THE
 *                    DEVICE (incrementState_IMUCam, src/orcvio.cpp:4468-4567: clone orientation / position and the extrinsics;
 *                    skipped when discard_large_update discards dx) -- what the reference's state is when pruneImuStateBuffer runs;
 *                    0: the same poses as the first update (the caller's state increment does not reach into this call)
 *   remove_clones    window indices (ascending) of the clones marginalised at the end; n_remove <= 8
 * Not in this call: features ENTERING the state (orcvio_msckf_upload_new_features / _cov_commit_new_features change the state's dimension
 * between the updates: such frames take the separate calls), Schmidt nuisance poses, a communicator on the handle.
 * Results: pointers into the handle's pinned output blocks, valid until the next call on the handle that takes tracks.
 * Status: a validation failure returns its code with NOTHING done.  A refusal on the device (ORCVIO_ERR_NOT_SPD: M not positive definite or
 * non-finite input) of the first update refuses the second as well; the covariance bookkeeping of the frame (propagation, augmentation,
 * marginalisation) stands either way, stats[3] / prune_stats[3] say which updates were applied, n_after is the dimension left.  A lost
 * in-launch hand-off is repaired inside the call (the affected updates run again in separate launches; orcvio_msckf_counters [0]). */
typedef struct orcvio_msckf_frame_step {
    int32_t leg_dim;
    const double* Phi;
    const double* Q;
    int32_t augment;
    const orcvio_msckf_slam_features* slam_features;
    const orcvio_msckf_tracks* prune_tracks;
    int32_t prune_apply_dx;
    const int32_t* remove_clones;
    int32_t n_remove;
} orcvio_msckf_frame_step;
typedef struct orcvio_msckf_frame_result {
    int32_t stats[8];              /* first update, as orcvio_msckf_result.stats                        */
    int32_t prune_stats[8];        /* second update                                                     */
    const double* dx;              /* [n]                                                               */
    const double* gamma;           /* [F]                                                               */
    const int32_t* accept;         /* [F]                                                               */
    const double* prune_dx;        /* [n]   NULL without prune_tracks                                   */
    const double* prune_gamma;     /* [F2]                                                              */
    const int32_t* prune_accept;   /* [F2]                                                              */
    int32_t n_after;               /* dimension of the resident covariance when the call returns        */
    int32_t status_first;          /* ORCVIO_OK or the refusal of the first update                      */
    int32_t status_prune;          /* ... of the second                                                 */
    int32_t repaired;              /* updates of this frame that were run again after a lost hand-off   */
} orcvio_msckf_frame_result;
int32_t orcvio_msckf_io_step_frame(orcvio_msckf_handle* h, const orcvio_msckf_frame_step* step, orcvio_msckf_frame_result* result);

/* ---- Object update straight from the wire format of the object mapper ------------------------------
 * One element per orcvio_ros_msgs/ObjectLM message (ros_wrapper/src/orcvio_ros_msgs/msg/ObjectLM.msg) as ObjectInitNode fills it
 * (ros_wrapper/src/orcvio/src/ObjectInitNode.cpp:1180-1207): the export block of single_levenberg_marquardt
 * (src/obj/ObjectFeatureInitializer.cpp:394-434).  Matrices are the `data` arrays of the std_msgs/Float64MultiArray fields;
 * tf::matrixEigenToMsg writes them ROW-MAJOR (dim[0] = rows, dim[1] = cols).  The reference's System::msgToEigen maps that
 * buffer as COLUMN-major (System.cpp:710-723: SURVEY note N4, the object update could never have been right over ROS);
 * wire_row_major = 1 (default semantics: what the sender meant) reads the buffers as they were written, 0 reproduces msgToEigen
 * literally.  Rows are ordered [2 per valid keypoint of frame 0, frame 1, ... ; 4 bbox rows of frame 0, frame 1, ...].
 * orcvio_msckf_update_object_lm_msgs does, per message, OrcVIO::constructObjectResidualJacobians (src/orcvio.cpp:2017-2151:
 * exact timestamp match against cur_window_timestamps, D = get_cam_wrt_imu_se3_jacobian of SE3::exp(valid_camera_pose_mat
 * column) with the CURRENT extrinsics, rows re-interleaved per in-window frame; host arithmetic), then the stacking of
 * System::processObjects (System.cpp:684-702; per-object projection unless ORCVIO_OPT_REF_STACK_HF) and
 * OrcVIO::removeLostObjects (:2154-2193) on the device. */
typedef struct orcvio_object_lm_msg {
    int64_t object_id;
    int32_t n_rows;                          /* rows of residual / both Jacobians                                     */
    int32_t n_obj_cols;                      /* columns of jacobian_wrt_object_state (45 for a 12-keypoint class)     */
    int32_t n_frames;                        /* timestamps.size() == zs_num_wrt_timestamps.size() == pose columns      */
    const double* residual;                  /* [n_rows]                                                              */
    const double* jacobian_wrt_object_state; /* n_rows x n_obj_cols                                                   */
    const double* jacobian_wrt_sensor_state; /* n_rows x 6                                                            */
    const double* valid_camera_pose_mat;     /* 6 x n_frames: se3 log (upsilon, omega) of every frame's wTc           */
    const double* timestamps;                /* [n_frames]                                                            */
    const int32_t* zs_num_wrt_timestamps;    /* [n_frames] valid keypoints per frame                                  */
} orcvio_object_lm_msg;
int32_t orcvio_msckf_update_object_lm_msgs(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags, int32_t n_clones,
                                           const double* cur_window_timestamps /* [n_clones] */, const double* R_b2c /* [9] */,
                                           const double* t_c_b /* [3] */, int32_t fix_dcampose_dimupose_to_identity,
                                           int32_t wire_row_major, const orcvio_object_lm_msg* msgs, int32_t n_msgs, const double* P,
                                           orcvio_msckf_result* result);

/* ---- Multi-GPU: the handle owns an RCCL communicator (SURVEY.md 8b "handle owns ... RCCL comm", 8e) --------------------
 * One process per GPU, one handle per process.  The reference has no collectives; the callers this serves are the same
 * three call sites (OrcVIO::removeLostFeatures src/orcvio.cpp:2497-2560, ::pruneImuStateBuffer :2803-2851,
 * System::processObjects -> removeLostObjects ros_wrapper/src/orcvio/src/System.cpp:622-708), each rank holding a share of the
 * tracks / objects and ALL of the window and the prior.  Per update: local tracks -> local compressed block -> ONE
 * ncclAllGather of the blocks over xGMI -> rank-ordered sum (bit-identical on every rank) -> replicated Kalman solve, so
 * every rank ends with the same delta_x and P+ and nothing is broadcast back.
 *   comm_unique_id   rank 0: ncclGetUniqueId into id[ORCVIO_COMM_ID_BYTES]; the caller ships the bytes to the other ranks
 *                    by whatever channel it has (MPI, a TCP store, torch.distributed, a file)
 *   comm_init        ncclCommInitRank on the handle's device; allocates the gather buffer
 *   comm_destroy     (also done by orcvio_msckf_destroy)
 * RCCL (librccl.so.1) is loaded with dlopen on the first of these calls: a single-GPU caller never needs it.
 * A second transport behind the same calls: with ORCVIO_COMM_TRANSPORT=ipc in the environment of every rank the blocks travel by
 * direct stores into the peers' gather buffers (HIP IPC) announced through a shared-memory segment (ranks of ONE node;
 * comm_unique_id then returns 128 random bytes and RCCL is never loaded).  Unlike RCCL it accepts several ranks on one device
 * (INTEGRATION.md 7a).  It has run with several ranks on ONE device only: ranks on DIFFERENT devices are unverified
 * (no multi-GPU machine was available; the gather buffer is fine-grained / uncached and every pushing thread fences at system scope,
 * which is what the case needs on paper).  Its device-side wait for a peer's block gives up after
 * min(ORCVIO_COMM_TIMEOUT_S, ORCVIO_IPC_WAIT_S [default 20]) seconds: the update then reports ORCVIO_ERR_TIMEOUT (no update); a block
 * that arrives with another update's sequence number (a rank whose call count has slipped) reports ORCVIO_ERR_PEER. */
#define ORCVIO_COMM_ID_BYTES 128
int32_t orcvio_msckf_comm_unique_id(uint8_t* id /* [ORCVIO_COMM_ID_BYTES] */);
int32_t orcvio_msckf_comm_init(orcvio_msckf_handle* h, const uint8_t* id, int32_t rank, int32_t world);
int32_t orcvio_msckf_comm_destroy(orcvio_msckf_handle* h);
int32_t orcvio_msckf_comm_info(orcvio_msckf_handle* h, int32_t* rank, int32_t* world); /* world = 0: no communicator */
/* What the communicator is, for logs and benchmarks (count <= 8 values are written): out[0] transport (0 none, 1 RCCL, 2 ipc),
 * [1] rank, [2] world, [3] ranks_seen: slots of the last sharded update that carried their sender's own number behind the block
 * (= world when every rank's block arrived), [4] 1 if a peer shares this rank's device (ipc), [5] 1 if the ipc gather buffer is
 * fine-grained / uncached memory, [6] 1 if a peer runs on ANOTHER device under the ipc transport: that case is UNVERIFIED (the transport
 * has only ever run with several ranks on one device) and comm_init refuses it unless ORCVIO_IPC_XDEV=1 is set. */
int32_t orcvio_msckf_comm_details(orcvio_msckf_handle* h, int32_t* out, int32_t count);
/* Device time of the three parts of orcvio_msckf_run_update_sharded on THIS rank (HIP events on the handle's stream, medians over
 * `reps` updates of the uploaded share), microseconds: us[0] local tracks + compression, us[1] the exchange (the RCCL all-gather, or
 * the ipc push + signal + wait; the wait for the slowest peer is part of it), us[2] rank-ordered sum + replicated solve, us[3] the
 * whole update.  COLLECTIVE: every rank calls it with the same reps.  What bench.py --gpus N reports beside DESIGN.md 5's model. */
int32_t orcvio_msckf_profile_sharded(orcvio_msckf_handle* h, int32_t reps, double* us /* [4] */);
/* Bounded waits: nothing here hangs on a rank that never arrives.  comm_unique_id / comm_init give up after ORCVIO_COMM_TIMEOUT_S
 * seconds (environment, default 180) and return ORCVIO_ERR_TIMEOUT; so does every call below that waits for a stream carrying a
 * collective (comm_barrier, comm_allreduce_max, the one-shot sharded updates, orcvio_msckf_sync on a handle with a communicator) --
 * the communicator is then ABORTED (comm_info reports world = 0) and must be created again by all ranks.
 *   comm_barrier         everything enqueued on the handle's stream before is finished on EVERY rank when it returns
 *   comm_allreduce_max   values[i] <- max over the ranks, count <= 8 (e.g. the slowest rank's time of a timed region)
 * so that a process needs no second communicator (and no second RCCL stream) beside the handle's for its bookkeeping. */
int32_t orcvio_msckf_comm_barrier(orcvio_msckf_handle* h);
int32_t orcvio_msckf_comm_allreduce_max(orcvio_msckf_handle* h, double* values, int32_t count);
/* The sharded feature update.  run_update_sharded: staged form on the tracks of the last orcvio_msckf_upload (this rank's
 * share), results stay in HBM (orcvio_msckf_download fetches them).  update_features_sharded: host buffers in and out like
 * orcvio_msckf_update_features; `tracks` are THIS RANK's tracks, result->accept / gamma are theirs, result->dx / P_out are
 * the joint update's (identical on every rank); stats[0], [2] count this rank's accepted rows / tracks.
 * Every rank must make the same call in the same order (it contains a collective).
 * Errors and the collective: the window and the prior are replicated, so a refusal they cause (null argument, leg_dim, capacity of
 * the window, P == NULL without a matching resident covariance) hits every rank alike BEFORE the collective and every rank
 * returns it.  A refusal only THIS rank's share can cause (capacity, a track longer than ORCVIO_MAX_TRACK, an index out of range)
 * does not leave the others waiting: the rank takes part with an empty share and a status word behind its block, every rank
 * finishes the collective, the failing rank returns its own status, all others ORCVIO_ERR_PEER, and no rank has an update to
 * commit.  A HIP / RCCL failure between the local part and the collective (ORCVIO_ERR_HIP) cannot be repaired that way: the
 * other ranks run into the bounded wait (ORCVIO_ERR_TIMEOUT, communicator aborted).  An in-launch time-out AFTER the collective
 * (ORCVIO_ERR_TIMEOUT from download) is rank-local: that rank has no update, the others do -- treat it as fatal for the joint
 * filter. */
int32_t orcvio_msckf_run_update_sharded(orcvio_msckf_handle* h, void* stream);
int32_t orcvio_msckf_update_features_sharded(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags,
                                             const orcvio_msckf_window* window, const orcvio_msckf_tracks* tracks,
                                             const double* P, orcvio_msckf_result* result);
/* The sharded object update (System::processObjects): this rank's object tracks -> local block -> all-gather of the blocks
 * and of the local degrees of freedom -> joint gate with the total dof (src/orcvio.cpp:2172-2176) -> replicated solve. */
int32_t orcvio_msckf_update_object_tracks_sharded(orcvio_msckf_handle* h, const orcvio_msckf_flags* flags,
                                                  const orcvio_object_eval_flags* eval_flags, int32_t n_clones,
                                                  const orcvio_object_track* tracks, int32_t n_tracks, const double* P,
                                                  orcvio_msckf_result* result);

/* ---- Feature triangulation --------------------------------------
 * Replaces, for every listed track, Feature::checkMotion followed by Feature::initializePosition
 * (include/orcvio/feat/feature.hpp:354-449; the Levenberg-Marquardt of ::triangulate_position, :583-719, with
 * ::generateInitialGuess :332-352, ::cost :270-290, ::jacobian :292-330) as called from
 * OrcVIO::removeLostFeatures (src/orcvio.cpp:2258-2270).  The tracks list the observations to use, in the order of
 * the reference's std::map (ascending clone); the caller has already dropped the current frame (curr_id).
 * The camera poses are the cached orientation_cam / position_cam of src/orcvio.cpp:954-961, formed from the window. */
typedef struct orcvio_triangulation_config { /* Feature::OptimizationConfig, feature.hpp:41-63 */
    double translation_threshold;     /* 0.2      */
    double huber_epsilon;             /* 0.01     */
    double estimation_precision;      /* 5e-7     */
    double initial_damping;           /* 1e-3     */
    int32_t outer_loop_max_iteration; /* 10       */
    int32_t inner_loop_max_iteration; /* 10       */
    double cost_threshold;            /* 4.7673e-4 */
    double init_final_dist_threshold; /* 5        */
} orcvio_triangulation_config;
void orcvio_msckf_triangulation_config_default(orcvio_triangulation_config* cfg);

enum { ORCVIO_TRI_NO_MOTION = 1, ORCVIO_TRI_NEG_DEPTH = 2, ORCVIO_TRI_BIG_PROJ = 4 };
typedef struct orcvio_triangulation_result { /* every pointer may be NULL */
    int32_t* valid;    /* [F] is_valid_solution */
    double* p_w;       /* [F][3] Feature::position (NaN where invalid and not initialised before) */
    double* inv_param; /* [F][3] Feature::invParam = (alpha, beta, rho) in the anchor (last listed) camera frame */
    int32_t* flags;    /* [F] ORCVIO_TRI_* bits: why a track is invalid (failed_by_neg_dpth, failed_by_big_proj) */
    double* cost;      /* [F] total squared reprojection error at the solution */
} orcvio_triangulation_result;
/* is_initialized: NULL, or [F]; where non-zero the track starts from tracks->p_w (feature.hpp:604-606) and the motion
 * check is skipped.  tracks->p_w may be NULL when is_initialized is NULL. */
int32_t orcvio_msckf_triangulate(orcvio_msckf_handle* h, const orcvio_triangulation_config* cfg,
                                 const orcvio_msckf_window* window, const orcvio_msckf_tracks* tracks,
                                 const int32_t* is_initialized, orcvio_triangulation_result* result);
/* The same on the tracks of the last orcvio_msckf_upload, in place on the device: valid tracks get their triangulated
 * position, the others are marked and take no part in the following orcvio_msckf_run_update (as the reference drops them
 * into invalid_feature_ids, src/orcvio.cpp:2262-2268).  Inputs never leave HBM between the two calls. */
int32_t orcvio_msckf_triangulate_uploaded(orcvio_msckf_handle* h, const orcvio_triangulation_config* cfg,
                                          const int32_t* is_initialized, void* stream);

/* ---- Device-resident covariance ------------------------------------------
 * The three places besides the update where the reference touches state_cov, on a copy of P that stays in HBM, so that
 * P does not cross PCIe every frame.  n = leg_dim + 6 * (clones in the window); no EKF-SLAM / nuisance states.
 *   cov_set / cov_get     host <-> resident P (n x n, row-major)
 *   cov_propagate         OrcVIO::processModel, src/orcvio.cpp:800-816: P_LL <- Phi P_LL Phi^T + Q (leg x leg blocks,
 *                         row-major), cross terms, symmetrised
 *   cov_augment           OrcVIO::stateAugmentation, :962-1010: n -> n + 6, the new clone copies the IMU (theta, p) covariance
 *   cov_remove_clones     OrcVIO::pruneImuStateBuffer, :2935-2951 (non-Schmidt branch): rows/cols of the listed window
 *                         indices (ascending rank in the window) are deleted
 *   cov_commit            resident P <- P+ of the update that has just run (features: always defined; objects: P if rejected)
 *   cov_prefactor         factor the resident P now, asynchronously (ORCVIO_OPT_RESIDENT_FACTOR): after cov_propagate no
 *                         square-root factor of P is known and the next update would start with the Cholesky of its prior;
 *                         called when the image arrives (behind propagate + augment), it moves that off the update's critical
 *                         path -- the update then finds the factor resident, as the later updates of a frame do.  No-op when
 *                         the factor is known or the window is too large for the register-resident factorisation (n > 224).
 * orcvio_msckf_upload / orcvio_msckf_update_features / orcvio_msckf_objects_local accept P == NULL: the resident P is
 * used (its dimension must match the window). */
int32_t orcvio_msckf_cov_set(orcvio_msckf_handle* h, int32_t n, const double* P);
int32_t orcvio_msckf_cov_get(orcvio_msckf_handle* h, int32_t* n_out, double* P_out /* may be NULL */);
int32_t orcvio_msckf_cov_propagate(orcvio_msckf_handle* h, int32_t leg_dim, const double* Phi, const double* Q);
int32_t orcvio_msckf_cov_augment(orcvio_msckf_handle* h);
int32_t orcvio_msckf_cov_remove_clones(orcvio_msckf_handle* h, int32_t leg_dim, const int32_t* clone_indices, int32_t count);
int32_t orcvio_msckf_cov_commit(orcvio_msckf_handle* h);
int32_t orcvio_msckf_cov_prefactor(orcvio_msckf_handle* h);
/* The tail of measurementUpdate_hybrid on the device (src/orcvio.cpp:1818-1821, :1904-1947; with ORCVIO_OPT_SCHMIDT_STATES :1920-1935):
 * after an update that carried entering features (orcvio_msckf_upload_new_features) the resident covariance becomes the augmented
 * one -- P+ with the d k new feature states behind it (in front of the nuisance block) -- computed from H_1, H_2, r_1 still on the
 * device; dx_new [d k] is returned.  Instead of orcvio_msckf_cov_commit for such an update.  The caller raises
 * ORCVIO_OPT_EXTRA_STATES by d k before its next upload. */
int32_t orcvio_msckf_cov_commit_new_features(orcvio_msckf_handle* h, double* dx_new);
/* Schmidt branch of pruneImuStateBuffer (src/orcvio.cpp:2881-2920): the listed clones (window ranks before the call, ascending)
 * leave the window but stay in the resident covariance as nuisance states -- their blocks move to the end, in the listed order. */
int32_t orcvio_msckf_cov_clones_to_nuisance(orcvio_msckf_handle* h, int32_t leg_dim, const int32_t* clone_indices, int32_t count);

/* ---- Environment switches (read once per process; diagnostics and A/B measurements -- every one of them leaves the results unchanged) ----
 *   ORCVIO_COMM_TIMEOUT_S   bound of every wait another rank can strand, seconds (default 180)
 *   ORCVIO_IPC_WAIT_S       ipc transport: bound of the device-side wait for a peer's block, seconds (default 20, at most the above)
 *   ORCVIO_IPC_COARSE       1: ipc transport's gather buffer in plain (coarse-grained) hipMalloc memory instead of fine-grained / uncached
 *   ORCVIO_RCCL_LIB         path of the RCCL library to dlopen first (then librccl.so.1 / librccl.so by the loader's search -- an already
 *                           loaded one, e.g. torch's bundled copy, wins --, then /opt/rocm/lib)
 *   ORCVIO_FRAME_OVERLAP    0: orcvio_msckf_io_update_frame runs its two halves one behind the other
 *   ORCVIO_FRAME_CHAIN      0 (read at create; the ONE switch of this list that changes results, in the last bits): the frame call's object
 *                           solve runs BEHIND the feature half, on the covariance and factor it commits -- bit-identical to the two calls.
 *                           Default 1 (windows from six block steps): the object solve is chained to the FEATURE update's prior factor and
 *                           M (M12 = M1 + L_a^T A' L_a: the same sequential update, by Woodbury) and runs on the objects' stream beside the
 *                           feature half's solve and commit -- 7 % of the config-3 frame; equal to the two calls to rounding (1e-10;
 *                           DESIGN.md 3.6), counted in orcvio_msckf_counters [6]
 *   ORCVIO_FRONT_U          1 (read at create): U = [A; b^T] L_a by the feature workgroups of k_front behind their Grams instead of the
 *                           k_gemm_asmA launch (bit-identical; measured 6 us slower per update: docs/LAB_NOTES.md); default 0
 *   ORCVIO_FUSE_FINISH      (read at create) P+ = s2 Z^T Z, dx and an object update's gate by finish workgroups of the factorisation + solve
 *                           launch instead of a k_finish_sqrt launch behind it (bit-identical either way): 1 (default) in the chained frame
 *                           call, both halves (0.167-0.170 -> 0.160 ms); 2 in every update whose solve takes the look-ahead form (measured
 *                           ~2 us slower per queued update); 0 never
 *   ORCVIO_FRAME_GRAPH      1: the frame call's feature half as a replayed launch graph (default: plain launches, with the objects'
 *                           compression enqueued in the middle of them: a graph's completion marker delays the object solve by ~14 us)
 *   ORCVIO_FRAME_EVENT_JOIN 1: the frame call's object solve joins the compression's stream with an event (default: its first product
 *                           polls the compression's completion word: a stream-level join costs ~10 us of dispatch even when long satisfied)
 *   ORCVIO_OBJ_FUSED        0: object tracks always through the three-launch compression over materialised rows (default 1: the one-launch
 *                           compression k_obj_fused whenever every track qualifies, orcvio_msckf_counters [5])
 *   ORCVIO_FUSED_STAMPS     phase stamps of k_obj_fused (object 0) on stderr
 *   ORCVIO_FUSED_TOL        k_obj_fused's pivot tolerance relative to the largest pivot (default 1e-10; tests of its verification step)
 *   ORCVIO_EARLY_INGEST     0: the whole arena is pulled by the ingest node of the launch graph (no early pull under the validation)
 *   ORCVIO_REV_PRIOR        0: plain Cholesky of the prior (M keeps its 15 IMU columns)
 *   ORCVIO_SPLIT_TRACKS     track count from which the tracks front end is two launches over E scratch in HBM (k_feature_e + k_feature_gate);
 *                           default 0 = never (round 5: no faster than k_feature at 2 000 tracks, 85 MB of scratch traffic per update)
 *   ORCVIO_FUSED_FRONT, ORCVIO_FUSED_SOLVE   0: the forked seven-launch front end / the two-launch solve (same as the options)
 *   ORCVIO_BLK2             0 (read at create): windows beyond the register-resident factorisations (n > 224: 34 .. 60 clones) through the LDS-panel
 *                           Cholesky and k_trsm_rl (0.95-2.1 ms per update) instead of the 2 x 2 block factorisation out of the register kernels
 *                           (0.26-0.41 ms); same results to rounding
 *   ORCVIO_LA_SOLVE         0 / 2 / 3: default of ORCVIO_OPT_LOOKAHEAD_SOLVE;  ORCVIO_LA_SPIN: polls before a wait inside k_potrf_solve_la gives up
 *                           (default 4 M, seconds; 0 makes every hand-off fail at once: the test of the fall-back)
 *   ORCVIO_FRONT_SPIN, ORCVIO_IO_SPIN_SECONDS   bounds of the in-launch hand-off of k_front (polls) and of the host's flag spin
 *   ORCVIO_OBJ_INGEST, ORCVIO_OBJ_PUBLISH   object update: 0 = copy engine instead of the ingest kernel; 1 = results through the flag word
 *   ORCVIO_TIMING           host wall times of the parts of the one-shot calls on stderr
 *   ORCVIO_ASM_DBG, ORCVIO_POTRF_ABLATE, ORCVIO_POTRF_COLD   kernel diagnostics (scripts/gpu_*.py) */

#ifdef __cplusplus
}
#endif
#endif /* ORCVIO_MSCKF_H */

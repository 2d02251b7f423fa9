"""Synthetic sliding-window inputs for the MSCKF update path.

Produces the flat (SoA / CSR) buffers the C-ABI in ``include/orcvio_msckf.h``
consumes.  The recipe is the one written down in SURVEY.md Appendix A /
§8(d): a smooth 6-DoF trajectory, point features in the frustum of the middle
clone, normalised-coordinate observations with pixel noise, a noisy world
position standing in for the triangulation step, and a dense SPD prior
covariance with the extrinsic / time-offset rows zeroed (reference
``config/euroc.yaml``: estimate_extrin = estimate_td = 0).

Nothing here touches the oracle; it is input generation only.
"""
from __future__ import annotations

import dataclasses
import numpy as np

LEG_DIM = 22  # reference src/orcvio.cpp:196-199 (no IMU-intrinsic calibration)

# Kalibr T_cam_imu of reference config/euroc.yaml:29-37 (body -> camera)
_T_CAM_IMU_EUROC = np.array(
    [[0.014865542981794, 0.999557249008346, -0.025774436697440, 0.065222909535531],
     [-0.999880929698575, 0.014967213324719, 0.003756188357967, -0.020706385492719],
     [0.004140296794224, 0.025715529947966, 0.999660727177902, -0.008054602460030],
     [0.0, 0.0, 0.0, 1.0]])


def so3_exp(w: np.ndarray) -> np.ndarray:
    th = float(np.linalg.norm(w))
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-10:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def euroc_extrinsics():
    """(R_b2c, t_c_b) exactly as reference src/orcvio.cpp:232-246 derives them."""
    T_inv = np.linalg.inv(_T_CAM_IMU_EUROC)
    R_b2c = T_inv[:3, :3].T
    t_c_b = T_inv[:3, 3].copy()
    return R_b2c, t_c_b


@dataclasses.dataclass
class Flags:
    """Switches that change hot-path arithmetic (SURVEY.md §5 'Config / flags')."""
    use_larvio: int = 1          # config/euroc.yaml:118
    use_left_perturbation: int = 0
    if_fej: int = 0
    estimate_td: int = 0
    leg_dim: int = LEG_DIM
    noise_feature: float = 0.008  # sigma, squared inside (src/orcvio.cpp:106,113)
    chi2_prob: float = 0.95
    discard_large_update: int = 0


@dataclasses.dataclass
class Window:
    """One update's worth of flat inputs (all float64, C-contiguous)."""
    R_b2w: np.ndarray      # [N,3,3] row-major
    t_b_w: np.ndarray      # [N,3]
    t_fej: np.ndarray      # [N,3]
    R_b2c: np.ndarray      # [N,3,3]
    t_c_b: np.ndarray      # [N,3]
    p_w: np.ndarray        # [F,3]
    obs_ptr: np.ndarray    # [F+1] int32, CSR
    obs_clone: np.ndarray  # [nobs] int32, ascending within a feature
    obs_z: np.ndarray      # [nobs,2]
    obs_zvel: np.ndarray   # [nobs,2]
    P: np.ndarray          # [n,n] symmetric
    flags: Flags
    n_extra: int = 0       # state columns behind the clones that no row touches (EKF-SLAM feature states)
    nui: dict = None       # Schmidt nuisance states (the LAST 6 each of the extra states): their poses, dict(R_b2w [k,3,3], t_b_w,
                           # t_fej, R_b2c, t_c_b) -- clones that left the window but stay in the covariance (use_schmidt)

    @property
    def n_nui(self):
        return 0 if self.nui is None else self.nui['R_b2w'].shape[0]

    @property
    def N(self):
        return self.R_b2w.shape[0]

    @property
    def F(self):
        return self.p_w.shape[0]

    @property
    def n(self):
        return self.flags.leg_dim + 6 * self.N + self.n_extra


def make_prior_cov(N: int, rng: np.random.Generator, leg_dim: int = LEG_DIM,
                   estimate_extrin: bool = False, estimate_td: bool = False) -> np.ndarray:
    """P = 1e-4*A*A^T + diag(initial covariances of config/euroc.yaml:76-82)."""
    n = leg_dim + 6 * N
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    P = 1e-4 * (A @ A.T)
    d = np.zeros(n)
    d[0:3] = 4e-4
    d[3:6] = 0.25
    d[6:9] = 1.0
    d[9:12] = 4e-4
    d[12:15] = 0.01
    d[15:18] = 3.0462e-8
    d[18:21] = 9e-8
    d[21] = 4e-6
    for i in range(N):
        d[leg_dim + 6 * i: leg_dim + 6 * i + 3] = 1e-3
        d[leg_dim + 6 * i + 3: leg_dim + 6 * i + 6] = 1e-2
    P += np.diag(d)
    lo = 15 if not estimate_extrin else 21
    hi = 22 if not estimate_td else 21
    if hi > lo:
        P[lo:hi, :] = 0.0
        P[:, lo:hi] = 0.0
    if leg_dim > 22:
        pass  # IMU-intrinsic block keeps its generic SPD values
    return 0.5 * (P + P.T)


def with_extra_states(win: "Window", k: int, seed: int = 0) -> "Window":
    """The same window with k more states behind the clones (inverse-depth feature states of the hybrid filter,
    src/orcvio.cpp:1495-1510): a new SPD prior of the larger size whose clone / IMU block keeps the structure of
    make_prior_cov (zero rows for the states that are not estimated) and whose cross terms are dense."""
    rng = np.random.default_rng(10_000 + seed)
    n0 = win.n - win.n_extra
    n = n0 + k
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    P = 1e-4 * (A @ A.T)
    P[:n0, :n0] += np.diag(np.diag(win.P))
    P[n0:, n0:] += np.diag(np.full(k, 2.5e-3))   # ~ (0.05 1/m)^2 inverse-depth variance
    zero = np.where(np.diag(win.P) == 0.0)[0]
    P[zero, :] = 0.0
    P[:, zero] = 0.0
    return dataclasses.replace(win, P=np.ascontiguousarray(0.5 * (P + P.T)), n_extra=k)


def with_nuisance_states(win: "Window", n_nui: int, seed: int = 0) -> "Window":
    """The same window with n_nui Schmidt nuisance states behind everything else (src/orcvio.cpp:2881-2920: clones that left
    the window but stay in state_cov): 6 n_nui more columns at the END of the extra states, correlated with the rest, and
    their poses (older than the window: the first clone's pose moved back along the trajectory)."""
    rng = np.random.default_rng(40_000 + seed)
    n0 = win.n
    k = 6 * n_nui
    n = n0 + k
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    P = 1e-5 * (A @ A.T)
    P[:n0, :n0] += win.P
    P[n0:, n0:] += np.diag(np.tile([4e-4] * 3 + [1e-2] * 3, n_nui))
    zero = np.where(np.diag(win.P) == 0.0)[0]
    P[zero, :] = 0.0
    P[:, zero] = 0.0
    step = win.t_b_w[0] - win.t_b_w[min(1, win.N - 1)]
    nui = dict(R_b2w=np.ascontiguousarray(np.stack([win.R_b2w[0]] * n_nui)),
               t_b_w=np.ascontiguousarray(np.stack([win.t_b_w[0] + (j + 1) * step + 0.02 * rng.standard_normal(3) for j in range(n_nui)])),
               R_b2c=np.ascontiguousarray(np.stack([win.R_b2c[0]] * n_nui)), t_c_b=np.ascontiguousarray(np.stack([win.t_c_b[0]] * n_nui)))
    nui['t_fej'] = np.ascontiguousarray(nui['t_b_w'] + 0.005 * rng.standard_normal((n_nui, 3)))
    return dataclasses.replace(win, P=np.ascontiguousarray(0.5 * (P + P.T)), n_extra=win.n_extra + k, nui=nui)


@dataclasses.dataclass
class SlamFeature:
    """One EKF-SLAM feature of the hybrid filter as the reference's Feature holds it (anchor clone, inverse-depth
    parametrisation in the anchor camera frame, world position) and its observation in the current state."""
    anchor: int
    state: int
    inv_param: np.ndarray
    obs_anchor: np.ndarray
    inv_depth: float
    p_w: np.ndarray
    z: np.ndarray
    z_vel: np.ndarray
    p_fej: np.ndarray = None


def make_slam_features(win: "Window", n_feat: int, seed: int = 0, outlier_frac: float = 0.0, nui_frac: float = 0.0,
                       sigma_px: float | None = None):
    """n_feat SLAM features anchored at random earlier clones and observed by the newest one (+ pixel noise; a
    fraction with gross errors that the 2-dof gate rejects).  nui_frac: fraction anchored at a Schmidt nuisance state
    (anchor index N + j, win.nui).  sigma_px: the observation noise if it is not the filter's noise_feature (kitti_raw.yaml sets
    noise_feature 1 on normalised coordinates)."""
    rng = np.random.default_rng(20_000 + seed)
    k = win.N - 1
    sig = win.flags.noise_feature if sigma_px is None else sigma_px
    out = []
    for _ in range(n_feat):
        a = int(rng.integers(0, win.N - 1))
        Ra, ta, Rbc, tcb = win.R_b2w[a], win.t_b_w[a], win.R_b2c[a], win.t_c_b[a]
        if win.n_nui > 0 and rng.random() < nui_frac:
            j = int(rng.integers(0, win.n_nui))
            a = win.N + j
            Ra, ta, Rbc, tcb = win.nui['R_b2w'][j], win.nui['t_b_w'][j], win.nui['R_b2c'][j], win.nui['t_c_b'][j]
        R_c2w = Ra @ Rbc.T
        t_c_w = ta + Ra @ tcb
        pc = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.uniform(4.0, 12.0)])
        pw = R_c2w @ pc + t_c_w
        inv = np.array([pc[0] / pc[2], pc[1] / pc[2], 1.0 / pc[2]])
        Rk = win.R_b2c[k] @ win.R_b2w[k].T
        tk = win.t_b_w[k] + win.R_b2w[k] @ win.t_c_b[k]
        pk = Rk @ (pw - tk)
        noise = sig * (40.0 if rng.random() < outlier_frac else 1.0)
        z = pk[:2] / pk[2] + noise * rng.standard_normal(2)
        out.append(SlamFeature(anchor=a, state=k, inv_param=inv, obs_anchor=np.array([inv[0], inv[1], 1.0]),
                               inv_depth=float(inv[2]), p_w=pw + 0.01 * rng.standard_normal(3), z=z,
                               z_vel=0.05 * rng.standard_normal(2), p_fej=pw + 0.01 * rng.standard_normal(3)))
    return out


def make_new_slam_features(win: "Window", n_feat: int, seed: int = 0, outlier_frac: float = 0.0):
    """Features about to ENTER the state of the hybrid filter: anchored at the first clone of a contiguous run of 3..7
    observations (oracle.mirror_hybrid.NewSlamFeature-shaped dicts: anchor, inv_param, obs_anchor, inv_depth, p_w, obs)."""
    rng = np.random.default_rng(30_000 + seed)
    sig = win.flags.noise_feature
    out = []
    for _ in range(n_feat):
        M = int(rng.integers(3, min(win.N, 7) + 1))
        start = int(rng.integers(0, win.N - M + 1))
        a = start
        R_c2w = win.R_b2w[a] @ win.R_b2c[a].T
        t_c_w = win.t_b_w[a] + win.R_b2w[a] @ win.t_c_b[a]
        pc = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.uniform(4.0, 12.0)])
        pw = R_c2w @ pc + t_c_w
        inv = np.array([pc[0] / pc[2], pc[1] / pc[2], 1.0 / pc[2]])
        noise = sig * (12.0 if rng.random() < outlier_frac else 1.0)
        obs = []
        for k in range(start, start + M):
            Rk = win.R_b2c[k] @ win.R_b2w[k].T
            tk = win.t_b_w[k] + win.R_b2w[k] @ win.t_c_b[k]
            pk = Rk @ (pw - tk)
            obs.append((k, pk[:2] / pk[2] + noise * rng.standard_normal(2), 0.05 * rng.standard_normal(2)))
        # the estimate the filter would hold after triangulation: perturbed in the anchor camera frame, world position
        # and inverse-depth parameters consistent with each other (Feature::initializePosition sets both from one solution)
        pce = pc + 0.01 * rng.standard_normal(3)
        inve = np.array([pce[0] / pce[2], pce[1] / pce[2], 1.0 / pce[2]])
        out.append(dict(anchor=a, inv_param=inve, obs_anchor=np.array([inve[0], inve[1], 1.0]), inv_depth=float(inve[2]),
                        p_w=R_c2w @ pce + t_c_w, obs=obs))
    return out


def make_window(N: int = 30, F: int = 400, seed: int = 0, track_len=None,
                flags: Flags | None = None, estimate_extrin: bool = False,
                sigma_px: float | None = None, outlier_frac: float = 0.0, depth=(4.0, 12.0)) -> Window:
    """SURVEY.md Appendix A synthetic generator.

    track_len: None -> every feature seen in all N clones; int M -> contiguous
    run of M clones at a random start; (lo, hi) -> ragged M_j in [lo, hi].
    outlier_frac: fraction of tracks with 12x observation noise, which the
    chi-square gate rejects (exercises the accept mask).
    """
    flags = flags or Flags()
    rng = np.random.default_rng(seed)
    sig = flags.noise_feature if sigma_px is None else sigma_px
    R_b2c0, t_c_b0 = euroc_extrinsics()

    R_b2w = np.empty((N, 3, 3))
    t_b_w = np.empty((N, 3))
    for i in range(N):
        R_b2w[i] = so3_exp(np.array([0.02 * i, 0.01 * np.sin(i), 0.015 * i]))
        t_b_w[i] = [0.15 * i, 0.05 * np.sin(0.3 * i), 0.03 * i]
    t_fej = t_b_w + 1e-3 * rng.standard_normal((N, 3))
    R_b2c = np.broadcast_to(R_b2c0, (N, 3, 3)).copy()
    t_c_b = np.broadcast_to(t_c_b0, (N, 3)).copy()

    # camera poses
    R_c2w = np.einsum('nij,nkj->nik', R_b2w, R_b2c)            # R_b2w * R_b2c^T
    t_c_w = t_b_w + np.einsum('nij,nj->ni', R_b2w, t_c_b)

    mid = N // 2
    depth = rng.uniform(depth[0], depth[1], F)   # (far features: tiny parallax, a nearly rank-deficient H_f)
    xy = rng.uniform(-0.35, 0.35, (F, 2))
    p_c_mid = np.stack([xy[:, 0] * depth, xy[:, 1] * depth, depth], axis=1)
    p_true = p_c_mid @ R_c2w[mid].T + t_c_w[mid]
    p_w = p_true + 0.02 * rng.standard_normal((F, 3))
    bad = np.zeros(F, dtype=bool)
    if outlier_frac > 0:
        # mismatched tracks: observation noise far above noise_feature, so that the
        # chi-square gate (src/orcvio.cpp:1953-1976) rejects the block
        bad = rng.random(F) < outlier_frac

    obs_ptr = [0]
    obs_clone, obs_z, obs_zvel = [], [], []
    for j in range(F):
        if track_len is None:
            ids = np.arange(N)
        else:
            if isinstance(track_len, (tuple, list)):
                M = int(rng.integers(track_len[0], track_len[1] + 1))
            else:
                M = int(track_len)
            M = min(M, N)
            s = int(rng.integers(0, N - M + 1))
            ids = np.arange(s, s + M)
        for i in ids:
            pc = R_c2w[i].T @ (p_true[j] - t_c_w[i])
            z = pc[:2] / pc[2] + (12.0 * sig if bad[j] else sig) * rng.standard_normal(2)
            obs_clone.append(i)
            obs_z.append(z)
            obs_zvel.append(0.05 * rng.standard_normal(2))
        obs_ptr.append(len(obs_clone))

    P = make_prior_cov(N, rng, flags.leg_dim, estimate_extrin, bool(flags.estimate_td))
    return Window(
        R_b2w=np.ascontiguousarray(R_b2w), t_b_w=np.ascontiguousarray(t_b_w),
        t_fej=np.ascontiguousarray(t_fej), R_b2c=R_b2c, t_c_b=t_c_b,
        p_w=np.ascontiguousarray(p_w),
        obs_ptr=np.asarray(obs_ptr, dtype=np.int32),
        obs_clone=np.asarray(obs_clone, dtype=np.int32),
        obs_z=np.asarray(obs_z, dtype=np.float64).reshape(-1, 2),
        obs_zvel=np.asarray(obs_zvel, dtype=np.float64).reshape(-1, 2),
        P=np.ascontiguousarray(P), flags=flags)


def config_window(config: int, seed: int = 0) -> Window:
    """BASELINE.json configs (feature part)."""
    if config == 1:   # euroc.yaml shape: N=20, tracks 3..6, LARVIO Jacobians
        return make_window(N=20, F=120, seed=seed, track_len=(3, 6), flags=Flags(use_larvio=1))
    if config == 2:   # the metric's configuration
        return make_window(N=30, F=400, seed=seed, flags=Flags(use_larvio=1))
    if config == 3:   # feature half of config 3 (objects are added by synth_objects)
        return make_window(N=30, F=400, seed=seed, flags=Flags(use_larvio=0, use_left_perturbation=0))
    if config == 4:
        return make_window(N=30, F=2000, seed=seed, flags=Flags(use_larvio=1))
    if config == 5:   # kitti_raw.yaml flags: OrcVIO right perturbation, sigma=1, discard on
        return make_window(N=30, F=2000, seed=seed, sigma_px=0.008,
                           flags=Flags(use_larvio=0, use_left_perturbation=0,
                                       noise_feature=1.0, discard_large_update=1))
    raise ValueError(config)


# ---------------------------------------------------------------------------------------------
# objects (BASELINE.json config 3): cars with 12 semantic keypoints and a bounding box per frame
# ---------------------------------------------------------------------------------------------
# reference config/object_feat_unity.yaml:6-19 (class "car")
CAR_KEYPOINTS_MEAN = np.array(
    [[-0.568, 0.568, 0.482, -0.482, -0.582, 0.582, 0.702, -0.702, -0.805, -0.805, 0.805, 0.805],
     [-0.253, -0.253, 1.570, 1.570, -1.988, -1.988, 1.961, 1.961, -1.286, 1.355, -1.286, 1.355],
     [1.331, 1.331, 1.331, 1.331, 0.702, 0.702, 0.924, 0.924, 0.329, 0.329, 0.329, 0.329]]).T
CAR_MEAN_SHAPE = np.array([1.6, 3.9, 1.0])


@dataclasses.dataclass
class ObjectTrack:
    """One object at a fixed state (GT + noise: stands in for the LM optimum) and its observations."""
    wTo: np.ndarray      # [4,4] object -> world
    shape: np.ndarray    # [3] ellipsoid semi-axes
    kps: np.ndarray      # [K,3] keypoints in the object frame
    frames: list         # per frame: dict(clone=int, wTc=[4,4], zs=[K,2] (NaN rows = not detected), bbox=[4])


def camera_poses(win: Window) -> np.ndarray:
    """wTc (camera -> world) of every clone, as src/orcvio.cpp:954-961."""
    T = np.tile(np.eye(4), (win.N, 1, 1))
    for i in range(win.N):
        T[i, :3, :3] = win.R_b2w[i] @ win.R_b2c[i].T
        T[i, :3, 3] = win.t_b_w[i] + win.R_b2w[i] @ win.t_c_b[i]
    return T


def make_objects(win: Window, n_objects: int = 20, seed: int = 0, missing_frac: float = 0.1,
                 frames_per_object=None, sigma_kp: float | None = None, bbox_only: bool = False) -> list:
    """bbox_only: object tracks WITHOUT keypoints (object state [pose 6 | shape 3], four bbox rows per frame) -- BASELINE config 5's
    "bbox-only OrcVIO-lite" read literally; an extension, the reference's lite mode sends no residuals at all (SURVEY note N4)."""
    rng = np.random.default_rng(seed + 7919)
    sig = win.flags.noise_feature if sigma_kp is None else sigma_kp
    wTc = camera_poses(win)
    mid = win.N // 2
    objs = []
    for o in range(n_objects):
        # pose: in front of the middle camera, random yaw, car "up" roughly along -y of the camera
        depth = rng.uniform(8.0, 20.0)
        pc = np.array([rng.uniform(-0.3, 0.3) * depth, rng.uniform(-0.1, 0.2) * depth, depth, 1.0])
        pw = wTc[mid] @ pc
        yaw = rng.uniform(-np.pi, np.pi)
        Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
        Rx = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0.0]])      # object z (up) -> camera -y
        T = np.eye(4)
        T[:3, :3] = wTc[mid][:3, :3] @ Rx @ Rz
        T[:3, 3] = pw[:3]
        kps = CAR_KEYPOINTS_MEAN + 0.03 * rng.standard_normal(CAR_KEYPOINTS_MEAN.shape)
        shape = CAR_MEAN_SHAPE * (1 + 0.05 * rng.standard_normal(3))
        # estimate = truth + small noise (the state the residuals are evaluated at)
        dxi = 0.01 * rng.standard_normal(6)
        T_est = T.copy()
        T_est[:3, 3] += dxi[:3]
        T_est[:3, :3] = T[:3, :3] @ so3_exp(dxi[3:])
        kps_est = kps + 0.01 * rng.standard_normal(kps.shape)
        ids = range(win.N) if frames_per_object is None else sorted(rng.choice(win.N, frames_per_object, replace=False))
        frames = []
        for i in ids:
            cTw = np.linalg.inv(wTc[i])
            Xc = (cTw @ T @ np.hstack([kps, np.ones((len(kps), 1))]).T).T
            uv = Xc[:, :2] / Xc[:, 2:3]
            zs = uv + sig * rng.standard_normal(uv.shape)
            miss = rng.random(len(kps)) < missing_frac
            zs[miss] = np.nan
            # box around the projected ellipsoid corners
            cor = np.array([[sx * shape[0], sy * shape[1], sz * shape[2], 1.0] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
            Cc = (cTw @ T @ cor.T).T
            cuv = Cc[:, :2] / Cc[:, 2:3]
            bbox = np.array([cuv[:, 0].min(), cuv[:, 1].min(), cuv[:, 0].max(), cuv[:, 1].max()]) + sig * rng.standard_normal(4)
            frames.append(dict(clone=int(i), wTc=wTc[i].copy(), zs=zs, bbox=bbox))
        if bbox_only:
            kps_est = np.zeros((0, 3))
            for fr in frames:
                fr['zs'] = np.zeros((0, 2))
        objs.append(ObjectTrack(wTo=T_est, shape=shape, kps=kps_est, frames=frames))
    return objs


# ---- streams of filter frames (bench.py's stream legs, tests/cpp/stream_bench.cpp) ----------------------------------------------
def subset_tracks(win: "Window", clone_ids, min_obs: int = 0) -> "Window":
    """CSR restricted to the observations of `clone_ids` (what pruneImuStateBuffer lists, reference src/orcvio.cpp:2810-2845);
    min_obs: tracks with fewer listed observations are emptied (the reference uses features seen in BOTH clones that leave)."""
    keep = np.isin(win.obs_clone, np.asarray(clone_ids))
    csum = np.concatenate([[0], np.cumsum(keep.astype(np.int64))])
    cnt = csum[win.obs_ptr[1:]] - csum[win.obs_ptr[:-1]]
    use = cnt >= max(min_obs, 0)
    keep = keep & np.repeat(use, np.diff(win.obs_ptr))
    ptr = np.concatenate([[0], np.cumsum(np.where(use, cnt, 0))]).astype(np.int32)
    return dataclasses.replace(win, obs_ptr=ptr, obs_clone=win.obs_clone[keep].copy(), obs_z=win.obs_z[keep].copy(),
                               obs_zvel=win.obs_zvel[keep].copy())


def make_stream(flags: "Flags", sigma_px=None, cycle: int = 8, seed: int = 0, n_slam: int = 12, idp: int = 1, leg: int = LEG_DIM):
    """A cycle of pre-generated filter frames at the reference's shipped operating point (sw_size 20, max_track_len 6,
    max_features_in_one_grid 1 -> hybrid filter with `n_slam` in-state features of `idp` parameter(s); config/euroc.yaml:49-109,
    config/kitti_raw.yaml:77-148): 19 / 20 clones alternating, 20-200 lost features per frame with 3-6 observations each; the
    20-clone frames carry the prune update on the two oldest clones (features seen in both) and their marginalisation.  Returns
    (frames, P0): frames[k] = dict(w, slam, prune | None, Phi, Q, remove), P0 the 18-clone covariance the loop starts from."""
    rng = np.random.default_rng(seed)
    frames = []
    for k in range(cycle):
        N = 20 if k % 2 else 19
        F = int(rng.integers(20, 201))
        w0 = make_window(N=N, F=F, seed=1000 + k, track_len=(3, 6), flags=flags, outlier_frac=0.05, sigma_px=sigma_px)
        w = with_extra_states(w0, idp * n_slam, seed=k)
        slam = make_slam_features(w, n_slam, seed=k, outlier_frac=0.1, sigma_px=sigma_px)
        prune = None
        if N == 20:
            sub = subset_tracks(w, [0, 1], min_obs=2)
            if int(sub.obs_ptr[-1]) > 0:
                prune = sub
        Phi = np.eye(leg) + 0.002 * rng.standard_normal((leg, leg))
        G = rng.standard_normal((leg, 12))
        frames.append(dict(w=w, slam=slam, prune=prune, Phi=np.ascontiguousarray(Phi), Q=np.ascontiguousarray(1e-7 * G @ G.T),
                           remove=[0, 1] if N == 20 else []))
    P0 = with_extra_states(make_window(N=18, F=1, seed=5, flags=flags), idp * n_slam, seed=1).P
    return frames, np.ascontiguousarray(P0)


def pack_poses(win: "Window") -> np.ndarray:
    """[N][28] pose records of the input arena (include/orcvio_msckf.h ORCVIO_POSE_STRIDE)."""
    p = np.zeros((win.N, 28))
    p[:, 0:9] = win.R_b2w.reshape(win.N, 9); p[:, 9:12] = win.t_b_w; p[:, 12:15] = win.t_fej
    p[:, 15:24] = win.R_b2c.reshape(win.N, 9); p[:, 24:27] = win.t_c_b
    return p


def write_stream(path: str, frames, P0: np.ndarray, flags: "Flags", idp: int = 1):
    """The frames of make_stream as one little-endian binary file for tests/cpp/stream_bench.cpp (its reader mirrors this writer):
    'ORCSTRM2', header, P0, then per frame the window's pose records, the tracks, the in-state features, the prune tracks, Phi, Q."""
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32).tobytes()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64).tobytes()
    n_slam = len(frames[0]['slam'])
    with open(path, 'wb') as f:
        f.write(b'ORCSTRM2')
        f.write(i32([len(frames), P0.shape[0], idp, n_slam, flags.leg_dim, flags.use_larvio, flags.use_left_perturbation, flags.if_fej,
                     flags.estimate_td, flags.discard_large_update]))
        f.write(f64([flags.noise_feature, flags.chi2_prob]))
        f.write(f64(P0))
        for fr in frames:
            w, pr = fr['w'], fr['prune']
            nobs = int(w.obs_ptr[-1])
            F2 = pr.F if pr is not None else 0
            nobs2 = int(pr.obs_ptr[-1]) if pr is not None else 0
            rem = list(fr['remove'])
            f.write(i32([w.N, w.F, nobs, 1 if pr is not None else 0, F2, nobs2, len(rem)] + rem + [0] * (4 - len(rem))))
            f.write(f64(pack_poses(w))); f.write(f64(w.p_w)); f.write(i32(w.obs_ptr)); f.write(i32(w.obs_clone)); f.write(f64(w.obs_z))
            sl = fr['slam']
            f.write(i32([s.anchor for s in sl])); f.write(i32([s.state for s in sl])); f.write(i32(list(range(len(sl)))))
            f.write(f64([s.inv_param if idp == 3 else s.obs_anchor for s in sl])); f.write(f64([s.inv_depth for s in sl]))
            f.write(f64([s.p_w for s in sl])); f.write(f64([s.p_fej if s.p_fej is not None else s.p_w for s in sl]))
            f.write(f64([s.z for s in sl])); f.write(f64([s.z_vel for s in sl]))
            if pr is not None:
                f.write(f64(pr.p_w)); f.write(i32(pr.obs_ptr)); f.write(i32(pr.obs_clone)); f.write(f64(pr.obs_z))
            f.write(f64(fr['Phi'])); f.write(f64(fr['Q']))

"""numpy mirror of the reference MSCKF update arithmetic -- TEST INFRASTRUCTURE ONLY.

This is the literal (operation-for-operation, FP64) restatement of the
reference hot path, written with numpy/scipy.  It is used only as a checker:
by tests/, by scripts/make_golden.py (to generate tests/golden/*.npz) and to
validate the plain-C oracle in oracle/msckf_oracle.c.  Nothing under
orcvio_amd/ may import it.

PARITY STATUS.  The reference cannot be built in this image (Eigen, SPQR,
Sophus, Boost, OpenCV are absent) and its own test-suite holds
no vector for the feature-side arithmetic (rows 3,5,7-11 of SURVEY.md §8a), so
for those rows **parity is unpinned**: the anchor is this restatement plus
central-difference checks of every Jacobian variant.  The object rows
(12-16) ARE pinned: against the reference's HDF5 goldens converted to
tests/golden/ref_*.npz (see oracle/mirror_objects.py).

All file:line citations are relative to /root/reference.
"""
from __future__ import annotations

import numpy as np
from scipy.stats import chi2 as _chi2


# ----------------------------------------------------------------------------
# helpers: include/orcvio/utils/math_utils.hpp:27-39, se3_ops.hpp:510-552
# ----------------------------------------------------------------------------
def skew(w):
    """math_utils.hpp:27-39 skewSymmetric."""
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def odot(x4):
    """se3_ops.hpp:510-519 odotOperator: [x4*I3, -skew(x123); 0] (4x6)."""
    T = np.zeros((4, 6))
    T[:3, 3:] = -skew(x4[:3])
    T[0, 0] = T[1, 1] = T[2, 2] = x4[3]
    return T


def cam_wrt_imu_se3_jacobian(R_b2c, t_c_b, R_w2c, t_b_w, left):
    """se3_ops.hpp:531-552 get_cam_wrt_imu_se3_jacobian (6x6)."""
    J = np.zeros((6, 6))
    if left:
        J[0:3, 0:3] = skew(t_b_w)
        J[3:6, 0:3] = np.eye(3)
        J[0:3, 3:6] = np.eye(3)
    else:
        J[0:3, 0:3] = -R_b2c @ skew(t_c_b)
        J[3:6, 0:3] = R_b2c
        J[0:3, 3:6] = R_w2c
    return J


def chi2_table(prob=0.95, nmax=500):
    """src/orcvio.cpp:481-494: boost quantile(chi_squared(i), prob), i=1..499."""
    t = np.zeros(nmax)
    t[1:] = _chi2.ppf(prob, np.arange(1, nmax))
    return t


def chi2_threshold(dof, prob=0.95, table=None):
    """src/orcvio.cpp:1961-1968."""
    if table is not None and dof < len(table):
        return float(table[dof])
    return float(_chi2.ppf(prob, dof))


# ----------------------------------------------------------------------------
# src/orcvio.cpp:1071-1168  measurementJacobian_msckf
# ----------------------------------------------------------------------------
def measurement_jacobian_msckf(win, i, p_w, z):
    f = win.flags
    R_b2c = win.R_b2c[i]
    t_c_b = win.t_c_b[i]
    R_b2w = win.R_b2w[i]
    R_w2b = R_b2w.T
    t_b_w = win.t_b_w[i]
    R_w2c = R_b2c @ R_w2b                         # :1090
    t_c_w = t_b_w + R_b2w @ t_c_b                 # :1091
    p_c = R_w2c @ (p_w - t_c_w)                   # :1099-1100
    p_bf_w = (p_w - win.t_fej[i]) if f.if_fej else (p_w - t_b_w)   # :1104
    dz = np.zeros((2, 3))                         # :1107-1111
    dz[0, 0] = 1 / p_c[2]
    dz[1, 1] = 1 / p_c[2]
    dz[0, 2] = -p_c[0] / (p_c[2] * p_c[2])
    dz[1, 2] = -p_c[1] / (p_c[2] * p_c[2])
    if not f.use_larvio:                          # :1115-1143
        temp = np.zeros((3, 4))
        temp[:, :3] = np.eye(3)
        wTc = np.eye(4)
        wTc[:3, :3] = R_w2c.T
        wTc[:3, 3] = t_c_w
        ul = np.append(p_w, 1.0)
        D = cam_wrt_imu_se3_jacobian(R_b2c, t_c_b, R_w2c, t_b_w, f.use_left_perturbation)
        if f.use_left_perturbation:
            dpc = temp @ np.linalg.inv(wTc) @ odot(ul) @ D
        else:
            dpc = temp @ odot(np.linalg.inv(wTc) @ ul) @ D
        H_x = -dz @ dpc
    else:                                         # :1145-1149
        dpc = np.zeros((3, 6))
        dpc[:, :3] = R_w2c @ skew(p_bf_w)
        dpc[:, 3:] = -R_w2c
        H_x = dz @ dpc
    dpe = np.zeros((3, 6))                        # :1152-1155
    dpe[:, :3] = R_w2c @ skew(p_bf_w) @ R_b2w - R_b2c @ skew(t_c_b)
    dpe[:, 3:] = -R_b2c
    H_e = dz @ dpe                                # :1160
    H_f = dz @ R_w2c                              # :1161
    r = z - np.array([p_c[0] / p_c[2], p_c[1] / p_c[2]])   # :1165
    return H_x, H_e, H_f, r


# ----------------------------------------------------------------------------
# math_utils.hpp:287-312  nullspace_project_inplace_svd
# ----------------------------------------------------------------------------
def nullspace_project_svd(H_f, H_x, res):
    if H_f.shape[0] <= H_f.shape[1]:
        return False, H_x, res
    U, _, _ = np.linalg.svd(H_f, full_matrices=True)
    A = U[:, H_f.shape[1]:]
    return True, A.T @ H_x, A.T @ res


# ----------------------------------------------------------------------------
# src/orcvio.cpp:1171-1226  featureJacobian_msckf
# ----------------------------------------------------------------------------
def feature_jacobian_msckf(win, j, clone_subset=None, project=True):
    """Returns (H_xj [rows x n], r_j, H_fj) -- projected unless project=False."""
    f = win.flags
    n = win.n
    lo, hi = int(win.obs_ptr[j]), int(win.obs_ptr[j + 1])
    ks = [k for k in range(lo, hi)
          if clone_subset is None or int(win.obs_clone[k]) in clone_subset]
    rows = 2 * len(ks)
    H_xj = np.zeros((rows, n))
    H_fj = np.zeros((rows, 3))
    r_j = np.zeros(rows)
    c = 0
    for k in ks:
        i = int(win.obs_clone[k])
        H_x, H_e, H_f, r = measurement_jacobian_msckf(win, i, win.p_w[j], win.obs_z[k])
        H_xj[c:c + 2, f.leg_dim + 6 * i: f.leg_dim + 6 * i + 6] = H_x     # :1209
        H_xj[c:c + 2, 15:21] = H_e                                        # :1210
        if f.estimate_td:
            H_xj[c:c + 2, 21] = win.obs_zvel[k]                           # :1211-1212
        H_fj[c:c + 2] = H_f
        r_j[c:c + 2] = r
        c += 2
    if not project:
        return H_xj, r_j, H_fj
    _, H_xj, r_j = nullspace_project_svd(H_fj, H_xj, r_j)                 # :1220
    return H_xj, r_j, H_fj


# ----------------------------------------------------------------------------
# src/orcvio.cpp:1953-1976  gatingTestFeature
# ----------------------------------------------------------------------------
def gating_gamma(H, r, P, sigma2):
    S = H @ P @ H.T + sigma2 * np.eye(H.shape[0])
    return float(r @ np.linalg.solve(S, r))


# ----------------------------------------------------------------------------
# src/orcvio.cpp:2532-2552 / :1664-1679  SPQR compression (dense Householder QR
# is the same map up to a left-orthogonal factor; top `ncols` rows kept)
# ----------------------------------------------------------------------------
def qr_compress(H, r):
    """Top n rows of Q^T*[H | r] for a full (m x m) Householder Q equal the
    reduced factorisation's R and Q1^T r; the reduced form avoids building Q."""
    m, n = H.shape
    if m <= n:
        return H, r
    Q1, R = np.linalg.qr(H, mode='reduced')
    return R, Q1.T @ r


# ----------------------------------------------------------------------------
# src/orcvio.cpp:1654-1763 (and the pure-MSCKF case of :1766-1950)
# ----------------------------------------------------------------------------
def measurement_update(H_thin, r_thin, P, sigma2):
    S = H_thin @ P @ H_thin.T + sigma2 * np.eye(H_thin.shape[0])
    K_T = np.linalg.solve(S, H_thin @ P)          # S.ldlt().solve(H*P)
    K = K_T.T
    dx = K @ r_thin
    I_KH = np.eye(P.shape[0]) - K @ H_thin
    Pn = I_KH @ P
    Pn = (Pn + Pn.T) / 2.0
    return dx, K, Pn


def msckf_update(win, clone_subset=None, table=None):
    """removeLostFeatures stacking (:2497-2560) or, with clone_subset, the
    pruneImuStateBuffer variant (:2803-2851).  Returns a dict of everything
    the parity tests compare."""
    f = win.flags
    sigma2 = f.noise_feature ** 2
    table = chi2_table(f.chi2_prob) if table is None else table
    n = win.n
    blocks, rs, gammas, accept, dofs = [], [], [], [], []
    for j in range(win.F):
        lo, hi = int(win.obs_ptr[j]), int(win.obs_ptr[j + 1])
        M = sum(1 for k in range(lo, hi)
                if clone_subset is None or int(win.obs_clone[k]) in clone_subset)
        if M < 2:
            gammas.append(np.nan); accept.append(0); dofs.append(0)
            blocks.append(np.zeros((0, n))); rs.append(np.zeros(0))
            continue
        Hj, rj, _ = feature_jacobian_msckf(win, j, clone_subset)
        dof = 2 * M - 3
        g = gating_gamma(Hj, rj, win.P, sigma2)
        ok = g < chi2_threshold(dof, f.chi2_prob, table)
        gammas.append(g); accept.append(int(ok)); dofs.append(dof)
        blocks.append(Hj); rs.append(rj)
    acc_blocks = [b for b, a in zip(blocks, accept) if a]
    acc_rs = [b for b, a in zip(rs, accept) if a]
    out = dict(gamma=np.array(gammas), accept=np.array(accept, dtype=np.int32),
               dof=np.array(dofs, dtype=np.int32), blocks=blocks, rs=rs)
    if not acc_blocks:
        out.update(dx=np.zeros(n), P_new=win.P.copy(), updated=False,
                   G=np.zeros((n, n)), K=np.zeros((n, 0)), H_thin=np.zeros((0, n)),
                   r_thin=np.zeros(0))
        return out
    H = np.vstack(acc_blocks)
    r = np.concatenate(acc_rs)
    H_thin, r_thin = qr_compress(H, r)
    dx, K, Pn = measurement_update(H_thin, r_thin, win.P, sigma2)
    out.update(dx=dx, P_new=Pn, K=K, G=K @ H_thin, H_thin=H_thin, r_thin=r_thin,
               H=H, r=r, updated=True)
    return out


# ----------------------------------------------------------------------------
# src/orcvio.cpp:4468-4567 incrementState_IMUCam  (math_utils.hpp:104-121,164-177)
# ----------------------------------------------------------------------------
def so3_exp(w):
    """Sophus v1.0.0 SO3d::exp = Rodrigues (via unit quaternion)."""
    th = np.linalg.norm(w)
    K = skew(w)
    if th < 1e-10:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def small_angle_quaternion(dtheta):
    """math_utils.hpp:104-121, [x,y,z,w]."""
    dq = dtheta / 2.0
    q = np.zeros(4)
    n2 = dq @ dq
    if n2 <= 1:
        q[:3] = dq
        q[3] = np.sqrt(1 - n2)
    else:
        q[:3] = dq
        q[3] = 1
        q = q / np.sqrt(1 + n2)
    return q


def quat_to_rot_hamilton(q):
    """Eigen Quaterniond(w,x,y,z).toRotationMatrix()."""
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def increment_state(state, dx, flags):
    """state: dict with imu R_b2w,v,p,bg,ba,R_b2c,t_c_b,td and clones R_b2w[N],t_b_w[N].
    Returns (new_state, applied)."""
    leg = flags.leg_dim
    d = dx[:leg]
    if flags.discard_large_update and (np.linalg.norm(d[3:6]) > 1.0 or np.linalg.norm(d[6:9]) > 1.5):
        return state, False                                    # :4479-4494
    s = {k: np.array(v, copy=True) for k, v in state.items()}
    left = bool(flags.use_larvio or flags.use_left_perturbation)
    Rt = so3_exp(d[0:3])
    s['R_b2w_imu'] = Rt @ s['R_b2w_imu'] if left else s['R_b2w_imu'] @ Rt
    s['v'] = s['v'] + d[3:6]
    s['p'] = s['p'] + d[6:9]
    s['bg'] = s['bg'] + d[9:12]
    s['ba'] = s['ba'] + d[12:15]
    q = small_angle_quaternion(d[15:18])
    s['R_b2c'] = s['R_b2c'] @ quat_to_rot_hamilton(q).T
    s['t_c_b'] = s['t_c_b'] + d[18:21]
    s['td'] = s['td'] + d[21]
    N = s['R_b2w'].shape[0]
    s['R_c2w'] = np.zeros((N, 3, 3))
    s['t_c_w'] = np.zeros((N, 3))
    for i in range(N):
        da = dx[leg + 6 * i: leg + 6 * i + 6]
        Rt = so3_exp(da[:3])
        s['R_b2w'][i] = Rt @ s['R_b2w'][i] if left else s['R_b2w'][i] @ Rt
        s['t_b_w'][i] = s['t_b_w'][i] + da[3:]
        s['R_c2w'][i] = s['R_b2w'][i] @ s['R_b2c'].T           # :4555-4564
        s['t_c_w'][i] = s['t_b_w'][i] + s['R_b2w'][i] @ s['t_c_b']
    return s, True

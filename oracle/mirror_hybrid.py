"""TEST INFRASTRUCTURE ONLY -- literal numpy restatement of the EKF-SLAM rows of the reference's hybrid filter and of
the joint update with them (existing SLAM features; the H_1 / H_2 initialisation of new ones is not restated here).

Follows src/orcvio.cpp:
  measurementJacobian_ekf_3didp   :1229-1353
  measurementJacobian_ekf_1didp   :1356-1478
  featureJacobian_ekf             :1575-1651  (one observation, the current IMU state; anchor inside the window)
  removeLostFeatures, EKF part    :2444-2464  (gatingTestFeature(H_xj, r_j, 2) per SLAM feature, then stacking)
  measurementUpdate_hybrid        :1766-1950  (case sz_new == 0: H_o = [H_msckf; H_ekf], K, dx, (I - K H) P)
State layout: [legacy LEG | clones 6N | feature states d each | Schmidt nuisance states 6 each] (:1495-1510).  Schmidt
(use_schmidt): a window with win.nui set carries the poses of the nuisance states; an anchor index N + j is nuisance state j
(:1247-1256 its pose, :1591-1606 its columns), the update leaves the nuisance block of P alone (:1740-1751, :1893-1902) and new
feature states are inserted in front of it (:1920-1935).
Parity unpinned: the reference has no test of these functions; anchored on its text and on central differences
(tests/test_oracle_hybrid.py).  Nothing under orcvio_amd/ may import this module.
"""
import dataclasses

import numpy as np

from . import mirror
from .mirror import skew


@dataclasses.dataclass
class SlamFeature:
    """One EKF-SLAM feature as Feature holds it (include/orcvio/feat/feature.hpp): anchor clone, inverse-depth
    parametrisation in the anchor CAMERA frame, world position, and its observation in the current state."""
    anchor: int            # index of the anchor clone in the window (id_anchor)
    state: int             # index of the observing clone (state_server.imu_state.id: the newest clone)
    inv_param: np.ndarray  # [3] (x/z, y/z, 1/z) in the anchor camera frame            (3-d idp: invParam)
    obs_anchor: np.ndarray # [3] (u, v, 1) corrected observation in the anchor frame    (1-d idp: obs_anchor)
    inv_depth: float       #                                                             (1-d idp: invDepth)
    p_w: np.ndarray        # [3] Feature::position
    z: np.ndarray          # [2] observation in the current state
    z_vel: np.ndarray      # [2] observations_vel (used under estimate_td)
    p_fej: np.ndarray = None   # position_FEJ (if_FEJ)


def _n_nui(win):
    return getattr(win, 'n_nui', 0)


def _anchor_pose(win, a):
    """(R_b2w, t_b_w, t_fej) of anchor a: a clone of the window, or (a >= N) nuisance state a - N (:1247-1256)."""
    if a < win.N:
        return win.R_b2w[a], win.t_b_w[a], win.t_fej[a]
    j = a - win.N
    return win.nui['R_b2w'][j], win.nui['t_b_w'][j], win.nui['t_fej'][j]


def anchor_col(win, a):
    """first column of anchor a's 6 x 6 block: LEG + 6 a, or inside the nuisance block at the end of the state (:1591-1606)"""
    if a < win.N:
        return win.flags.leg_dim + 6 * a
    return win.n - 6 * _n_nui(win) + 6 * (a - win.N)


def _poses(win, k, a):
    R_b2c, t_c_b = win.R_b2c[k], win.t_c_b[k]
    R_w2bk = win.R_b2w[k].T
    t_bk_w = win.t_b_w[k]
    R_w2ck = R_b2c @ R_w2bk
    t_ck_w = t_bk_w + win.R_b2w[k] @ t_c_b
    Ra, ta, _ = _anchor_pose(win, a)
    R_w2ba = Ra.T
    t_ba_w = ta
    R_w2ca = R_b2c @ R_w2ba
    return R_b2c, t_c_b, R_w2bk, t_bk_w, R_w2ck, t_ck_w, R_w2ba, t_ba_w, R_w2ca


def measurement_jacobian_ekf(win, ft: SlamFeature, idp_dim: int):
    """(H_f [2,d], H_a [2,6], H_x [2,6], H_e [2,6], r [2]) -- :1229-1353 (d = 3) / :1356-1478 (d = 1)."""
    f = win.flags
    k, a = ft.state, ft.anchor
    R_b2c, t_c_b, R_w2bk, t_bk_w, R_w2ck, t_ck_w, R_w2ba, t_ba_w, R_w2ca = _poses(win, k, a)
    fej = bool(f.if_fej)
    p_fej = ft.p_fej if ft.p_fej is not None else ft.p_w
    if fej:                                                     # :1281-1282 / :1410-1411
        p_ca = R_b2c @ (R_w2ba @ (p_fej - _anchor_pose(win, a)[2]) - t_c_b)
    elif idp_dim == 3:
        p_ca = np.array([ft.inv_param[0] / ft.inv_param[2], ft.inv_param[1] / ft.inv_param[2], 1.0 / ft.inv_param[2]])
    else:
        p_ca = np.array([ft.obs_anchor[0] / ft.inv_depth, ft.obs_anchor[1] / ft.inv_depth, 1.0 / ft.inv_depth])
    p_ck = R_w2ck @ (ft.p_w - t_ck_w)                           # :1295-1296
    r = ft.z - np.array([p_ck[0] / p_ck[2], p_ck[1] / p_ck[2]])   # :1299
    d = idp_dim
    if k == a:                                                  # :1302-1310 / :1432-1440
        H_f = np.zeros((2, d))
        if d == 3:
            H_f[0, 0] = 1.0
            H_f[1, 1] = 1.0
        else:
            r = np.zeros(2)
        return H_f, np.zeros((2, 6)), np.zeros((2, 6)), np.zeros((2, 6)), r
    J_k = np.zeros((2, 3))                                      # :1312-1316
    J_k[0, 0] = 1 / p_ck[2]
    J_k[1, 1] = 1 / p_ck[2]
    J_k[0, 2] = -p_ck[0] / (p_ck[2] * p_ck[2])
    J_k[1, 2] = -p_ck[1] / (p_ck[2] * p_ck[2])
    p_baf_w = (p_fej - _anchor_pose(win, a)[2]) if fej else (ft.p_w - t_ba_w)   # :1320-1323
    p_bkf_w = (p_fej - win.t_fej[k]) if fej else (ft.p_w - t_bk_w)
    J_xa = np.zeros((3, 6))                                     # :1325-1327
    J_xa[:, :3] = -R_w2ck @ skew(p_baf_w)
    J_xa[:, 3:] = R_w2ck
    J_xk = np.zeros((3, 6))                                     # :1329-1331
    J_xk[:, :3] = R_w2ck @ skew(p_bkf_w)
    J_xk[:, 3:] = -R_w2ck
    J_e = np.zeros((3, 6))                                      # :1333-1337
    SkewMx = skew(R_w2bk @ p_bkf_w - t_c_b)
    Mx = R_w2bk @ R_w2ba.T @ skew(R_b2c.T @ p_ca)
    J_e[:, :3] = R_b2c @ (SkewMx - Mx)
    J_e[:, 3:] = R_b2c @ (R_w2bk @ R_w2ba.T - np.eye(3))
    if d == 3:                                                  # :1318, :1339-1345
        J_p = R_w2ck @ R_w2ca.T
        fc = ft.inv_param
        J_f = np.eye(3)
        J_f[0, 2] = -fc[0] / fc[2]
        J_f[1, 2] = -fc[1] / fc[2]
        J_f[2, 2] = -1 / fc[2]
        J_f = J_f / fc[2]
        H_f = J_k @ J_p @ J_f
    else:                                                       # :1447, :1469-1471
        J_d = R_w2ck @ R_w2ca.T @ ft.obs_anchor
        J_rho = -1 / (ft.inv_depth * ft.inv_depth)
        H_f = (J_k @ J_d * J_rho).reshape(2, 1)
    return H_f, J_k @ J_xa, J_k @ J_xk, J_k @ J_e, r


def feature_jacobian_ekf(win, ft: SlamFeature, idx: int, idp_dim: int):
    """:1575-1651 -- the two rows of SLAM feature number idx over the whole state (width win.n)."""
    f = win.flags
    H = np.zeros((2, win.n))
    H_f, H_a, H_x, H_e, r = measurement_jacobian_ekf(win, ft, idp_dim)
    fi = f.leg_dim + 6 * win.N + idp_dim * idx                  # :1612
    H[:, fi:fi + idp_dim] = H_f                                 # :1630 / :1636
    H[:, anchor_col(win, ft.anchor): anchor_col(win, ft.anchor) + 6] = H_a   # :1639 (assignment order as the reference:
    H[:, f.leg_dim + 6 * ft.state: f.leg_dim + 6 * ft.state + 6] = H_x     # :1640  a later block overwrites an earlier one)
    H[:, 15:21] = H_e                                           # :1641
    if f.estimate_td:
        H[:, 21] = ft.z_vel                                     # :1642-1643
    return H, r


def hybrid_update(win, slam, idp_dim: int, table=None):
    """removeLostFeatures with SLAM features in the state and none being initialised (:2444-2560): the MSCKF blocks as
    mirror.msckf_update builds and gates them, the SLAM rows gated one feature at a time with 2 degrees of freedom,
    one update with everything that passed."""
    f = win.flags
    sigma2 = f.noise_feature ** 2
    table = mirror.chi2_table(f.chi2_prob) if table is None else table
    base = mirror.msckf_update(win, table=table)
    acc_blocks = [b for b, a in zip(base['blocks'], base['accept']) if a]
    acc_rs = [b for b, a in zip(base['rs'], base['accept']) if a]
    eg, ea, erows = [], [], []
    for idx, ft in enumerate(slam):
        H, r = feature_jacobian_ekf(win, ft, idx, idp_dim)
        g = mirror.gating_gamma(H, r, win.P, sigma2)            # :2457 gatingTestFeature(H_xj, r_j, 2)
        ok = g < mirror.chi2_threshold(2, f.chi2_prob, table)
        eg.append(g); ea.append(int(ok)); erows.append((H, r))
        if ok:
            acc_blocks.append(H); acc_rs.append(r)
    out = dict(gamma=base['gamma'], accept=base['accept'], ekf_gamma=np.array(eg), ekf_accept=np.array(ea, dtype=np.int32),
               ekf_rows=erows)
    n = win.n
    if not acc_blocks:
        out.update(dx=np.zeros(n), P_new=win.P.copy(), G=np.zeros((n, n)))
        return out
    H = np.vstack(acc_blocks)
    r = np.concatenate(acc_rs)
    H_thin, r_thin = mirror.qr_compress(H, r)
    dx, K, Pn = mirror.measurement_update(H_thin, r_thin, win.P, sigma2)
    Pn = keep_nuisance_block(win, Pn)
    out.update(dx=dx, P_new=Pn, G=K @ H_thin)
    return out


def keep_nuisance_block(win, P_upd):
    """:1740-1751 / :1893-1902 -- Schmidt: the nuisance block of the updated covariance is the prior's."""
    m = 6 * _n_nui(win)
    if m == 0:
        return P_upd
    P = P_upd.copy()
    P[-m:, -m:] = win.P[-m:, -m:]
    return 0.5 * (P + P.T)


# ----------------------------------------------------------------------------------------------------------------
# New SLAM features: featureJacobian_ekf_new (:1481-1572), the W = [V | U] split (:2337-2436) and the H_1 / H_2 part of
# measurementUpdate_hybrid (:1766-1947).  Without Schmidt nuisance states.
# ----------------------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class NewSlamFeature:
    """A feature about to enter the state: anchor, parametrisation, position and ALL its observations."""
    anchor: int
    inv_param: np.ndarray
    obs_anchor: np.ndarray
    inv_depth: float
    p_w: np.ndarray
    obs: list               # [(clone index, z [2], z_vel [2])], ascending clone index
    p_fej: np.ndarray = None


def feature_jacobian_ekf_new(win, ft: NewSlamFeature, idx_new: int, n_new: int, idp_dim: int):
    """:1481-1572 -- rows of every listed observation over [state (win.n) | new feature states (d n_new)]."""
    f = win.flags
    d = idp_dim
    obs = [(k, z, zv) for (k, z, zv) in ft.obs if not (d == 1 and k == ft.anchor)]   # :1494-1496
    H = np.zeros((2 * len(obs), win.n + d * n_new))
    r = np.zeros(2 * len(obs))
    fi = win.n + d * idx_new                                                          # :1533 (no nuisance states)
    for c, (k, z, zv) in enumerate(obs):
        one = SlamFeature(anchor=ft.anchor, state=k, inv_param=ft.inv_param, obs_anchor=ft.obs_anchor, inv_depth=ft.inv_depth,
                          p_w=ft.p_w, z=z, z_vel=zv, p_fej=ft.p_fej)
        H_f, H_a, H_x, H_e, rr = measurement_jacobian_ekf(win, one, d)
        H[2 * c:2 * c + 2, fi:fi + d] = H_f                                           # :1550 / :1557
        H[2 * c:2 * c + 2, anchor_col(win, ft.anchor): anchor_col(win, ft.anchor) + 6] = H_a   # :1561
        H[2 * c:2 * c + 2, f.leg_dim + 6 * k: f.leg_dim + 6 * k + 6] = H_x            # :1562
        H[2 * c:2 * c + 2, 15:21] = H_e                                               # :1563
        if f.estimate_td:
            H[2 * c:2 * c + 2, 21] = zv                                               # :1564-1565
        r[2 * c:2 * c + 2] = rr
    return H, r


def msckf_gate_of_feature(win, ft: NewSlamFeature, table=None):
    """:2361-2367 -- the new feature is judged by the MSCKF test of its track: featureJacobian_msckf over all its
    observations, gatingTestFeature with 2M - 3 degrees of freedom."""
    f = win.flags
    table = mirror.chi2_table(f.chi2_prob) if table is None else table
    M = len(ft.obs)
    Hx = np.zeros((2 * M, win.n)); Hf = np.zeros((2 * M, 3)); r = np.zeros(2 * M)
    for c, (k, z, zv) in enumerate(ft.obs):
        H_x, H_e, H_f, rr = mirror.measurement_jacobian_msckf(win, k, ft.p_w, z)
        Hx[2 * c:2 * c + 2, f.leg_dim + 6 * k: f.leg_dim + 6 * k + 6] = H_x
        Hx[2 * c:2 * c + 2, 15:21] = H_e
        if f.estimate_td:
            Hx[2 * c:2 * c + 2, 21] = zv
        Hf[2 * c:2 * c + 2] = H_f
        r[2 * c:2 * c + 2] = rr
    _, Hp, rp = mirror.nullspace_project_svd(Hf, Hx, r)
    g = mirror.gating_gamma(Hp, rp, win.P, f.noise_feature ** 2)
    return g, bool(g < mirror.chi2_threshold(2 * M - 3, f.chi2_prob, table))


def split_new_rows(win, new_feats, idp_dim: int, table=None):
    """:2337-2436 -- stack the rows of the new features that pass, then rotate them with W = [V | U] (V: left null space
    of the new-feature columns, U: their column space): returns (accepted indices, H_top [rows - sz, n], r_top, H_1 [sz, n],
    H_2 [sz, sz], r_1)."""
    d = idp_dim
    acc = [i for i, ft in enumerate(new_feats) if msckf_gate_of_feature(win, ft, table)[1]]
    n_new = len(acc)
    if n_new == 0:
        z = np.zeros
        return acc, z((0, win.n)), z(0), z((0, win.n)), z((0, 0)), z(0)
    Hs, rs = [], []
    for j, i in enumerate(acc):
        H, r = feature_jacobian_ekf_new(win, new_feats[i], j, n_new, d)
        Hs.append(H); rs.append(r)
    H = np.vstack(Hs); r = np.concatenate(rs)
    sz = d * n_new
    Q, _ = np.linalg.qr(H[:, win.n:], mode='complete')       # U = first sz columns, V = the rest (any orthonormal bases
    W = np.hstack([Q[:, sz:], Q[:, :sz]])                    #  of the two subspaces give the same update)
    Hn = W.T @ H; rn = W.T @ r
    m = H.shape[0]
    return acc, Hn[:m - sz, :win.n], rn[:m - sz], Hn[m - sz:, :win.n], Hn[m - sz:, win.n:], rn[m - sz:]


def augment_after_update(P_upd, dx_leg, H_1, H_2, r_1, sigma2, nui_rows=0, ref_ldlt=False):
    """:1811-1821 and :1904-1947: the new states' correction and the augmented covariance from the UPDATED legacy covariance.
    This part stays with the caller in the integration.  nui_rows > 0 (Schmidt): the new states are
    inserted in front of the trailing nuisance rows / columns, block by block as :1920-1935 does.
    ref_ldlt: the reference's LITERAL `H_2.ldlt().solve(..)` (:1826-1827).  H_2 is upper triangular (the R of the new features'
    H_f) and Eigen's LDLT<MatrixXd, Lower> reads the lower triangle only: it factors diag(H_2).  False: the triangular system is
    solved, which is what the derivation (and :1907-1908's own H_2^T H_2) means.  Equal for feature_idp_dim = 1."""
    n = P_upd.shape[0]
    sz = H_2.shape[0]
    if sz == 0:
        return dx_leg.copy(), P_upd.copy()
    if ref_ldlt:
        H_2s = np.diag(np.diag(np.tril(H_2)))   # what the LDLT of the lower triangle of an upper-triangular matrix sees
    else:
        H_2s = H_2
    if nui_rows > 0:
        HH = np.linalg.solve(H_2s, H_1)
        dx_new = -HH @ dx_leg + np.linalg.solve(H_2s, r_1)
        nHHP = -HH @ P_upd
        P22 = -nHHP @ HH.T + sigma2 * np.linalg.inv(H_2.T @ H_2)
        old_rows = old_cols = n
        P = np.zeros((n + sz, n + sz))
        P[:n, :n] = P_upd                                                        # conservativeResize
        nr = nc = nui_rows
        P[old_rows + sz - nr:, :old_cols] = P[old_rows - nr:old_rows, :old_cols].copy()                 # :1924-1925
        P[:old_rows + sz, old_cols + sz - nc:] = P[:old_rows + sz, old_cols - nc:old_cols].copy()       # :1926-1927
        P[old_rows - nr:old_rows - nr + sz, :old_cols - nc] = nHHP[:, :old_cols - nc]                   # :1929
        P[old_rows - nr:old_rows - nr + sz, old_cols + sz - nc:] = nHHP[:, old_cols - nc:]              # :1930
        P[:old_rows - nr, old_cols - nc:old_cols - nc + sz] = nHHP[:, :old_cols - nc].T                 # :1931
        P[old_rows + sz - nr:, old_cols - nc:old_cols - nc + sz] = nHHP[:, old_cols - nc:].T            # :1932
        P[old_rows - nr:old_rows - nr + sz, old_cols - nc:old_cols - nc + sz] = P22                     # :1933
        return np.concatenate([dx_leg, dx_new]), 0.5 * (P + P.T)
    HH = np.linalg.solve(H_2s, H_1)
    dx_new = -HH @ dx_leg + np.linalg.solve(H_2s, r_1)
    nHHP = -HH @ P_upd
    P22 = -nHHP @ HH.T + sigma2 * np.linalg.inv(H_2.T @ H_2)
    P = np.zeros((n + sz, n + sz))
    P[:n, :n] = P_upd
    P[n:, :n] = nHHP
    P[:n, n:] = nHHP.T
    P[n:, n:] = P22
    return np.concatenate([dx_leg, dx_new]), 0.5 * (P + P.T)


def hybrid_update_full(win, slam, new_feats, idp_dim: int, table=None, ref_ldlt=False):
    """removeLostFeatures + measurementUpdate_hybrid with MSCKF tracks, existing SLAM features and new ones."""
    f = win.flags
    sigma2 = f.noise_feature ** 2
    table = mirror.chi2_table(f.chi2_prob) if table is None else table
    base = mirror.msckf_update(win, table=table)
    blocks = [b for b, a in zip(base['blocks'], base['accept']) if a]
    rs = [b for b, a in zip(base['rs'], base['accept']) if a]
    ea = []
    for idx, ft in enumerate(slam):
        H, r = feature_jacobian_ekf(win, ft, idx, idp_dim)
        ok = mirror.gating_gamma(H, r, win.P, sigma2) < mirror.chi2_threshold(2, f.chi2_prob, table)
        ea.append(int(ok))
        if ok:
            blocks.append(H); rs.append(r)
    acc, H_top, r_top, H_1, H_2, r_1 = split_new_rows(win, new_feats, idp_dim, table)
    if H_top.shape[0]:
        blocks.append(H_top); rs.append(r_top)
    # (no row at all in the top stack -- nothing passed its gate and every entering feature has exactly idp_dim rows, or none
    #  enters: the reference goes on with a 0-row H_o, :1777-1818: dx_leg = 0, P unchanged; with sz_r == 0 it returns, :1775)
    H_o = np.vstack(blocks) if blocks else np.zeros((0, win.n)); r_o = np.concatenate(rs) if rs else np.zeros(0)
    P = win.P
    S = H_o @ P @ H_o.T + sigma2 * np.eye(H_o.shape[0])                  # :1811-1815
    K = np.linalg.solve(S, H_o @ P).T
    dx_leg = K @ r_o
    P_upd = (np.eye(win.n) - K @ H_o) @ P                                # :1889-1902
    P_upd = keep_nuisance_block(win, P_upd)
    P_upd = 0.5 * (P_upd + P_upd.T)
    dx, P_full = augment_after_update(P_upd, dx_leg, H_1, H_2, r_1, sigma2, 6 * _n_nui(win), ref_ldlt=ref_ldlt)
    return dict(dx=dx, P_new=P_full, accept=base['accept'], ekf_accept=np.array(ea, dtype=np.int32), new_accept=acc,
                dx_leg=dx_leg, P_upd=P_upd)

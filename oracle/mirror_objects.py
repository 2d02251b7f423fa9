"""numpy mirror of the reference OBJECT residual rows and their filter update -- TEST INFRASTRUCTURE ONLY.

Restates, operation for operation (FP64), SURVEY.md 8a rows 12-17:
  rows 12-14  CameraLM / ObjectLM residual + Jacobian functors evaluated at a fixed state
              (src/obj/ObjectResJacCam.cpp:153-519, src/obj/ObjectLM.cpp:250-632,
               include/orcvio/utils/se3_ops.hpp:229-240,325-453)
  row 15      the export block of single_levenberg_marquardt (src/obj/ObjectFeatureInitializer.cpp:394-434)
  row 16      OrcVIO::constructObjectResidualJacobians (src/orcvio.cpp:2017-2151)
  row 17      OrcVIO::removeLostObjects (src/orcvio.cpp:2154-2193)

PARITY STATUS: pinned for the keypoint rows and the *old* bbox rows against the reference's own
HDF5 goldens (tests/golden/ref_test_error_*.npz, converted by scripts/convert_ref_h5.py); the
*new* bbox residual has no reference test (SURVEY.md note N8) and is restated literally, including
its Jacobian's use of the world-frame plane.  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

from . import mirror


# ---------------------------------------------------------------------------------------------
# se3_ops.hpp helpers
# ---------------------------------------------------------------------------------------------
def project_image_df(x):
    """se3_ops.hpp:325-339."""
    z = x[2]
    return np.array([[1 / z, 0, -x[0] / (z * z)], [0, 1 / z, -x[1] / (z * z)]])


def project_object_points(P, wTo, points_w):
    """se3_ops.hpp:351-355: uv = pi(P (wTo X)) for homogeneous object points X [K][4], P 3 x 4."""
    uvh = np.asarray(P) @ (np.asarray(wTo) @ np.asarray(points_w).T)
    return (uvh[:2] / uvh[2:3]).T


def circled_circ(x4):
    """se3_ops.hpp:229-240: 6x4, rows 3:6 cols 0:3 = -skew(x[:3]), rows 0:3 col 3 = x[:3]."""
    T = np.zeros((6, 4))
    T[3:6, 0:3] = -mirror.skew(x4[:3])
    T[0:3, 3] = x4[:3]
    return T


def bbox2poly(b):
    """ObjectLM.cpp:380-392 (xmin, ymin, xmax, ymax)."""
    return np.array([[b[0], b[1]], [b[2], b[1]], [b[2], b[3]], [b[0], b[3]]])


def poly2lineh(pts):
    """ObjectLM.cpp:394-405."""
    L = np.zeros((len(pts), 3))
    for i in range(len(pts)):
        a = np.array([pts[i, 0], pts[i, 1], 1.0])
        b = np.array([pts[(i + 1) % len(pts), 0], pts[(i + 1) % len(pts), 1], 1.0])
        L[i] = np.cross(a, b)
    return L


def ellipse_from_shape(v):
    """ObjectLM.cpp:407-414."""
    return np.diag([v[0] ** 2, v[1] ** 2, v[2] ** 2, -1.0])


def se3_exp(xi):
    """Sophus v1.0.0 SE3d::exp, tangent = (upsilon, omega)."""
    u, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    W = mirror.skew(w)
    R = mirror.so3_exp(w)
    if th < 1e-10:
        V = np.eye(3) + 0.5 * W + W @ W / 6.0
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ u
    return T


def se3_log(T):
    """Sophus v1.0.0 SE3d::log."""
    R = T[:3, :3]
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-10:
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    else:
        w = th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    W = mirror.skew(w)
    if th < 1e-10:
        Vinv = np.eye(3) - 0.5 * W + W @ W / 12.0
    else:
        Vinv = np.eye(3) - 0.5 * W + (1 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * W @ W
    return np.concatenate([Vinv @ T[:3, 3], w])


# ---------------------------------------------------------------------------------------------
# rows 12-15: residual rows of one object over its frames
# ---------------------------------------------------------------------------------------------
def keypoint_rows(cTw, wTo, kps, zs, left):
    """One frame.  Returns (res [2v], J_cam [2v x 6], Hf_pose [2v x 6], Hf_kp list[(kpid, 2x3)], valid ids).
    ObjectResJacCam.cpp:153-282, ObjectLM.cpp:272-346, se3_ops.hpp:351-453."""
    P = cTw[:3, :]
    valid = [i for i in range(len(zs)) if np.all(np.isfinite(zs[i]))]     # ObjectLM.cpp:171-198
    res, Jc, Hp, Hk = [], [], [], []
    for i in valid:
        X = np.append(kps[i], 1.0)
        Mc = P @ wTo @ X
        dpi = project_image_df(Mc)
        res.append(Mc[:2] / Mc[2] - zs[i])
        sel = np.hstack([np.eye(3), np.zeros((3, 1))])
        if left:
            Jc.append(-dpi @ sel @ cTw @ mirror.odot(wTo @ X))            # se3_ops.hpp:431-436
            Hp.append(dpi @ P @ mirror.odot(wTo @ X))                       # :386-389
        else:
            Jc.append(-dpi @ sel @ mirror.odot(cTw @ wTo @ X))             # :440-443
            Hp.append(dpi @ P @ wTo @ mirror.odot(X))                       # :391-394
        Hk.append((i, dpi @ P @ wTo[:, :3]))                                # ObjectLM.cpp:336-341
    return res, Jc, Hp, Hk, valid


def bbox_rows(cTw, wTo, shape, bbox, left, new_residual):
    """One frame, 4 rows.  Returns (res [4], J_cam [4x6], Hf_pose [4x6], Hf_shape [4x3]).
    ObjectResJacCam.cpp:308-494, ObjectLM.cpp:441-616."""
    Qi = ellipse_from_shape(shape)
    lines = poly2lineh(bbox2poly(bbox))
    P_res = (cTw @ wTo)[:3, :]          # residual: K * (cTw * wTo)
    P = cTw[:3, :]                      # Jacobians: K * cTw
    P_prime = np.eye(4)[:3, :]
    wTc_invT = cTw.T                    # object.get_wTc(f).inverse().transpose()
    res = np.zeros(4)
    Jc = np.zeros((4, 6))
    Hp = np.zeros((4, 6))
    Hs = np.zeros((4, 3))
    U2 = Qi[:3, :3]
    for i in range(4):
        li = lines[i]
        if not new_residual:
            Ci = P_res @ Qi @ P_res.T
            res[i] = li @ Ci @ li
        else:
            ub = P_res.T @ li
            b = ub[:3]
            sign = 1.0 if ub[3] > 0 else -1.0
            res[i] = (ub[3] - sign * np.sqrt(b @ U2 @ b)) / np.linalg.norm(b)
        yyw = li @ P                    # 1x4
        yyw_prime = li @ P_prime
        yyo = yyw @ wTo
        if not new_residual:
            if left:
                p_eb_p_oxi = 2 * yyo @ Qi @ wTo.T @ circled_circ(yyw).T
                Jc[i] = -p_eb_p_oxi                                                   # ResJacCam.cpp:425-430
                Hp[i] = p_eb_p_oxi                                                    # ObjectLM.cpp:527-533
            else:
                Jc[i] = -2 * yyo @ Qi @ wTo.T @ wTc_invT @ circled_circ(yyw_prime).T  # :436
                Hp[i] = 2 * yyo @ Qi @ circled_circ(wTo.T @ yyw).T                    # ObjectLM.cpp:540-541
            Hs[i] = 2 * shape * (yyo[:3] ** 2)                                        # ObjectLM.cpp:546-548
        else:
            # new_residual = True / 1: the reference's Jacobians LITERALLY -- plane in the WORLD frame (ResJacCam.cpp:405,446,
            # ObjectLM.cpp:512,560) although the residual uses the object frame, shape Jacobian without -sign(b4) (note N8).
            # new_residual = 2: the opt-in CORRECTED Jacobians -- plane in the object frame, sign restored (central differences
            # agree to 1e-9: tests/test_oracle_objects.py)
            corrected = int(new_residual) == 2
            ub = P_res.T @ li if corrected else P.T @ li
            b = ub[:3]
            bn = np.linalg.norm(b)
            if left:
                dO = wTo.T @ circled_circ(yyw).T
                dC = dO
            else:
                dO = circled_circ(wTo.T @ yyw).T
                dC = wTo.T @ wTc_invT @ circled_circ(yyw_prime).T
            term1a = np.array([0, 0, 0, 1.0])
            term2a = Qi.copy()
            term2a[3, 3] = 0
            sign = 1.0 if ub[3] > 0 else -1.0
            sq = np.sqrt(b @ U2 @ b)
            p_be_p_ua = term1a - sign * (ub @ term2a) / sq
            term2b = np.eye(4)
            term2b[3, 3] = 0
            p_ua_ub = np.eye(4) / bn - np.outer(ub, ub) @ term2b / bn ** 3
            Jc[i] = -p_be_p_ua @ p_ua_ub @ dC                                         # ResJacCam.cpp:487
            Hp[i] = p_be_p_ua @ p_ua_ub @ dO                                          # ObjectLM.cpp:598-599
            Hs[i] = shape * b * b / (bn * sq) * (-sign if corrected else 1.0)         # ObjectLM.cpp:602-603 (no sign there)
    return res, Jc, Hp, Hs


def object_rows(wTo, shape, kps, frames, left, new_bbox):
    """Export block (row 15): rows ordered [keypoint rows of all frames ; 4 bbox rows of all frames].
    frames: list of dict(wTc, zs [K x 2, NaN rows invalid], bbox [xmin,ymin,xmax,ymax]).
    Returns res, Hf [rows x (9+3K)], J_cam [rows x 6], counts (valid keypoints per frame)."""
    K = len(kps)
    ncol = 9 + 3 * K
    r1, J1, H1 = [], [], []
    r2, J2, H2 = [], [], []
    counts = []
    for fr in frames:
        cTw = np.linalg.inv(fr['wTc'])
        res, Jc, Hp, Hk, valid = keypoint_rows(cTw, wTo, kps, fr['zs'], left)
        counts.append(len(valid))
        for q in range(len(valid)):
            h = np.zeros((2, ncol))
            h[:, 0:6] = Hp[q]
            kpid, blk = Hk[q]
            h[:, 9 + 3 * kpid: 12 + 3 * kpid] = blk
            r1.append(res[q]); J1.append(Jc[q]); H1.append(h)
        rb, Jb, Hpb, Hsb = bbox_rows(cTw, wTo, shape, fr['bbox'], left, new_bbox)
        h = np.zeros((4, ncol))
        h[:, 0:6] = Hpb
        h[:, 6:9] = Hsb
        r2.append(rb); J2.append(Jb); H2.append(h)
    res = np.concatenate([np.concatenate(r1) if r1 else np.zeros(0), np.concatenate(r2)])
    J_cam = np.vstack(([np.vstack(J1)] if J1 else []) + [np.vstack(J2)])
    Hf = np.vstack(([np.vstack(H1)] if H1 else []) + [np.vstack(H2)])
    return res, Hf, J_cam, counts


# ---------------------------------------------------------------------------------------------
# row 16: constructObjectResidualJacobians (src/orcvio.cpp:2017-2151)
# ---------------------------------------------------------------------------------------------
def construct_object_residual_jacobians(J_cam, frame_to_clone, Hf, res, counts, wTc_list, R_b2c, t_c_b,
                                        vio_left, leg_dim, n_clones, fix_D_identity=False):
    """frame_to_clone[f] = window index of the clone with exactly that timestamp, or -1 (:2073).
    Returns (Hx [rows' x n], Hf', res', row_clone [rows'], Hx6 [rows' x 6]) or None if no frame is in
    the window (:2149).  Output rows interleaved per frame: [kp rows f ; 4 bbox rows f]."""
    n = leg_dim + 6 * n_clones
    sum_zs = 2 * sum(counts)
    Hx_rows, Hf_rows, r_rows, rc, hx6 = [], [], [], [], []
    src = 0
    for f, cnt in enumerate(counts):
        zf = 2 * cnt
        idx = frame_to_clone[f]
        if idx >= 0:
            if fix_D_identity:
                D = np.eye(6)
            else:
                wTc = se3_exp(se3_log(wTc_list[f]))                                  # :2083 exp(valid_camera_pose_mat.col)
                R_w2c = np.linalg.inv(wTc)[:3, :3]
                t_b_w = wTc[:3, :3] @ (-R_b2c @ t_c_b) + wTc[:3, 3]                  # :2091
                D = mirror.cam_wrt_imu_se3_jacobian(R_b2c, t_c_b, R_w2c, t_b_w, vio_left)
            rows = list(range(src, src + zf)) + list(range(sum_zs + 4 * f, sum_zs + 4 * f + 4))
            for q in rows:
                h6 = J_cam[q] @ D
                hx = np.zeros(n)
                hx[leg_dim + 6 * idx: leg_dim + 6 * idx + 6] = h6
                Hx_rows.append(hx); Hf_rows.append(Hf[q]); r_rows.append(res[q]); rc.append(idx); hx6.append(h6)
        src += zf
    if not Hx_rows:
        return None
    return (np.array(Hx_rows), np.array(Hf_rows), np.array(r_rows), np.array(rc, dtype=np.int32), np.array(hx6))


# ---------------------------------------------------------------------------------------------
# row 17: removeLostObjects (src/orcvio.cpp:2154-2193)
# ---------------------------------------------------------------------------------------------
def remove_lost_objects(Hx, Hf, res, P, sigma2, chi2_prob=0.95, table=None):
    """Returns dict(updated, gamma, dof, dx, P_new, G)."""
    n = P.shape[0]
    out = dict(updated=False, gamma=np.nan, dof=0, dx=np.zeros(n), P_new=P.copy(), G=np.zeros((n, n)))
    if Hx.shape[0] == 0:
        return out
    ok, Hp, rp = mirror.nullspace_project_svd(Hf, Hx, res)
    if not ok:
        return out
    dof = Hp.shape[0]
    out['dof'] = dof
    g = mirror.gating_gamma(Hp, rp, P, sigma2)
    out['gamma'] = g
    if not (g < mirror.chi2_threshold(dof, chi2_prob, table)):
        return out
    if np.isnan(Hp).any() or np.isnan(rp).any():
        return out
    Ht, rt = mirror.qr_compress(Hp, rp)
    dx, K, Pn = mirror.measurement_update(Ht, rt, P, sigma2)
    out.update(updated=True, dx=dx, P_new=Pn, G=K @ Ht, H_proj=Hp, r_proj=rp)
    return out

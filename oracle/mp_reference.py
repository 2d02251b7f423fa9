"""TEST INFRASTRUCTURE ONLY -- the reference's update evaluated in 60-digit arithmetic (mpmath), from the POSES to delta_x,
independent of the double-precision restatements (oracle/mirror*.py, oracle/*.c) and of every factorisation they or the
device use.

  measurement_jacobian_mp    measurementJacobian_msckf (src/orcvio.cpp:1071-1168) restated statement by statement in mp arithmetic:
                             the 4 x 4 pose matrices, odotOperator, get_cam_wrt_imu_se3_jacobian (se3_ops.hpp:510-552), all three
                             H_x variants, FEJ, H_e, H_f, r
  measurement_numdiff_mp     the same blocks WITHOUT any Jacobian code: 60-digit central differences of the measurement function
                             pi(p_c) under the error-state increments incrementState_IMUCam applies (:4468-4567) -- clone rotation
                             exp(dtheta) R (LARVIO / left) or R exp(dtheta) (right), positions additive, extrinsic rotation
                             R_b2c R(smallAngleQuaternion(dtheta))^T (math_utils.hpp:104-121), t_c_b additive, p_w additive.
                             With a step of 1e-20 the truncation error is 1e-40 relative: the differences ARE the Jacobian to
                             double precision.  (Not defined under if_FEJ, which linearises at another point on purpose: :1104.)
  feature_jacobian_msckf_mp  featureJacobian_msckf's stacking (:1171-1226), td column included
  msckf_update_mp            nullspace projection (an orthonormal left-null-space basis from an mp Householder QR: any basis gives
                             the same gamma / delta_x / P+, math_utils.hpp:287-312), gatingTestFeature (:1953-1976),
                             K = P H^T (H P H^T + s2 I)^-1, delta_x = K r, P+ = (I - K H) P symmetrised (:1682-1753); the QR
                             compression of the stack (:1664-1679) changes none of these and is skipped.
                             jacobians = 'mp' (default: poses -> delta_x without a double anywhere), 'numdiff' (no Jacobian
                             code at all), or 'mirror' (round 2's form: the double blocks of mirror.py as data)
  objects_update_mp          removeLostObjects (:2154-2193) on given row blocks (H_x, H_f, r per object; data): projection onto the
                             left null space of H_f by an mp QR WITH the rank decision the reference's SVD leaves to rounding
                             noise made explicit (rank = pivots above 1e-25 of the largest), the gate with BOTH counts of the
                             degrees of freedom (rows - columns: the reference's; rows - rank: what gamma sums), the update
  hybrid_update_mp           measurementUpdate_hybrid without entering features (:1766-1950, sz_new = 0): given the MSCKF tracks
                             (as above) and the row pairs of the in-state features (data), their 2-dof gates and the joint update

This is the closest thing to a pin of the feature half that this image allows (the reference cannot be built: no Eigen /
SuiteSparse): tests/test_gpu_conditioning.py measures how far the device AND the double restatements are from it as the
problem's conditioning is swept.  Pure-Python mpmath: windows of a few clones and a dozen features only.
"""
import numpy as np
import mpmath as mp

from oracle import mirror


def _to_mp(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        return mp.matrix([mp.mpf(float(v)) for v in a])
    return mp.matrix([[mp.mpf(float(v)) for v in row] for row in a])


def _to_np(m):
    return np.array([[float(m[i, j]) for j in range(m.cols)] for i in range(m.rows)])


# ---- math_utils.hpp:27-39, se3_ops.hpp:510-552 in mp ---------------------------------------------------------------------
def _skew(w):
    return mp.matrix([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def _odot(x4):
    t = mp.zeros(4, 6)
    S = _skew(x4)
    for i in range(3):
        t[i, i] = x4[3]
        for j in range(3):
            t[i, 3 + j] = -S[i, j]
    return t


def _cam_wrt_imu(R_b2c, t_c_b, R_w2c, t_b_w, left):
    D = mp.zeros(6, 6)
    if left:
        S = _skew(t_b_w)
        for i in range(3):
            for j in range(3):
                D[i, j] = S[i, j]
            D[3 + i, i] = 1
            D[i, 3 + i] = 1
    else:
        A = -R_b2c * _skew(t_c_b)
        for i in range(3):
            for j in range(3):
                D[i, j] = A[i, j]
                D[3 + i, j] = R_b2c[i, j]
                D[i, 3 + j] = R_w2c[i, j]
    return D


def _so3_exp(w):
    """Rodrigues (Sophus SO3d::exp is the same rotation)."""
    th2 = w[0] ** 2 + w[1] ** 2 + w[2] ** 2
    th = mp.sqrt(th2)
    W = _skew(w)
    if th == 0:
        return mp.eye(3)
    return mp.eye(3) + (mp.sin(th) / th) * W + ((1 - mp.cos(th)) / th2) * (W * W)


def _small_angle_rot(dth):
    """R(smallAngleQuaternion(dtheta)) -- math_utils.hpp:104-121 (q = [dtheta/2, sqrt(1 - |dtheta/2|^2)], Hamilton matrix :164-177)."""
    x, y, z = dth[0] / 2, dth[1] / 2, dth[2] / 2
    n2 = x * x + y * y + z * z
    if n2 <= 1:
        w = mp.sqrt(1 - n2)
    else:
        s = 1 / mp.sqrt(1 + n2)
        x, y, z, w = x * s, y * s, z * s, s
    return mp.matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _pose_mp(win, i):
    return (_to_mp(win.R_b2w[i]), _to_mp(win.t_b_w[i]), _to_mp(win.t_fej[i]), _to_mp(win.R_b2c[i]), _to_mp(win.t_c_b[i]))


def measurement_jacobian_mp(win, i, p_w, z):
    """measurementJacobian_msckf, src/orcvio.cpp:1071-1168, statement by statement.  p_w, z: mp vectors."""
    f = win.flags
    R_b2w, t_b_w, t_fej, R_b2c, t_c_b = _pose_mp(win, i)
    R_w2b = R_b2w.T
    R_w2c = R_b2c * R_w2b                                   # :1090
    t_c_w = t_b_w + R_b2w * t_c_b                           # :1091
    p_c = R_w2c * (p_w - t_c_w)                             # :1099-1100
    p_bf_w = (p_w - t_fej) if f.if_fej else (p_w - t_b_w)   # :1104
    dz = mp.zeros(2, 3)                                     # :1107-1111
    dz[0, 0] = 1 / p_c[2]
    dz[1, 1] = 1 / p_c[2]
    dz[0, 2] = -p_c[0] / (p_c[2] * p_c[2])
    dz[1, 2] = -p_c[1] / (p_c[2] * p_c[2])
    if not f.use_larvio:                                    # :1115-1143
        temp = mp.zeros(3, 4)
        for k in range(3):
            temp[k, k] = 1
        wTc = mp.eye(4)
        Rc2w = R_w2c.T
        for a in range(3):
            for b in range(3):
                wTc[a, b] = Rc2w[a, b]
            wTc[a, 3] = t_c_w[a]
        ul = mp.matrix([p_w[0], p_w[1], p_w[2], 1])
        D = _cam_wrt_imu(R_b2c, t_c_b, R_w2c, t_b_w, f.use_left_perturbation)
        cTw = mp.inverse(wTc)
        if f.use_left_perturbation:
            dpc = temp * cTw * _odot(ul) * D
        else:
            dpc = temp * _odot(cTw * ul) * D
        H_x = -dz * dpc
    else:                                                   # :1145-1149
        dpc = mp.zeros(3, 6)
        A = R_w2c * _skew(p_bf_w)
        for a in range(3):
            for b in range(3):
                dpc[a, b] = A[a, b]
                dpc[a, 3 + b] = -R_w2c[a, b]
        H_x = dz * dpc
    dpe = mp.zeros(3, 6)                                    # :1152-1155
    A = R_w2c * _skew(p_bf_w) * R_b2w - R_b2c * _skew(t_c_b)
    for a in range(3):
        for b in range(3):
            dpe[a, b] = A[a, b]
            dpe[a, 3 + b] = -R_b2c[a, b]
    H_e = dz * dpe                                          # :1160
    H_f = dz * R_w2c                                        # :1161
    r = mp.matrix([z[0] - p_c[0] / p_c[2], z[1] - p_c[1] / p_c[2]])   # :1165
    return H_x, H_e, H_f, r


def _project_mp(win, i, p_w, d):
    """pi(p_c) with the 15 error-state increments d = [dtheta 3, dp 3, dtheta_ext 3, dt_ext 3, dp_w 3] applied as
    incrementState_IMUCam does (src/orcvio.cpp:4497-4564)."""
    f = win.flags
    R, t_b_w, _, R_b2c, t_c_b = _pose_mp(win, i)
    Rt = _so3_exp(d[0:3])
    left = bool(f.use_larvio or f.use_left_perturbation)    # :4498, :4543
    R_b2w = Rt * R if left else R * Rt
    t_b_w = t_b_w + mp.matrix(d[3:6])
    R_b2c = R_b2c * _small_angle_rot(d[6:9]).T              # :4512-4516
    t_c_b = t_c_b + mp.matrix(d[9:12])
    R_w2c = R_b2c * R_b2w.T
    t_c_w = t_b_w + R_b2w * t_c_b
    pc = R_w2c * (p_w + mp.matrix(d[12:15]) - t_c_w)
    return mp.matrix([pc[0] / pc[2], pc[1] / pc[2]])


def measurement_numdiff_mp(win, i, p_w, z, step=None):
    """The blocks of measurement_jacobian_mp from central differences of pi alone (no Jacobian code).  r = z - pi."""
    h = mp.mpf(10) ** (-(mp.mp.dps // 3)) if step is None else mp.mpf(step)
    J = mp.zeros(2, 15)
    for c in range(15):
        d = [mp.mpf(0)] * 15
        d[c] = h
        zp = _project_mp(win, i, p_w, d)
        d[c] = -h
        zm = _project_mp(win, i, p_w, d)
        J[0, c] = (zp[0] - zm[0]) / (2 * h)
        J[1, c] = (zp[1] - zm[1]) / (2 * h)
    z0 = _project_mp(win, i, p_w, [mp.mpf(0)] * 15)
    return J[:, 0:6], J[:, 6:12], J[:, 12:15], mp.matrix([z[0] - z0[0], z[1] - z0[1]])


def feature_jacobian_msckf_mp(win, j, jacobians='mp'):
    """featureJacobian_msckf (:1171-1226) before the projection: (H_xj [2M x n], r_j, H_fj [2M x 3]) as mp matrices."""
    f = win.flags
    n = win.n
    lo, hi = int(win.obs_ptr[j]), int(win.obs_ptr[j + 1])
    M = hi - lo
    if jacobians == 'mirror':
        Hx, r, Hf = mirror.feature_jacobian_msckf(win, j, project=False)   # stacked double blocks: DATA
        return _to_mp(Hx), _to_mp(r), _to_mp(Hf)
    Hx, Hf, r = mp.zeros(2 * M, n), mp.zeros(2 * M, 3), mp.zeros(2 * M, 1)
    p_w = _to_mp(win.p_w[j])
    for k in range(M):
        o = lo + k
        i = int(win.obs_clone[o])
        fn = measurement_numdiff_mp if jacobians == 'numdiff' else measurement_jacobian_mp
        H_xi, H_ei, H_fi, r_i = fn(win, i, p_w, _to_mp(win.obs_z[o]))
        for s in range(2):
            for c in range(6):
                Hx[2 * k + s, f.leg_dim + 6 * i + c] = H_xi[s, c]      # :1207
                Hx[2 * k + s, 15 + c] = H_ei[s, c]                      # :1208
            if f.estimate_td:
                Hx[2 * k + s, 21] = mp.mpf(float(win.obs_zvel[o][s]))  # :1210-1211
            for c in range(3):
                Hf[2 * k + s, c] = H_fi[s, c]
            r[2 * k + s] = r_i[s]
    return Hx, r, Hf


def _gain_update(H, r, P, s2):
    """K = P H^T (H P H^T + s2 I)^-1, dx = K r, P+ = (I - K H) P symmetrised (:1682-1753), in mp."""
    HP = H * P
    S = HP * H.T + s2 * mp.eye(H.rows)
    KT = mp.inverse(S) * HP                              # K^T = S^-1 (H P)   (60 digits: the explicit inverse is harmless)
    dx = KT.T * r
    Pn = P - KT.T * HP                                   # (I - K H) P
    return dx, (Pn + Pn.T) / 2


def _stack(blocks, rs, n):
    m = sum(b.rows for b in blocks)
    H, r = mp.zeros(m, n), mp.zeros(m, 1)
    i0 = 0
    for b, rr in zip(blocks, rs):
        H[i0:i0 + b.rows, :] = b
        r[i0:i0 + b.rows, 0] = rr
        i0 += b.rows
    return H, r


def msckf_update_mp(win, digits=60, jacobians='mp'):
    """dict(gamma [F], accept [F], dx [n], P_new [n, n]) of removeLostFeatures' update on `win`, in `digits`-digit arithmetic."""
    mp.mp.dps = digits
    f = win.flags
    n = win.n
    s2 = mp.mpf(float(f.noise_feature)) ** 2
    P = _to_mp(win.P)
    table = mirror.chi2_table(f.chi2_prob)
    blocks, rs, gam, acc = [], [], [], []
    for j in range(win.F):
        lo, hi = int(win.obs_ptr[j]), int(win.obs_ptr[j + 1])
        M = hi - lo
        if M < 2:
            gam.append(float('nan')); acc.append(0)
            continue
        Hx, r, Hf = feature_jacobian_msckf_mp(win, j, jacobians)
        Q, _ = mp.qr(Hf)                                # full Q (2M x 2M), Householder in mp arithmetic
        A = Q[:, 3:]                                    # orthonormal basis of the left nullspace of H_f
        Hp, rp = A.T * Hx, A.T * r
        S = Hp * P * Hp.T + s2 * mp.eye(Hp.rows)
        g = (rp.T * mp.lu_solve(S, rp))[0]
        ok = float(g) < mirror.chi2_threshold(2 * M - 3, f.chi2_prob, table)
        gam.append(float(g)); acc.append(int(ok))
        if ok:
            blocks.append(Hp); rs.append(rp)
    if not blocks:
        return dict(gamma=np.array(gam), accept=np.array(acc, dtype=np.int32), dx=np.zeros(n), P_new=np.array(win.P))
    H, r = _stack(blocks, rs, n)
    dx, Pn = _gain_update(H, r, P, s2)
    return dict(gamma=np.array(gam), accept=np.array(acc, dtype=np.int32), dx=_to_np(dx).ravel(), P_new=_to_np(Pn))


def _left_nullspace(Hf, rank_tol=mp.mpf(10) ** -25):
    """Orthonormal basis of the left null space of H_f (rows x cols, rows > cols) with an EXPLICIT rank decision: Householder QR with
    column pivoting in mp; columns whose remaining norm falls below rank_tol of the largest pivot are dependent.  Returns
    (A [rows x (rows - rank)], rank)."""
    rows, cols = Hf.rows, Hf.cols
    R = Hf.copy()
    Q = mp.eye(rows)
    piv = list(range(cols))
    rank = 0
    first = None
    for k in range(cols):
        # pivot: the remaining column of largest norm
        best, bn = k, mp.mpf(-1)
        for c in range(k, cols):
            nn = sum(R[i, c] ** 2 for i in range(k, rows))
            if nn > bn:
                best, bn = c, nn
        if best != k:
            for i in range(rows):
                R[i, k], R[i, best] = R[i, best], R[i, k]
            piv[k], piv[best] = piv[best], piv[k]
        nrm = mp.sqrt(bn)
        if first is None:
            first = nrm
        if first == 0 or nrm <= rank_tol * first:
            break
        alpha = R[k, k]
        beta = -nrm if alpha >= 0 else nrm
        v = mp.zeros(rows, 1)
        v[k] = alpha - beta
        for i in range(k + 1, rows):
            v[i] = R[i, k]
        vv = sum(v[i] ** 2 for i in range(k, rows))
        if vv != 0:
            for c in range(k, cols):
                s = sum(v[i] * R[i, c] for i in range(k, rows)) * 2 / vv
                for i in range(k, rows):
                    R[i, c] -= s * v[i]
            for c in range(rows):          # Q <- Q (I - 2 v v^T / vv)
                s = sum(Q[c, i] * v[i] for i in range(k, rows)) * 2 / vv
                for i in range(k, rows):
                    Q[c, i] -= s * v[i]
        rank += 1
    return Q[:, rank:], rank


def objects_update_mp(blocks, P, sigma, chi2_prob=0.95, digits=50):
    """removeLostObjects (src/orcvio.cpp:2154-2193) with per-object projection on given row blocks [(H_x [m x n], H_f [m x no], r [m])]
    (double data): projection onto the WHOLE left null space of every H_f, gamma of the stacked projected rows, the two counts of
    its degrees of freedom, and the update (applied when gamma passes the reference's count).  dict(gamma, dof_ref, dof_rank,
    rank_deficient, thr_ref, thr_rank, accept_ref, accept_rank, dx, P_new)."""
    mp.mp.dps = digits
    n = P.shape[0]
    s2 = mp.mpf(float(sigma)) ** 2
    Pm = _to_mp(P)
    Hs, rs, dof_ref, dof_rank, deficient = [], [], 0, 0, 0
    for Hx, Hf, r in blocks:
        if Hx.shape[0] <= Hf.shape[1]:
            continue
        A, rank = _left_nullspace(_to_mp(Hf))
        deficient += int(rank < Hf.shape[1])
        Hs.append(A.T * _to_mp(Hx)); rs.append(A.T * _to_mp(r))
        dof_ref += Hx.shape[0] - Hf.shape[1]
        dof_rank += Hx.shape[0] - rank
    if not Hs:
        return dict(gamma=float('nan'), dof_ref=0, dof_rank=0, rank_deficient=0, accept_ref=0, accept_rank=0, dx=np.zeros(n), P_new=np.array(P))
    H, r = _stack(Hs, rs, n)
    S = H * Pm * H.T + s2 * mp.eye(H.rows)
    g = float((r.T * mp.lu_solve(S, r))[0])
    thr_ref, thr_rank = mirror.chi2_threshold(dof_ref, chi2_prob), mirror.chi2_threshold(dof_rank, chi2_prob)
    dx, Pn = _gain_update(H, r, Pm, s2)
    return dict(gamma=g, dof_ref=dof_ref, dof_rank=dof_rank, rank_deficient=deficient, thr_ref=thr_ref, thr_rank=thr_rank,
                accept_ref=int(g < thr_ref), accept_rank=int(g < thr_rank), dx=_to_np(dx).ravel(), P_new=_to_np(Pn))


def hybrid_update_mp(win, ekf_blocks, digits=60, jacobians='mp'):
    """measurementUpdate_hybrid without entering features (src/orcvio.cpp:1766-1950, sz_new = 0): the MSCKF tracks of `win` (poses ->
    blocks in mp) and the row pairs [(H [2 x n], r [2])] of the in-state features the current state observes (double data, e.g.
    mirror_hybrid.feature_jacobian_ekf), every pair gated with two degrees of freedom (:2457), ONE update with what passed.
    dict(accept, ekf_accept, dx, P_new)."""
    mp.mp.dps = digits
    f = win.flags
    n = win.n
    s2 = mp.mpf(float(f.noise_feature)) ** 2
    P = _to_mp(win.P)
    table = mirror.chi2_table(f.chi2_prob)
    blocks, rs, acc, eacc = [], [], [], []
    for j in range(win.F):
        M = int(win.obs_ptr[j + 1]) - int(win.obs_ptr[j])
        if M < 2:
            acc.append(0)
            continue
        Hx, r, Hf = feature_jacobian_msckf_mp(win, j, jacobians)
        Q, _ = mp.qr(Hf)
        A = Q[:, 3:]
        Hp, rp = A.T * Hx, A.T * r
        S = Hp * P * Hp.T + s2 * mp.eye(Hp.rows)
        ok = float((rp.T * mp.lu_solve(S, rp))[0]) < mirror.chi2_threshold(2 * M - 3, f.chi2_prob, table)
        acc.append(int(ok))
        if ok:
            blocks.append(Hp); rs.append(rp)
    for H2, r2 in ekf_blocks:
        Hm, rm = _to_mp(H2), _to_mp(r2)
        S = Hm * P * Hm.T + s2 * mp.eye(2)
        ok = float((rm.T * mp.lu_solve(S, rm))[0]) < mirror.chi2_threshold(2, f.chi2_prob, table)
        eacc.append(int(ok))
        if ok:
            blocks.append(Hm); rs.append(rm)
    if not blocks:
        return dict(accept=np.array(acc, dtype=np.int32), ekf_accept=np.array(eacc, dtype=np.int32), dx=np.zeros(n), P_new=np.array(win.P))
    H, r = _stack(blocks, rs, n)
    dx, Pn = _gain_update(H, r, P, s2)
    return dict(accept=np.array(acc, dtype=np.int32), ekf_accept=np.array(eacc, dtype=np.int32), dx=_to_np(dx).ravel(), P_new=_to_np(Pn))

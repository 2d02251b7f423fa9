"""TEST INFRASTRUCTURE ONLY -- the reference's update FORMULA evaluated in 60-digit arithmetic (mpmath), independent of both
double-precision restatements (oracle/mirror.py, oracle/msckf_oracle.c) and of every factorisation they or the device use.

Inputs are the per-observation Jacobian blocks H_x, H_e, H_f and residuals r of measurementJacobian_msckf
(src/orcvio.cpp:1071-1168; double closed forms from mirror.py -- they are data here).  Everything behind them is restated in
exact-enough arithmetic:

  featureJacobian_msckf      stacking of the blocks (:1171-1226)
  nullspace projection       the projector onto the left nullspace of H_f (math_utils.hpp:287-312): any orthonormal basis A gives
                             the same gamma / delta_x / P+, so the update is written with  N = I - H_f (H_f^T H_f)^-1 H_f^T
                             (H'^T W H' = H_x^T N W ... is avoided: an explicit orthonormal A comes from mp.qr)
  gatingTestFeature          gamma = r'^T (H' P H'^T + s2 I)^-1 r'  (:1953-1976)
  measurementUpdate_msckf    K = P H^T (H P H^T + s2 I)^-1, delta_x = K r, P+ = (I - K H) P, symmetrised (:1682-1753); the QR
                             compression of the stack (:1664-1679) does not change any of these and is skipped

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


def msckf_update_mp(win, digits=60):
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
        Hx, r, Hf = mirror.feature_jacobian_msckf(win, j, project=False)   # stacked double blocks: DATA
        Hx, r, Hf = _to_mp(Hx), _to_mp(r), _to_mp(Hf)
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
    m = sum(b.rows for b in blocks)
    H = mp.zeros(m, n)
    r = mp.zeros(m, 1)
    i0 = 0
    for b, rr in zip(blocks, rs):
        H[i0:i0 + b.rows, :] = b
        r[i0:i0 + b.rows, 0] = rr
        i0 += b.rows
    HP = H * P
    S = HP * H.T + s2 * mp.eye(m)
    KT = mp.inverse(S) * HP                              # K^T = S^-1 (H P)   (60 digits: the explicit inverse is harmless)
    dx = KT.T * r
    Pn = P - KT.T * HP                                   # (I - K H) P
    Pn = (Pn + Pn.T) / 2
    return dict(gamma=np.array(gam), accept=np.array(acc, dtype=np.int32), dx=_to_np(dx).ravel(), P_new=_to_np(Pn))

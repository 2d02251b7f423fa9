"""TEST INFRASTRUCTURE ONLY -- literal numpy restatement of the reference's feature triangulation.

Follows include/orcvio/feat/feature.hpp:
  cost                   :270-290
  jacobian               :292-330
  generateInitialGuess   :332-352
  checkMotion            :354-397
  initializePosition     :399-449
  triangulate_position   :583-719
  OptimizationConfig     :41-63 (defaults)
Camera poses are the cached (orientation_cam, position_cam) of src/orcvio.cpp:954-961: R_c2w = R_b2w R_b2c^T,
t_c_w = t_b_w + R_b2w t_c_b.  Parity unpinned: the reference has no test for these functions; the restatement is
anchored on its own text, on re-projection of synthetic ground truth and on the committed vectors
tests/golden/tri_*.npz.  Nothing under orcvio_amd/ may import this module.
"""
import dataclasses

import numpy as np


@dataclasses.dataclass
class OptimizationConfig:   # feature.hpp:41-63
    translation_threshold: float = 0.2
    huber_epsilon: float = 0.01
    estimation_precision: float = 5e-7
    initial_damping: float = 1e-3
    outer_loop_max_iteration: int = 10
    inner_loop_max_iteration: int = 10
    cost_threshold: float = 4.7673e-04
    init_final_dist_threshold: float = 5.0


FLAG_NO_MOTION = 1       # checkMotion returned false (triangulation not attempted)
FLAG_NEG_DEPTH = 2       # failed_by_neg_dpth
FLAG_BIG_PROJ = 4        # failed_by_big_proj (distance to the initial guess, or normalised cost)


def cam_pose(R_b2w, t_b_w, R_b2c, t_c_b):
    """(R_c2w, t_c_w) as src/orcvio.cpp:954-961."""
    return R_b2w @ R_b2c.T, t_b_w + R_b2w @ t_c_b


def _cost(R, t, x, z):   # feature.hpp:270-290
    h = R @ np.array([x[0], x[1], 1.0]) + x[2] * t
    zh = np.array([h[0] / h[2], h[1] / h[2]])
    return float(np.sum((zh - z) ** 2))


def _jacobian(R, t, x, z, huber):   # feature.hpp:292-330
    h = R @ np.array([x[0], x[1], 1.0]) + x[2] * t
    W = np.column_stack([R[:, 0], R[:, 1], t])
    J = np.vstack([W[0] / h[2] - h[0] / (h[2] * h[2]) * W[2], W[1] / h[2] - h[1] / (h[2] * h[2]) * W[2]])
    r = np.array([h[0] / h[2], h[1] / h[2]]) - z
    e = float(np.linalg.norm(r))
    w = 1.0 if e <= huber else np.sqrt(2.0 * huber / e)
    return J, r, w


def check_motion(Rs, ts, z_first):   # feature.hpp:354-397 (first and last listed observation)
    d = np.array([z_first[0], z_first[1], 1.0])
    d = Rs[0] @ (d / np.linalg.norm(d))
    tr = ts[-1] - ts[0]
    par = float(tr @ d)
    return float(np.linalg.norm(tr - par * d))


def triangulate(Rs, ts, zs, cfg: OptimizationConfig, prior_p_w=None):
    """Rs[i], ts[i]: camera i -> world; zs[i]: normalised observation.  prior_p_w: the feature's current position if it
    is_initialized (feature.hpp:604-606), else None.  Returns dict(valid, p_w, solution, flags, cost, iterations)."""
    M = len(Rs)
    Rl, tl = Rs[-1], ts[-1]
    # pose_i^-1 * pose_last : last camera frame -> camera i frame (:590-592)
    Rr = [Rs[i].T @ Rl for i in range(M)]
    tr = [Rs[i].T @ (tl - ts[i]) for i in range(M)]
    if prior_p_w is None:   # generateInitialGuess(cam_poses[0], z_last, z_0) (:332-352, :597-599)
        m = Rr[0] @ np.array([zs[-1][0], zs[-1][1], 1.0])
        A = np.array([m[0] - zs[0][0] * m[2], m[1] - zs[0][1] * m[2]])
        b = np.array([zs[0][0] * tr[0][2] - tr[0][0], zs[0][1] * tr[0][2] - tr[0][1]])
        depth = float(A @ b) / float(A @ A)
        init = np.array([zs[-1][0] * depth, zs[-1][1] * depth, depth])
    else:
        init = Rl.T @ (np.asarray(prior_p_w) - tl)
    x = np.array([init[0] / init[2], init[1] / init[2], 1.0 / init[2]])
    lam = cfg.initial_damping
    inner = 0
    outer = 0
    reduced = False
    delta_norm = 0.0
    total = sum(_cost(Rr[i], tr[i], x, zs[i]) for i in range(M))
    iters = 0
    while True:   # outer do-while (:621-679)
        A = np.zeros((3, 3))
        b = np.zeros(3)
        for i in range(M):
            J, r, w = _jacobian(Rr[i], tr[i], x, zs[i], cfg.huber_epsilon)
            w2 = 1.0 if w == 1 else w * w
            A += w2 * J.T @ J
            b += w2 * J.T @ r
        while True:   # inner do-while (:642-668)
            delta = np.linalg.solve(A + lam * np.eye(3), b)
            xn = x - delta
            delta_norm = float(np.linalg.norm(delta))
            new = sum(_cost(Rr[i], tr[i], xn, zs[i]) for i in range(M))
            iters += 1
            if new < total:
                reduced = True
                x = xn
                total = new
                lam = lam / 10 if lam / 10 > 1e-10 else 1e-10
            else:
                reduced = False
                lam = lam * 10 if lam * 10 < 1e12 else 1e12
            cont = inner < cfg.inner_loop_max_iteration and not reduced
            inner += 1
            if not cont:
                break
        inner = 0
        cont = outer < cfg.outer_loop_max_iteration and delta_norm > cfg.estimation_precision
        outer += 1
        if not cont:
            break
    final = np.array([x[0] / x[2], x[1] / x[2], 1.0 / x[2]])
    flags = 0
    for i in range(M):   # in front of every camera (:690-699)
        if (Rr[i] @ final + tr[i])[2] <= 0:
            flags |= FLAG_NEG_DEPTH
            break
    normalized = total / (2 * M * M)
    if np.linalg.norm(final - init) > cfg.init_final_dist_threshold:
        flags |= FLAG_BIG_PROJ
    if normalized > cfg.cost_threshold:
        flags |= FLAG_BIG_PROJ
    return dict(valid=flags == 0, p_w=Rl @ final + tl, solution=x, flags=flags, cost=total, iterations=iters,
                final_position=final, initial_position=init)


def triangulate_tracks(win, cfg: OptimizationConfig = None, is_initialized=None):
    """Batch form over a synth.Window (observations as listed: the caller has already dropped the current frame).
    checkMotion first (src/orcvio.cpp:2260-2269), then initializePosition.  Returns arrays over the F tracks."""
    cfg = cfg or OptimizationConfig()
    F = win.F
    out = dict(valid=np.zeros(F, np.int32), p_w=np.full((F, 3), np.nan), solution=np.full((F, 3), np.nan),
               flags=np.zeros(F, np.int32), cost=np.full(F, np.nan), motion=np.full(F, np.nan))
    pose = [cam_pose(win.R_b2w[i], win.t_b_w[i], win.R_b2c[i], win.t_c_b[i]) for i in range(win.N)]
    for j in range(F):
        lo, hi = int(win.obs_ptr[j]), int(win.obs_ptr[j + 1])
        if hi - lo < 2:
            out['flags'][j] = FLAG_NO_MOTION
            continue
        cl = win.obs_clone[lo:hi]
        Rs = [pose[c][0] for c in cl]
        ts = [pose[c][1] for c in cl]
        zs = [win.obs_z[o] for o in range(lo, hi)]
        init = is_initialized is not None and bool(is_initialized[j])
        mo = check_motion(Rs, ts, zs[0])
        out['motion'][j] = mo
        if not init and not (mo > cfg.translation_threshold):
            out['flags'][j] = FLAG_NO_MOTION
            continue
        r = triangulate(Rs, ts, zs, cfg, prior_p_w=win.p_w[j] if init else None)
        out['valid'][j] = 1 if r['valid'] else 0
        out['p_w'][j] = r['p_w']
        out['solution'][j] = r['solution']
        out['flags'][j] = r['flags']
        out['cost'][j] = r['cost']
    return out

import glob
import os

import numpy as np

from orcvio_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, 'feat_*.npz')))


def window_from_golden(path):
    g = np.load(path)
    fl = g['flags']
    flags = synth.Flags(leg_dim=int(fl[0]), use_larvio=int(fl[1]), use_left_perturbation=int(fl[2]), if_fej=int(fl[3]),
                        estimate_td=int(fl[4]), discard_large_update=int(fl[5]),
                        noise_feature=float(g['noise_feature']), chi2_prob=float(g['chi2_prob']))
    w = synth.Window(R_b2w=g['R_b2w'], t_b_w=g['t_b_w'], t_fej=g['t_fej'], R_b2c=g['R_b2c'], t_c_b=g['t_c_b'],
                     p_w=g['p_w'], obs_ptr=g['obs_ptr'].astype(np.int32), obs_clone=g['obs_clone'].astype(np.int32),
                     obs_z=g['obs_z'], obs_zvel=g['obs_zvel'], P=g['P'], flags=flags)
    return w, g


def subset_window(win, clone_ids):
    """CSR restricted to the observations of `clone_ids` -- what the host side passes for the
    pruneImuStateBuffer update (reference src/orcvio.cpp:2810-2845)."""
    keep = np.isin(win.obs_clone, np.asarray(clone_ids))
    ptr = [0]
    for j in range(win.F):
        ptr.append(ptr[-1] + int(keep[win.obs_ptr[j]:win.obs_ptr[j + 1]].sum()))
    import dataclasses
    return dataclasses.replace(win, obs_ptr=np.asarray(ptr, dtype=np.int32), obs_clone=win.obs_clone[keep].copy(),
                               obs_z=win.obs_z[keep].copy(), obs_zvel=win.obs_zvel[keep].copy())


def scatter_tracks(win, rng, lo, hi):
    """Every track keeps a random SUBSET (size in [lo, hi]) of its observations: non-contiguous clone lists, what tracks look
    like after clones in the middle of the window were marginalised (src/orcvio.cpp:2874-2956)."""
    import dataclasses
    ptr, oc, oz, ov = [0], [], [], []
    for j in range(win.F):
        a, b = int(win.obs_ptr[j]), int(win.obs_ptr[j + 1])
        M = b - a
        k = int(rng.integers(min(lo, M), min(hi, M) + 1))
        keep = np.sort(rng.choice(M, k, replace=False)) + a
        oc.append(win.obs_clone[keep]); oz.append(win.obs_z[keep]); ov.append(win.obs_zvel[keep])
        ptr.append(ptr[-1] + k)
    return dataclasses.replace(win, obs_ptr=np.asarray(ptr, dtype=np.int32), obs_clone=np.ascontiguousarray(np.concatenate(oc)),
                               obs_z=np.ascontiguousarray(np.concatenate(oz)), obs_zvel=np.ascontiguousarray(np.concatenate(ov)))


def object_rows_reference(win, obj, obj_left, new_bbox, vio_left):
    """Rows of one object track in window coordinates through the mirror (functor rows + constructObjectResidualJacobians)."""
    from oracle import mirror_objects as mo
    res, Hf, Jc, counts = mo.object_rows(obj.wTo, obj.shape, obj.kps, obj.frames, obj_left, new_bbox)
    f2c = [fr['clone'] for fr in obj.frames]
    return mo.construct_object_residual_jacobians(Jc, f2c, Hf, res, counts, [fr['wTc'] for fr in obj.frames],
                                                  win.R_b2c[0], win.t_c_b[0], vio_left, win.flags.leg_dim, win.N)


def objects_update_reference(win, objs, P, obj_left=True, new_bbox=False, vio_left=0, full_nullspace=False, rank_dof=False):
    """The object update of System::processObjects -> removeLostObjects with per-object projection (SURVEY note N3) on the
    prior P: dict(gamma, accept, dof, dx, P_new, blocks) -- the mirror's literal arithmetic (full-U nullspace per object,
    QR of the stack, S, K, (I - KH) P).
    full_nullspace: project every object onto its WHOLE left null space (rows - rank(H_f) directions) and keep the reference's
    count rows - columns for the gate -- what the device does when H_f is rank deficient (a keypoint never seen in the window, or
    seen once).  There the reference keeps rows - columns directions of that space chosen by Eigen's column-pivoted Householder
    QR inside JacobiSVD: not determined by the inputs; for a full-rank H_f the two are the same.
    rank_dof (with full_nullspace): the gate counts rows - rank(H_f) degrees of freedom -- the number of directions gamma sums
    (ORCVIO_OPT_OBJECT_DOF = 1)."""
    from oracle import mirror
    blocks, Hp, rp, dof, deficient = [], [], [], 0, 0
    for ob in objs:
        rows = object_rows_reference(win, ob, obj_left, new_bbox, vio_left)
        if rows is None:   # no frame of the track is in the window
            continue
        Hx, Hf, r, rc, hx6 = rows
        if Hx.shape[0] <= Hf.shape[1]:
            continue
        blocks.append(dict(row_clone=rc, Hx6=hx6, Hf=Hf, res=r))
        if full_nullspace:
            U, sv, _ = np.linalg.svd(Hf, full_matrices=True)
            rank = int((sv > 1e-11 * sv[0]).sum())
            deficient += int(rank < Hf.shape[1])
            A = U[:, rank:]
            H1, r1 = A.T @ Hx, A.T @ r
        else:
            ok, H1, r1 = mirror.nullspace_project_svd(Hf, Hx, r)
            rank = Hf.shape[1]
        Hp.append(H1); rp.append(r1)
        dof += Hx.shape[0] - (rank if (rank_dof and full_nullspace) else Hf.shape[1])
    n = P.shape[0]
    if not Hp:
        return dict(gamma=float('nan'), accept=0, dof=0, dx=np.zeros(n), P_new=P.copy(), blocks=blocks, rank_deficient=0)
    H = np.vstack(Hp); r = np.concatenate(rp)
    s2 = win.flags.noise_feature ** 2
    Q1, R = np.linalg.qr(H)
    r1 = Q1.T @ r
    gamma = float(r1 @ np.linalg.solve(R @ P @ R.T + s2 * np.eye(R.shape[0]), r1) + (r @ r - r1 @ r1) / s2)
    accept = int(gamma < mirror.chi2_threshold(dof, win.flags.chi2_prob))
    if accept:
        Ht, rt = mirror.qr_compress(H, r)
        dx, K, Pn = mirror.measurement_update(Ht, rt, P, s2)
    else:
        dx, Pn = np.zeros(n), P.copy()
    return dict(gamma=gamma, accept=accept, dof=dof, dx=dx, P_new=Pn, blocks=blocks, rank_deficient=deficient)


def random_object_case(seed, bbox_only_frac=0.0):
    """A random window with object tracks for the randomised object-update checks (scripts/gpu_soak_objects.py and
    test_gpu_objects.py): number of objects, keypoints per object, frames inside / outside the window, missing keypoints,
    residual form, perturbation sides, keypoint noise.  Small windows with few frames give rank-deficient H_f blocks."""
    rng = np.random.default_rng(880000 + seed)
    N = int(rng.integers(4, 31))
    nobj = int(rng.integers(1, 9))
    obj_left = bool(rng.integers(0, 2)); new_bbox = bool(rng.integers(0, 2)); vio_left = int(rng.integers(0, 2))
    flags = synth.Flags(use_larvio=0, use_left_perturbation=vio_left, leg_dim=int(rng.choice([22, 22, 46])),
                        noise_feature=float(rng.choice([0.008, 0.05])))
    fpo = None if rng.integers(0, 2) else int(rng.integers(2, N + 1))
    sig = float(rng.choice([0.004, 0.004, 0.1]))
    par = dict(seed=seed, N=N, nobj=nobj, obj_left=obj_left, new_bbox=new_bbox, vio_left=vio_left, leg=flags.leg_dim, fpo=fpo, sig=sig)
    win = synth.make_window(N=N, F=4, seed=seed, flags=flags, track_len=min(4, N))
    objs = synth.make_objects(win, n_objects=nobj, seed=seed, missing_frac=float(rng.choice([0.0, 0.1, 0.4])),
                              frames_per_object=fpo, sigma_kp=sig)
    for ob in objs:
        if rng.random() < 0.3:   # fewer keypoints: a narrower object state
            K = int(rng.choice([4, 8]))
            ob.kps = ob.kps[:K].copy()
            for fr in ob.frames:
                fr['zs'] = fr['zs'][:K].copy()
        if rng.random() < 0.3:   # some frames left the window
            for fr in ob.frames:
                if rng.random() < 0.4:
                    fr['clone'] = -1
    if bbox_only_frac > 0.0:   # (a stream of its own: the cases of the default call are the same as ever)
        rng2 = np.random.default_rng(770000 + seed)
        for ob in objs:
            if rng2.random() < bbox_only_frac:   # a bbox-only track (no keypoints: object state 9 columns)
                ob.kps = np.zeros((0, 3))
                for fr in ob.frames:
                    fr['zs'] = np.zeros((0, 2))
    return dict(win=win, objs=objs, obj_left=obj_left, new_bbox=new_bbox, vio_left=vio_left, flags=flags, par=par,
                resident=bool(rng.integers(0, 2)))

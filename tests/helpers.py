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


def object_rows_reference(win, obj, obj_left, new_bbox, vio_left):
    """Rows of one object track in window coordinates through the mirror (functor rows + constructObjectResidualJacobians)."""
    from oracle import mirror_objects as mo
    res, Hf, Jc, counts = mo.object_rows(obj.wTo, obj.shape, obj.kps, obj.frames, obj_left, new_bbox)
    f2c = [fr['clone'] for fr in obj.frames]
    return mo.construct_object_residual_jacobians(Jc, f2c, Hf, res, counts, [fr['wTc'] for fr in obj.frames],
                                                  win.R_b2c[0], win.t_c_b[0], vio_left, win.flags.leg_dim, win.N)


def objects_update_reference(win, objs, P, obj_left=True, new_bbox=False, vio_left=0):
    """The object update of System::processObjects -> removeLostObjects with per-object projection (SURVEY note N3) on the
    prior P: dict(gamma, accept, dof, dx, P_new, blocks) -- the mirror's literal arithmetic (full-U nullspace per object,
    QR of the stack, S, K, (I - KH) P)."""
    from oracle import mirror
    blocks, Hp, rp = [], [], []
    for ob in objs:
        Hx, Hf, r, rc, hx6 = object_rows_reference(win, ob, obj_left, new_bbox, vio_left)
        if Hx.shape[0] <= Hf.shape[1]:
            continue
        blocks.append(dict(row_clone=rc, Hx6=hx6, Hf=Hf, res=r))
        ok, H1, r1 = mirror.nullspace_project_svd(Hf, Hx, r)
        Hp.append(H1); rp.append(r1)
    n = P.shape[0]
    if not Hp:
        return dict(gamma=float('nan'), accept=0, dof=0, dx=np.zeros(n), P_new=P.copy(), blocks=blocks)
    H = np.vstack(Hp); r = np.concatenate(rp)
    s2 = win.flags.noise_feature ** 2
    Q1, R = np.linalg.qr(H)
    r1 = Q1.T @ r
    gamma = float(r1 @ np.linalg.solve(R @ P @ R.T + s2 * np.eye(R.shape[0]), r1) + (r @ r - r1 @ r1) / s2)
    accept = int(gamma < mirror.chi2_threshold(H.shape[0], win.flags.chi2_prob))
    if accept:
        Ht, rt = mirror.qr_compress(H, r)
        dx, K, Pn = mirror.measurement_update(Ht, rt, P, s2)
    else:
        dx, Pn = np.zeros(n), P.copy()
    return dict(gamma=gamma, accept=accept, dof=H.shape[0], dx=dx, P_new=Pn, blocks=blocks)

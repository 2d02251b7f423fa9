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

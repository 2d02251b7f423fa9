"""CPU tests of the triangulation restatement (oracle/mirror_triangulate.py): the committed vectors, and properties the
reference's algorithm must have whatever the basis of comparison (no reference test exists for these functions:
parity unpinned)."""
import glob
import os

import numpy as np
import pytest

from orcvio_amd import synth
from oracle import mirror_triangulate as mt
from helpers import GOLDEN, rel


def tri_files():
    return sorted(glob.glob(os.path.join(GOLDEN, 'tri_*.npz')))


def window_from_tri(g):
    N = g['R_b2w'].shape[0]
    nobs = g['obs_clone'].shape[0]
    return synth.Window(R_b2w=g['R_b2w'], t_b_w=g['t_b_w'], t_fej=g['t_b_w'].copy(), R_b2c=g['R_b2c'], t_c_b=g['t_c_b'],
                        p_w=g['p_w'], obs_ptr=g['obs_ptr'].astype(np.int32), obs_clone=g['obs_clone'].astype(np.int32),
                        obs_z=g['obs_z'], obs_zvel=np.zeros((nobs, 2)), P=np.eye(22 + 6 * N), flags=synth.Flags())


@pytest.mark.parametrize('path', tri_files(), ids=lambda p: os.path.basename(p)[4:-4])
def test_golden_vectors(path):
    g = np.load(path)
    w = window_from_tri(g)
    ini = g['is_initialized'] if g['is_initialized'].size else None
    r = mt.triangulate_tracks(w, is_initialized=ini)
    assert np.array_equal(r['valid'], g['exp_valid'])
    assert np.array_equal(r['flags'], g['exp_flags'])
    ok = g['exp_valid'] == 1
    assert rel(r['p_w'][ok], g['exp_p_w'][ok]) < 1e-12
    assert rel(r['solution'][ok], g['exp_solution'][ok]) < 1e-12


def test_noise_free_observations_recover_the_point():
    w = synth.make_window(N=10, F=30, seed=3, track_len=(3, 8), sigma_px=0.0)
    r = mt.triangulate_tracks(w)
    ok = r['valid'] == 1
    assert ok.sum() >= 25
    # the generator's p_w is truth + 2 cm: the noise-free fit must sit within that ball, with ~zero cost
    assert np.max(np.linalg.norm(r['p_w'][ok] - w.p_w[ok], axis=1)) < 0.15
    assert np.nanmax(r['cost'][ok]) < 1e-20


def test_solution_is_a_stationary_point_of_the_cost():
    # noise well below huber_epsilon: every weight is 1 and the fit minimises the plain squared cost
    w = synth.make_window(N=8, F=10, seed=5, track_len=(4, 8), sigma_px=1e-4)
    pose = [mt.cam_pose(w.R_b2w[i], w.t_b_w[i], w.R_b2c[i], w.t_c_b[i]) for i in range(w.N)]
    r = mt.triangulate_tracks(w)
    for j in np.flatnonzero(r['valid']):
        lo, hi = int(w.obs_ptr[j]), int(w.obs_ptr[j + 1])
        cl = w.obs_clone[lo:hi]
        Rl, tl = pose[cl[-1]]

        def cost(x):
            c = 0.0
            for k, ci in enumerate(cl):
                Rr = pose[ci][0].T @ Rl
                tr = pose[ci][0].T @ (tl - pose[ci][1])
                c += mt._cost(Rr, tr, x, w.obs_z[lo + k])
            return c
        x = r['solution'][j]
        g = np.array([(cost(x + h) - cost(x - h)) / 2e-6 for h in 1e-6 * np.eye(3)])
        assert np.linalg.norm(g) < 1e-7


def test_motion_check_and_short_tracks():
    w = synth.make_window(N=6, F=12, seed=9, track_len=2)
    r = mt.triangulate_tracks(w)
    # two neighbouring frames are 0.15 m apart: below the 0.2 m translation threshold (feature.hpp:52)
    assert (r['flags'] == mt.FLAG_NO_MOTION).all() and not r['valid'].any()
    cfg = mt.OptimizationConfig(translation_threshold=0.05)
    r2 = mt.triangulate_tracks(w, cfg)
    assert (r2['flags'] & mt.FLAG_NO_MOTION == 0).all()


def test_prior_start_skips_motion_check_and_converges_to_the_same_point():
    w = synth.make_window(N=8, F=12, seed=4, track_len=(4, 8), sigma_px=1e-4)
    a = mt.triangulate_tracks(w)
    b = mt.triangulate_tracks(w, is_initialized=np.ones(w.F, np.int32))
    ok = (a['valid'] == 1) & (b['valid'] == 1)
    assert ok.sum() >= 10
    assert np.max(np.linalg.norm(a['p_w'][ok] - b['p_w'][ok], axis=1)) < 1e-4

"""GPU parity of the triangulation kernel (k_triangulate, through orcvio_msckf_triangulate) with the oracle.
Tolerance: positions 1e-6 relative (north_star's figure for floating point), flags and validity identical."""
import dataclasses
import os

import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import mirror_triangulate as mt, oracle
from helpers import rel
from test_oracle_triangulate import tri_files, window_from_tri

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=40, max_features=2048, max_observations=65536)
    yield u
    u.close()


def _compare(got, ref):
    assert np.array_equal(got['valid'], ref['valid'])
    assert np.array_equal(got['flags'], ref['flags'])
    ok = ref['valid'] == 1
    if ok.any():
        assert rel(got['p_w'][ok], ref['p_w'][ok]) < TOL
        assert rel(got['solution'][ok], ref['solution'][ok]) < TOL
        assert np.max(np.abs(got['cost'][ok] - ref['cost'][ok])) < 1e-9 * max(1.0, np.max(ref['cost'][ok]) / 1e-6)


@pytest.mark.parametrize('path', tri_files(), ids=lambda p: os.path.basename(p)[4:-4])
def test_golden_vectors(upd, path):
    g = np.load(path)
    w = window_from_tri(g)
    ini = g['is_initialized'] if g['is_initialized'].size else None
    got = upd.triangulate(w, is_initialized=ini)
    _compare(got, dict(valid=g['exp_valid'], flags=g['exp_flags'], p_w=g['exp_p_w'], solution=g['exp_solution'],
                       cost=g['exp_cost']))


@pytest.mark.parametrize('cfg', [1, 2])
def test_config_windows(upd, cfg):
    w = synth.config_window(cfg)
    _compare(upd.triangulate(w), mt.triangulate_tracks(w))


def test_thresholds_and_iteration_limits(upd):
    w = synth.make_window(N=10, F=60, seed=31, track_len=(2, 9), outlier_frac=0.3)
    for c in (mt.OptimizationConfig(translation_threshold=0.05, cost_threshold=1e-5),
              mt.OptimizationConfig(outer_loop_max_iteration=1, inner_loop_max_iteration=2, huber_epsilon=1e-3),
              mt.OptimizationConfig(init_final_dist_threshold=0.05)):
        _compare(upd.triangulate(w, cfg=c), mt.triangulate_tracks(w, c))


def test_maximum_track_length_and_empty(upd):
    w = synth.make_window(N=32, F=5, seed=2)
    _compare(upd.triangulate(w), mt.triangulate_tracks(w))
    e = synth.make_window(N=4, F=0, seed=1)
    got = upd.triangulate(e)
    assert got['valid'].shape == (0,)


def test_triangulate_then_update_on_device(upd):
    """Raw observations in, update out: tracks uploaded WITHOUT positions, triangulated in place, failed tracks take no
    part -- equal to the oracle's update of the tracks the oracle's triangulation keeps, at the oracle's positions."""
    w = synth.make_window(N=12, F=80, seed=17, track_len=(2, 10), outlier_frac=0.2)
    tri = mt.triangulate_tracks(w)
    keep = tri['valid'] == 1
    assert 10 < keep.sum() < w.F
    # reference flow: invalid tracks are erased from the map before the update (src/orcvio.cpp:2262-2268, 2325-2327)
    ptr = [0]
    cl, zz, zv = [], [], []
    for j in np.flatnonzero(keep):
        lo, hi = int(w.obs_ptr[j]), int(w.obs_ptr[j + 1])
        cl += list(w.obs_clone[lo:hi]); zz += list(w.obs_z[lo:hi]); zv += list(w.obs_zvel[lo:hi])
        ptr.append(len(cl))
    wk = dataclasses.replace(w, p_w=np.ascontiguousarray(tri['p_w'][keep]), obs_ptr=np.array(ptr, np.int32),
                             obs_clone=np.array(cl, np.int32), obs_z=np.array(zz).reshape(-1, 2), obs_zvel=np.array(zv).reshape(-1, 2))
    ref = oracle.msckf_update(wk)
    upd.upload(w, without_positions=True)
    with pytest.raises(capi.MsckfError):
        upd.run_update()           # positions missing: refused
    upd.triangulate_uploaded()
    upd.run_update()
    upd.sync()
    got = upd.download()
    assert np.array_equal(got['accept'][keep], ref['accept'])
    assert not got['accept'][~keep].any()
    assert rel(got['dx'], ref['dx']) < TOL
    assert rel(got['P_new'], ref['P_new']) < TOL

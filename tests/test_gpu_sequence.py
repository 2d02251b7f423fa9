"""The updates of one frame in sequence, each on the covariance the previous one left (SURVEY.md note N7:
removeLostFeatures -> pruneImuStateBuffer -> processObjects; src/orcvio.cpp:591-594, System.cpp:551-555), with the covariance
AND its square-root factor resident in HBM between them (orcvio_msckf_cov_commit, ORCVIO_OPT_RESIDENT_FACTOR): the second and
third update skip the Cholesky factorisation of their prior.  Checked against the oracle run step by step on the host."""
import dataclasses

import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import oracle, mirror_cov as mc
from helpers import rel, subset_window, objects_update_reference

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    yield u
    u.close()


def _obj_args(win, objs, P, new_bbox):
    return (win.flags, win.N, objs, P, win.R_b2c[0], win.t_c_b[0], True, new_bbox, 0)


@pytest.mark.parametrize('N,F,nobj,new_bbox', [(10, 60, 3, False), (30, 400, 20, True), (30, 400, 20, False)])
def test_feature_update_then_object_update(upd, N, F, nobj, new_bbox):
    """BASELINE config 3 as the reference runs it: the 400-feature update, then the 20-object update on its P+ (N = 30 is
    the full size).  OrcVIO right-perturbation Jacobians for the features (config_window(3))."""
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=N, F=F, seed=4, flags=flags, track_len=None if N == 30 else (3, N), outlier_frac=0.05)
    objs = synth.make_objects(win, n_objects=nobj, seed=2, sigma_kp=0.004)
    ref1 = oracle.msckf_update(win, want_blocks=False, want_K=False)
    ref2 = objects_update_reference(win, objs, ref1['P_new'], True, new_bbox, 0)
    for factor in (True, False):
        upd._chk(upd.lib.orcvio_msckf_set_option(upd.h, 7, int(factor)), 'set_option')
        try:
            upd.cov_set(win.P)
            got1 = upd.update_features(win, resident_cov=True, want_P=False)   # P+ stays in HBM
            assert np.array_equal(got1['accept'], ref1['accept']) and rel(got1['dx'], ref1['dx']) < TOL
            upd.cov_commit()
            assert rel(upd.cov_get(), ref1['P_new']) < TOL
            got2 = upd.update_object_tracks(*_obj_args(win, objs, None, new_bbox))   # prior = resident P+ (and its factor)
        finally:
            upd._chk(upd.lib.orcvio_msckf_set_option(upd.h, 7, 1), 'set_option')
        assert got2['accept'] == ref2['accept']
        assert abs(got2['gamma'] - ref2['gamma']) < 1e-6 * abs(ref2['gamma'])
        assert rel(got2['dx'], ref2['dx']) < TOL and rel(got2['P_new'], ref2['P_new']) < TOL
        if ref2['accept']:
            assert got2['stats'][0] == ref2['dof']
        upd.cov_commit()
        assert rel(upd.cov_get(), ref2['P_new']) < TOL
    if not new_bbox:
        assert ref2['accept'] == 1


def test_three_updates_of_a_frame_with_marginalisation(upd):
    """features -> prune update on the two oldest clones -> their rows / columns deleted -> objects on the smaller window:
    the factor kept from the first update has more columns than the window has states afterwards (M keeps its size)."""
    flags = synth.Flags(use_larvio=1)
    N = 12
    win = synth.make_window(N=N, F=80, seed=9, flags=flags, track_len=(3, N), outlier_frac=0.1)
    # update 1: every track;  update 2: the observations of clones 0 and 1 of the tracks that see both (src/orcvio.cpp:2777-2778)
    ref1 = oracle.msckf_update(win, want_blocks=False, want_K=False)
    sub = subset_window(win, [0, 1])
    both = np.diff(sub.obs_ptr) == 2
    keep = np.repeat(both, np.diff(sub.obs_ptr))
    ptr = np.concatenate([[0], np.cumsum(np.where(both, 2, 0))]).astype(np.int32)
    sub = dataclasses.replace(sub, obs_ptr=ptr, obs_clone=sub.obs_clone[keep].copy(), obs_z=sub.obs_z[keep].copy(),
                              obs_zvel=sub.obs_zvel[keep].copy())
    sub_ref = dataclasses.replace(sub, P=ref1['P_new'])
    ref2 = oracle.msckf_update(sub_ref, want_blocks=False, want_K=False)
    assert ref2['accept'].sum() > 0
    P3 = mc.remove_clones(ref2['P_new'], 22, [0, 1])
    # the window after marginalisation, and objects seen from it
    w3 = dataclasses.replace(win, R_b2w=win.R_b2w[2:].copy(), t_b_w=win.t_b_w[2:].copy(), t_fej=win.t_fej[2:].copy(),
                             R_b2c=win.R_b2c[2:].copy(), t_c_b=win.t_c_b[2:].copy(), P=P3)
    objs = synth.make_objects(w3, n_objects=4, seed=3, sigma_kp=0.004)
    oflags = dataclasses.replace(flags)
    ref3 = objects_update_reference(w3, objs, P3, True, False, 0)
    assert ref3['accept'] == 1

    upd.cov_set(win.P)
    g1 = upd.update_features(win, resident_cov=True, want_P=False)
    upd.cov_commit()
    g2 = upd.update_features(sub, resident_cov=True, want_P=False)      # prior: resident P+ and its factor
    assert np.array_equal(g2['accept'], ref2['accept']) and rel(g2['dx'], ref2['dx']) < TOL
    upd.cov_commit()
    upd.cov_remove_clones(22, [0, 1])
    assert rel(upd.cov_get(), P3) < TOL
    g3 = upd.update_object_tracks(oflags, w3.N, objs, None, w3.R_b2c[0], w3.t_c_b[0], True, False, 0)
    assert g3['accept'] == 1 and abs(g3['gamma'] - ref3['gamma']) < 1e-6 * abs(ref3['gamma'])
    assert rel(g3['dx'], ref3['dx']) < TOL and rel(g3['P_new'], ref3['P_new']) < TOL
    assert np.array_equal(g1['accept'], ref1['accept'])


def test_rejected_object_update_keeps_the_factor(upd):
    """A gated object update that is rejected must leave the covariance AND the resident factor as they were: the next
    update on the resident prior equals the update on P itself."""
    flags = synth.Flags(use_larvio=0)
    win = synth.make_window(N=10, F=40, seed=4, flags=flags, track_len=(3, 10))
    bad = synth.make_objects(win, n_objects=2, seed=6, sigma_kp=0.2)   # 25 sigma keypoint noise -> the gate fails
    ref = oracle.msckf_update(win, want_blocks=False, want_K=False)
    win2 = dataclasses.replace(win, P=ref['P_new'])
    ref_again = oracle.msckf_update(win2, want_blocks=False, want_K=False)
    upd.cov_set(win.P)
    upd.update_features(win, resident_cov=True, want_P=False)
    upd.cov_commit()
    rej = upd.update_object_tracks(*_obj_args(win, bad, None, False))
    assert rej['accept'] == 0 and not rej['dx'].any()
    upd.cov_commit()
    assert rel(upd.cov_get(), ref['P_new']) < TOL
    again = upd.update_features(win, resident_cov=True)
    assert rel(again['dx'], ref_again['dx']) < TOL and rel(again['P_new'], ref_again['P_new']) < TOL

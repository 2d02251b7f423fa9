"""EKF-SLAM rows of the hybrid filter through the C-ABI against the literal restatement (oracle/mirror_hybrid.py)."""
import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import mirror_hybrid as mh

pytestmark = pytest.mark.gpu
TOL = 1e-6   # north_star tolerance (relative Frobenius); the figures reached are ~1e-12


def rel(a, b):
    d = np.linalg.norm(np.asarray(a) - np.asarray(b))
    return d / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    yield u
    u.close()


def compact_rows(win, slam, idp):
    """What the caller hands over: the four blocks of featureJacobian_ekf for every SLAM feature."""
    He, Ha, Hx, Hf, r = [], [], [], [], []
    for ft in slam:
        H_f, H_a, H_x, H_e, rr = mh.measurement_jacobian_ekf(win, ft, idp)
        He.append(H_e); Ha.append(H_a); Hx.append(H_x); Hf.append(H_f); r.append(rr)
    return (np.array(He), np.array(Ha), np.array(Hx), np.array(Hf).reshape(len(slam), 2, idp), np.array(r))


def run(upd, win, slam, idp, on_device=False):
    upd.set_extra_states(win.n_extra)
    upd.set_ekf_rows_mode(True)
    try:
        upd.upload(win)
        if on_device:   # the SLAM features themselves: measurementJacobian_ekf_* evaluated by k_ekf_eval
            upd.upload_slam_features(idp, slam)
        else:
            He, Ha, Hx, Hf, r = compact_rows(win, slam, idp)
            upd.upload_ekf_rows(idp, [f.anchor for f in slam], [f.state for f in slam], list(range(len(slam))), He, Ha, Hx, Hf, r,
                                z_vel=np.array([f.z_vel for f in slam]))
        upd.run_update()
        upd.sync()
        got = upd.download(want_G=True)
        got['ekf_gamma'], got['ekf_accept'] = upd.download_ekf()
    finally:
        upd.set_ekf_rows_mode(False)
        upd.set_extra_states(0)
    return got


@pytest.mark.parametrize('on_device', [False, True], ids=['rows', 'features'])
@pytest.mark.parametrize('idp', [3, 1])
@pytest.mark.parametrize('case', [dict(N=9, F=40, nf=6, flags={}), dict(N=14, F=90, nf=20, flags=dict(estimate_td=1)),
                                  dict(N=30, F=400, nf=12, flags={}), dict(N=8, F=0, nf=5, flags=dict(if_fej=1)),
                                  dict(N=11, F=30, nf=9, flags=dict(if_fej=1, estimate_td=1))])
def test_joint_update_with_slam_rows(upd, idp, case, on_device):
    fl = synth.Flags(use_larvio=1, **case['flags'])
    w0 = synth.make_window(N=case['N'], F=case['F'], seed=31 + case['nf'], track_len=None if case['F'] == 400 else (3, min(case['N'], 9)),
                           flags=fl)
    slam = synth.make_slam_features(w0, case['nf'], seed=idp, outlier_frac=0.25)
    w = synth.with_extra_states(w0, idp * len(slam), seed=7)
    ref = mh.hybrid_update(w, slam, idp)
    got = run(upd, w, slam, idp, on_device)
    assert np.array_equal(got['ekf_accept'], ref['ekf_accept'])
    assert 0 < ref['ekf_accept'].sum() < len(slam) or case['nf'] < 8   # the gate does both
    assert rel(got['ekf_gamma'], ref['ekf_gamma']) < 1e-9
    assert np.array_equal(got['accept'], ref['accept'])
    assert rel(got['dx'], ref['dx']) < TOL
    assert rel(got['P_new'], ref['P_new']) < TOL
    assert rel(got['P_new'] - w.P, ref['P_new'] - w.P) < TOL
    assert rel(got['G'], ref['G']) < TOL
    assert np.array_equal(got['P_new'], got['P_new'].T)


def test_anchor_equal_to_the_observing_state(upd):
    """state == anchor (src/orcvio.cpp:1302-1310): the 3-d row pair observes the first two parameters directly; on the
    device and through the restatement."""
    import dataclasses
    w0 = synth.make_window(N=6, F=20, seed=2, track_len=(3, 6))
    slam = synth.make_slam_features(w0, 4, seed=9)
    slam[1] = dataclasses.replace(slam[1], anchor=slam[1].state)
    w = synth.with_extra_states(w0, 3 * len(slam), seed=1)
    ref = mh.hybrid_update(w, slam, 3)
    got = run(upd, w, slam, 3, on_device=True)
    assert np.array_equal(got['ekf_accept'], ref['ekf_accept'])
    assert rel(got['ekf_gamma'], ref['ekf_gamma']) < 1e-9
    assert rel(got['dx'], ref['dx']) < TOL
    assert rel(got['P_new'], ref['P_new']) < TOL

"""GPU tests of the device-resident covariance against oracle/mirror_cov.py, and of a small
filter loop that keeps P in HBM across propagate / augment / update / marginalise."""
import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import mirror_cov as mc, oracle
from helpers import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=40, max_features=512, max_observations=16384)
    yield u
    u.close()


def _spd(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    return A @ A.T * 1e-3 + np.diag(rng.uniform(1e-4, 1e-2, n))


@pytest.mark.parametrize('leg,N', [(22, 0), (22, 7), (46, 5), (22, 30)])
def test_propagate(upd, leg, N):
    n = leg + 6 * N
    rng = np.random.default_rng(n)
    P = _spd(n, n)
    Phi = np.eye(leg) + 0.05 * rng.standard_normal((leg, leg))
    G = rng.standard_normal((leg, 12))
    Q = 1e-5 * G @ G.T
    upd.cov_set(P)
    upd.cov_propagate(Phi, Q)
    got = upd.cov_get()
    ref = mc.propagate(P, Phi, Q)
    assert got.shape == ref.shape and rel(got, ref) < 1e-14
    assert np.array_equal(got, got.T)


def test_augment_and_remove_are_exact(upd):
    P = _spd(22 + 6 * 5, 3)
    upd.cov_set(P)
    upd.cov_augment()
    got = upd.cov_get()
    assert np.array_equal(got, mc.augment(P))
    upd.cov_remove_clones(22, [1, 4])
    got2 = upd.cov_get()
    assert np.array_equal(got2, mc.remove_clones(mc.augment(P), 22, [1, 4]))
    # removing the clone that was just added gives the old matrix back
    upd.cov_set(P)
    upd.cov_augment()
    upd.cov_remove_clones(22, [5])
    assert np.array_equal(upd.cov_get(), 0.5 * (P + P.T))


@pytest.mark.parametrize('max_clones,leg,N', [(8, 22, 8), (8, 46, 8), (40, 46, 40), (18, 22, 18)])
def test_propagate_on_a_full_window(built, max_clones, leg, N):
    """the scratch of cov_propagate must hold Phi P and Phi, Q for a window filled to the handle's capacity."""
    u = capi.MsckfUpdater(device=0, max_clones=max_clones, max_features=64, max_observations=1024)
    try:
        n = leg + 6 * N
        rng = np.random.default_rng(n)
        P = _spd(n, n)
        Phi = np.eye(leg) + 0.05 * rng.standard_normal((leg, leg))
        G = rng.standard_normal((leg, 12))
        Q = 1e-5 * G @ G.T
        u.cov_set(P)
        u.cov_propagate(Phi, Q)
        assert rel(u.cov_get(), mc.propagate(P, Phi, Q)) < 1e-14
    finally:
        u.close()


def test_augment_in_front_of_the_feature_states(upd):
    """stateAugmentation with EKF-SLAM feature / nuisance states behind the clones (src/orcvio.cpp:976-1003): the new clone
    is inserted in front of them."""
    k = 9
    P = _spd(22 + 6 * 4 + k, 8)
    upd.set_extra_states(k)
    try:
        upd.cov_set(P)
        upd.cov_augment()
        got = upd.cov_get()
    finally:
        upd.set_extra_states(0)
    ref = mc.augment(P, rest=k)
    assert np.array_equal(got, ref)
    n = P.shape[0]
    assert np.array_equal(got[n - k + 6:, n - k + 6:], 0.5 * (P + P.T)[n - k:, n - k:])   # the feature block moved down by 6


def test_limits(upd):
    with pytest.raises(capi.MsckfError):
        upd.cov_remove_clones(22, [99])
    upd.cov_set(_spd(46 + 6 * 40, 1))   # capacity: leg_dim 46 and max_clones = 40 states
    with pytest.raises(capi.MsckfError):
        upd.cov_augment()


@pytest.mark.parametrize('prefactor', [False, True], ids=['chol_in_update', 'prefactored'])
def test_filter_loop_keeps_the_covariance_on_the_device(upd, prefactor):
    """Three frames of propagate -> augment -> update (prior = resident P) -> commit -> marginalise, with P never sent
    after the first frame; every step equals the same loop run with the oracle on the host.  prefactored: the Cholesky of the
    propagated covariance is started behind the augmentation (orcvio_msckf_cov_prefactor) and the update finds the factor
    resident -- same results."""
    rng = np.random.default_rng(5)
    N0 = 6
    w0 = synth.make_window(N=N0, F=30, seed=40, track_len=(3, 6))
    P = w0.P.copy()
    upd.cov_set(P)
    for frame in range(3):
        leg = 22
        Phi = np.eye(leg) + 0.01 * rng.standard_normal((leg, leg))
        G = rng.standard_normal((leg, 12))
        Q = 1e-6 * G @ G.T
        upd.cov_propagate(Phi, Q)
        P = mc.propagate(P, Phi, Q)
        upd.cov_augment()
        P = mc.augment(P)
        if prefactor:
            upd.cov_prefactor()
        N = (P.shape[0] - leg) // 6
        w = synth.make_window(N=N, F=30, seed=41 + frame, track_len=(3, N))
        w.P[:] = P
        ref = oracle.msckf_update(w)
        upd.upload(w, resident_cov=True)
        upd.run_update()
        upd.sync()
        got = upd.download()
        assert np.array_equal(got['accept'], ref['accept'])
        assert rel(got['dx'], ref['dx']) < 1e-6 and rel(got['P_new'], ref['P_new']) < 1e-6
        upd.cov_commit()
        P = ref['P_new']
        upd.cov_remove_clones(leg, [0])
        P = mc.remove_clones(P, leg, [0])
        assert rel(upd.cov_get(), P) < 1e-6


def test_clones_to_nuisance_moves_blocks_and_keeps_the_factor(upd):
    """orcvio_msckf_cov_clones_to_nuisance: the Schmidt branch of pruneImuStateBuffer (src/orcvio.cpp:2881-2920) on the resident
    covariance equals the literal restatement; the resident square-root factor is permuted with it, so an update right after
    (which uses that factor) still equals the oracle on the permuted covariance."""
    leg = 22
    w = synth.make_window(N=8, F=40, seed=50, track_len=(3, 8))
    upd.cov_set(w.P)
    got = upd.update_features(w, resident_cov=True, want_P=False)
    upd.cov_commit()                                  # P+ and its factor resident
    P1 = oracle.msckf_update(w)['P_new']
    assert rel(got['dx'], oracle.msckf_update(w)['dx']) < 1e-6
    upd.cov_clones_to_nuisance(leg, [1, 4])
    P2 = mc.clones_to_nuisance(P1, leg, [1, 4])
    assert rel(upd.cov_get(), P2) < 1e-9
    # the window is now 6 clones + 12 nuisance columns; an MSCKF update with the resident prior and its (permuted) factor
    import dataclasses
    w2 = synth.make_window(N=6, F=25, seed=51, track_len=(3, 6))
    w2 = dataclasses.replace(w2, P=np.ascontiguousarray(P2), n_extra=12)
    upd.set_extra_states(12)
    upd.set_schmidt_states(2)
    try:
        got2 = upd.update_features(w2, resident_cov=True, want_P=True)
    finally:
        upd.set_schmidt_states(0)
        upd.set_extra_states(0)
    from oracle import mirror
    ref2 = mirror.msckf_update(w2)     # (the numpy restatement knows about extra states)
    Pn = ref2['P_new'].copy()
    Pn[-12:, -12:] = P2[-12:, -12:]
    assert rel(got2['dx'], ref2['dx']) < 1e-6
    assert rel(got2['P_new'], Pn) < 1e-6

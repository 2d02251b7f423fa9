"""GPU unit tests of the factorisation kernels against numpy (every template instantiation), through the test hooks of the
diagnostics build (liborcvio_msckf_dbg.so: the product library does not export them)."""
import numpy as np
import pytest

from orcvio_amd import capi
from helpers import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=40, max_features=64, max_observations=1024, debug_hooks=True)
    yield u
    u.close()


def _spd(n, seed, cond=1e4):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.logspace(0, -np.log10(cond), n)
    return (Q * ev) @ Q.T


@pytest.mark.parametrize('n', [5, 16, 17, 52, 64, 94, 100, 128, 142, 160, 187, 202, 208, 214, 224])
def test_potrf_register_path(upd, n):
    X = _spd(n, n)
    L, Dinv, info = capi.debug_potrf(upd, X)
    assert np.isfinite(L).all()
    assert info[0] == 0 and info[1] == 0
    assert rel(L @ L.T, X) < 1e-13
    assert np.abs(np.triu(L, 1)).max() == 0.0
    for kb in range((n + 15) // 16):
        k0, k1 = 16 * kb, min(n, 16 * kb + 16)
        D = L[k0:k1, k0:k1]
        assert rel(Dinv[kb][:k1 - k0, :k1 - k0] @ D, np.eye(k1 - k0)) < 1e-10


@pytest.mark.parametrize('n', [40, 202, 250, 286])
def test_potrf_lds_panel_path(upd, n):
    X = _spd(n, n + 1)
    L, Dinv, info = capi.debug_potrf(upd, X, force_lds_path=True)
    assert rel(L @ L.T, X) < 1e-13


@pytest.mark.parametrize('n', [30, 120, 202])
def test_potrf_semidefinite_zero_variance_states(upd, n):
    """Zero rows/columns (states that are not estimated) give zero columns of the factor."""
    X = _spd(n, 3 * n)
    dead = [3, 4, 17, n - 2]
    X[dead, :] = 0.0
    X[:, dead] = 0.0
    L, Dinv, info = capi.debug_potrf(upd, X, tol_rel=8 * 2.2e-16)
    assert info[0] == len(dead) and info[1] == 0
    assert rel(L @ L.T, X) < 1e-13
    assert not L[:, dead].any()


@pytest.mark.parametrize('n,nrhs', [(20, 7), (100, 33), (202, 203), (214, 215), (187, 16), (250, 40)])
def test_trsm(upd, n, nrhs):
    X = _spd(n, 7 * n)
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, nrhs))
    Z = capi.debug_trsm(upd, X, B)
    L = np.linalg.cholesky(X)
    assert rel(Z, np.linalg.solve(L, B)) < 1e-10


@pytest.mark.parametrize('la', [0, 2, 3])
@pytest.mark.parametrize('n,nrhs', [(5, 3), (16, 16), (17, 40), (94, 95), (96, 112), (100, 33), (142, 143), (160, 161), (176, 177), (187, 203),
                                    (192, 208), (202, 203), (208, 17), (224, 225)])
def test_fused_potrf_solve(upd, n, nrhs, la):
    """chol(X) and Z = L^-1 B in one launch: k_potrf_solve (la = 0) and k_potrf_solve_la (the trailing update spread over far
    workgroups, look-ahead 2 and 3) against numpy; no hand-off may be lost."""
    X = _spd(n, 11 * n + la, cond=1e6)
    rng = np.random.default_rng(n + nrhs)
    B = rng.standard_normal((n, nrhs))
    r = capi.debug_potrf_solve(upd, X, B, la=la)
    assert r['info'].tolist() == [0, 0, 0]
    L = np.linalg.cholesky(X)
    assert rel(r['L'], L) < 1e-11
    assert np.abs(np.triu(r['L'], 1)).max() == 0.0
    assert rel(r['Z'], np.linalg.solve(L, B)) < 1e-9


def test_fused_potrf_solve_lookahead_repeats(upd):
    """The look-ahead form two hundred times on one matrix: every launch identical to the first (no hand-off read early)."""
    n, nrhs = 187, 203
    X = _spd(n, 77, cond=1e5)
    B = np.random.default_rng(3).standard_normal((n, nrhs))
    first = capi.debug_potrf_solve(upd, X, B, la=3)
    old = capi.debug_potrf_solve(upd, X, B, la=0)   # (the panels are applied in the same order: the two kernels agree bit for bit)
    assert np.array_equal(old['Z'], first['Z']) and np.array_equal(old['L'], first['L'])
    for it in range(200):
        r = capi.debug_potrf_solve(upd, X, B, la=3 if it % 2 else 2)
        assert r['info'].tolist() == [0, 0, 0]
        assert np.array_equal(r['Z'], first['Z']) and np.array_equal(r['L'], first['L'])

"""CPU tests of oracle/mirror_cov.py (covariance bookkeeping of processModel / stateAugmentation / pruneImuStateBuffer):
properties that follow from the reference's formulas (src/orcvio.cpp:800-816, 962-1010, 2935-2951)."""
import numpy as np

from oracle import mirror_cov as mc


def _spd(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n)) / np.sqrt(n)
    return A @ A.T + 1e-3 * np.eye(n)


def test_propagate_is_the_congruence_with_blockdiag_phi():
    leg, N = 22, 4
    n = leg + 6 * N
    rng = np.random.default_rng(0)
    P = _spd(n, 1)
    Phi = np.eye(leg) + 0.1 * rng.standard_normal((leg, leg))
    Q = _spd(leg, 2) * 1e-4
    F = np.eye(n)
    F[:leg, :leg] = Phi
    Qf = np.zeros((n, n))
    Qf[:leg, :leg] = Q
    ref = F @ P @ F.T + Qf
    got = mc.propagate(P, Phi, Q)
    assert np.allclose(got, 0.5 * (ref + ref.T), rtol=0, atol=1e-14)
    assert np.linalg.eigvalsh(got).min() > 0


def test_augment_is_J_P_Jt_and_remove_undoes_it():
    P = _spd(22 + 12, 3)
    A = mc.augment(P)
    n = P.shape[0]
    J = np.zeros((6, n)); J[:3, :3] = np.eye(3); J[3:, 6:9] = np.eye(3)
    full = np.vstack([np.eye(n), J])
    assert np.allclose(A, full @ P @ full.T, atol=1e-15)
    assert np.array_equal(mc.remove_clones(A, 22, [2]), 0.5 * (P + P.T))
    # removing an inner clone keeps the others in order
    B = mc.remove_clones(A, 22, [0])
    assert B.shape == (n, n) and np.array_equal(B[22:28, 22:28], A[28:34, 28:34])


def test_augment_with_rest_rows_inserts_in_front_of_them():
    k = 5
    P = _spd(22 + 12 + k, 4)
    n = P.shape[0]
    A = mc.augment(P, rest=k)
    J = np.zeros((6, n)); J[:3, :3] = np.eye(3); J[3:, 6:9] = np.eye(3)
    T = np.vstack([np.eye(n)[:n - k], J, np.eye(n)[n - k:]])   # old poses, the new clone, then the rest
    assert np.allclose(A, T @ P @ T.T, atol=1e-15)
    assert np.array_equal(mc.augment(P, rest=0), mc.augment(P))


def test_clones_to_nuisance_is_a_symmetric_permutation():
    """The block moves of the Schmidt branch of pruneImuStateBuffer (src/orcvio.cpp:2881-2920), restated literally, are the
    symmetric permutation that takes the listed clones' blocks to the end in the listed order."""
    rng = np.random.default_rng(3)
    leg, N, extra = 22, 6, 9
    n = leg + 6 * N + extra
    A = rng.standard_normal((n, n))
    P = A @ A.T
    for idx in ([1], [0, 3], [2, 3, 5]):
        got = mc.clones_to_nuisance(P, leg, idx)
        moved = [leg + 6 * c + e for c in idx for e in range(6)]
        order = [i for i in range(n) if i not in moved] + moved
        assert np.array_equal(got, P[np.ix_(order, order)])

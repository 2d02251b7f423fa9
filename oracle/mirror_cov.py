"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the covariance bookkeeping around the update.

  propagate       OrcVIO::processModel, src/orcvio.cpp:800-816   P_LL <- Phi P_LL Phi^T + Q, cross terms, symmetrise
  augment         OrcVIO::stateAugmentation, :962-1010 (rest = EKF-SLAM feature + nuisance states behind the clones)
                  J = [I3 at cols 0:3 ; I3 at cols 6:9] -> the new clone copies the IMU (theta, p) covariance
  remove_clones   OrcVIO::pruneImuStateBuffer, :2935-2951 (the non-Schmidt branch): rows/cols of the pruned clones deleted
Parity unpinned (no reference test touches these lines).  Nothing under orcvio_amd/ may import this module.
"""
import numpy as np


def propagate(P, Phi, Q):
    leg = Phi.shape[0]
    P = P.copy()
    P[:leg, :leg] = Phi @ P[:leg, :leg] @ Phi.T + Q
    if P.shape[0] > leg:
        P[:leg, leg:] = Phi @ P[:leg, leg:]
        P[leg:, :leg] = P[leg:, :leg] @ Phi.T
    return 0.5 * (P + P.T)


def augment(P, rest=0):
    """rest = feature_rows + nui_rows (:976-981): the new clone is inserted in front of those states (:988-1003)."""
    n = P.shape[0]
    pose = n - rest
    J = np.zeros((6, n))
    J[0:3, 0:3] = np.eye(3)
    J[3:6, 6:9] = np.eye(3)
    P12 = J @ P
    P11 = P12 @ J.T
    out = np.zeros((n + 6, n + 6))
    old = np.r_[np.arange(pose), np.arange(pose + 6, n + 6)]
    new = np.arange(pose, pose + 6)
    out[np.ix_(old, old)] = P
    out[np.ix_(new, old)] = P12
    out[np.ix_(old, new)] = P12.T
    out[np.ix_(new, new)] = P11
    return 0.5 * (out + out.T)


def remove_clones(P, leg, clone_indices):
    keep = np.ones(P.shape[0], bool)
    for c in clone_indices:
        keep[leg + 6 * c: leg + 6 * c + 6] = False
    return P[np.ix_(keep, keep)].copy()


def clones_to_nuisance(P, leg, clone_indices):
    """Schmidt branch of pruneImuStateBuffer (src/orcvio.cpp:2881-2920), literally: for every listed clone (window ranks
    BEFORE the call, ascending) the rows / columns behind its block move up by 6 and its own block and cross terms go to the
    end; the imu state is erased from the window afterwards, so the ranks of the later ones drop by one."""
    P = P.copy()
    done = 0
    for c in clone_indices:
        rows = cols = P.shape[0]
        start = leg + 6 * (c - done)
        end = start + 6
        if end < rows:
            P_ss = P[start:end, start:end].copy()
            P_os_1 = P[start:end, :start].copy()
            P_os_2 = P[start:end, end:].copy()
            P[start:start + rows - end, :] = P[end:, :].copy()
            P[:, start:start + cols - end] = P[:, end:].copy()
            P[rows - 6:, cols - 6:] = P_ss
            P[rows - 6:, :start] = P_os_1
            P[rows - 6:, start:start + cols - end] = P_os_2
            P[:start, cols - 6:] = P_os_1.T
            P[start:start + rows - end, cols - 6:] = P_os_2.T
        done += 1
    return P

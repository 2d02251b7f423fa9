"""oracle/mirror_hybrid.py (EKF-SLAM rows of the hybrid filter) against central differences of the measurement model and
against the structure of the joint update.  CPU only."""
import dataclasses

import numpy as np
import pytest

from orcvio_amd import synth
from oracle import mirror, mirror_hybrid as mh


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def _project(win, ft, idp, d_anchor, d_state, d_ext, d_feat):
    """pi(p_ck) with error-state increments applied: clones [dtheta, dp] with the LARVIO (left) convention of the
    feature rows (tests/test_oracle.py::_project), extrinsics [dtheta_e, dt_e], feature parameters additive.  The
    world point is rebuilt from the anchor camera and the inverse-depth parameters, as measurementUpdate_hybrid does
    after the update (src/orcvio.cpp:1857-1885)."""
    def clone(i, d):
        return mirror.so3_exp(d[:3]) @ win.R_b2w[i], win.t_b_w[i] + d[3:]
    q = mirror.small_angle_quaternion(d_ext[:3])
    R_b2c = win.R_b2c[0] @ mirror.quat_to_rot_hamilton(q).T
    t_c_b = win.t_c_b[0] + d_ext[3:]
    Ra, ta = clone(ft.anchor, d_anchor)
    Rk, tk = clone(ft.state, d_state)
    if idp == 3:
        fc = ft.inv_param + d_feat
        p_ca = np.array([fc[0] / fc[2], fc[1] / fc[2], 1.0 / fc[2]])
    else:
        rho = ft.inv_depth + d_feat[0]
        p_ca = np.array([ft.obs_anchor[0] / rho, ft.obs_anchor[1] / rho, 1.0 / rho])
    p_w = Ra @ (R_b2c.T @ p_ca + t_c_b) + ta
    pc = R_b2c @ (Rk.T @ (p_w - tk) - t_c_b)
    return pc[:2] / pc[2]


@pytest.mark.parametrize('idp', [3, 1])
def test_ekf_rows_against_central_differences(idp):
    """H_a, H_x, H_e, H_f of measurementJacobian_ekf_{3,1}didp are d pi / d(error state) when the feature's world position
    is the one its anchor camera and inverse-depth parameters define."""
    w = synth.make_window(N=6, F=2, seed=3, track_len=6, flags=synth.Flags(use_larvio=1))
    slam = synth.make_slam_features(w, 8, seed=4)
    eps = 1e-6
    for ft in slam:
        # make the stored world position exactly consistent with the parametrisation (the Jacobians assume it)
        R_c2w = w.R_b2w[ft.anchor] @ w.R_b2c[ft.anchor].T
        t_c_w = w.t_b_w[ft.anchor] + w.R_b2w[ft.anchor] @ w.t_c_b[ft.anchor]
        pc = np.array([ft.inv_param[0] / ft.inv_param[2], ft.inv_param[1] / ft.inv_param[2], 1 / ft.inv_param[2]])
        ft = dataclasses.replace(ft, p_w=R_c2w @ pc + t_c_w)
        H_f, H_a, H_x, H_e, r = mh.measurement_jacobian_ekf(w, ft, idp)
        assert np.allclose(r, ft.z - _project(w, ft, idp, np.zeros(6), np.zeros(6), np.zeros(6), np.zeros(3)), atol=1e-12)
        num = np.zeros((2, 21))
        for c in range(18 + idp):
            d = np.zeros(21)
            d[c] = eps
            zp = _project(w, ft, idp, d[0:6], d[6:12], d[12:18], d[18:21])
            zm = _project(w, ft, idp, -d[0:6], -d[6:12], -d[12:18], -d[18:21])
            num[:, c] = (zp - zm) / (2 * eps)
        assert np.allclose(H_a, num[:, 0:6], atol=5e-8)
        assert np.allclose(H_x, num[:, 6:12], atol=5e-8)
        assert np.allclose(H_e, num[:, 12:18], atol=5e-8)
        assert np.allclose(H_f, num[:, 18:18 + idp], atol=5e-8)


def test_anchor_frame_observation_3didp():
    """state == anchor (:1302-1310): the row pair observes the first two inverse-depth parameters directly."""
    w = synth.make_window(N=5, F=2, seed=1, track_len=5)
    ft = synth.make_slam_features(w, 1, seed=0)[0]
    ft = dataclasses.replace(ft, anchor=ft.state)
    H_f, H_a, H_x, H_e, _ = mh.measurement_jacobian_ekf(w, ft, 3)
    assert np.array_equal(H_f, np.array([[1.0, 0, 0], [0, 1.0, 0]]))
    assert not H_a.any() and not H_x.any() and not H_e.any()


@pytest.mark.parametrize('idp', [3, 1])
def test_joint_update_is_the_sequential_update(idp):
    """One update with [H_msckf; H_ekf] (measurementUpdate_hybrid) equals the MSCKF update followed by the update with
    the SLAM rows linearised at the SAME prior: information adds.  Also: rejected SLAM features change nothing."""
    w0 = synth.make_window(N=7, F=25, seed=11, track_len=(3, 7))
    slam = synth.make_slam_features(w0, 7, seed=2, outlier_frac=0.3)
    w = synth.with_extra_states(w0, idp * len(slam), seed=3)
    joint = mh.hybrid_update(w, slam, idp)
    assert 0 < joint['ekf_accept'].sum() < len(slam)
    s2 = w.flags.noise_feature ** 2
    first = mirror.msckf_update(w)
    H2 = np.vstack([H for (H, _), a in zip(joint['ekf_rows'], joint['ekf_accept']) if a])
    r2 = np.concatenate([r for (_, r), a in zip(joint['ekf_rows'], joint['ekf_accept']) if a])
    P1 = first['P_new']
    S = H2 @ P1 @ H2.T + s2 * np.eye(H2.shape[0])
    K2 = np.linalg.solve(S, H2 @ P1).T
    dx = first['dx'] + K2 @ (r2 - H2 @ first['dx'])
    P2 = (np.eye(w.n) - K2 @ H2) @ P1
    assert rel(dx, joint['dx']) < 1e-8
    assert rel(0.5 * (P2 + P2.T), joint['P_new']) < 1e-9
    # gate values are those of the individual row pairs against the prior
    for (H, r), g in zip(joint['ekf_rows'], joint['ekf_gamma']):
        assert abs(g - r @ np.linalg.solve(H @ w.P @ H.T + s2 * np.eye(2), r)) <= 1e-12 * max(1.0, g)


@pytest.mark.parametrize('idp', [3, 1])
def test_new_feature_update_does_not_depend_on_the_bases(idp):
    """The reference takes V from a full-U SVD and U from SPQR's Q (src/orcvio.cpp:2420-2431); the restatement takes both
    from one complete QR.  The update only depends on the two SUBSPACES: rotating either basis changes nothing; and the
    augmented covariance stays symmetric positive semi-definite."""
    w0 = synth.make_window(N=9, F=30, seed=5, track_len=(3, 9))
    slam = synth.make_slam_features(w0, 5, seed=1)
    w = synth.with_extra_states(w0, idp * len(slam), seed=2)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 4, seed=3)]
    ref = mh.hybrid_update_full(w, slam, new, idp)
    acc, H_top, r_top, H_1, H_2, r_1 = mh.split_new_rows(w, new, idp)
    rng = np.random.default_rng(0)
    Qv, _ = np.linalg.qr(rng.standard_normal((H_top.shape[0], H_top.shape[0])))
    Qu, _ = np.linalg.qr(rng.standard_normal((H_2.shape[0], H_2.shape[0])))
    s2 = w.flags.noise_feature ** 2
    base = mirror.msckf_update(w)
    blocks = [b for b, a in zip(base['blocks'], base['accept']) if a]
    rs = [b for b, a in zip(base['rs'], base['accept']) if a]
    for idx, ft in enumerate(slam):
        H, r = mh.feature_jacobian_ekf(w, ft, idx, idp)
        if ref['ekf_accept'][idx]:
            blocks.append(H); rs.append(r)
    blocks.append(Qv.T @ H_top); rs.append(Qv.T @ r_top)
    H_o = np.vstack(blocks); r_o = np.concatenate(rs)
    dx_leg, _, P_upd = mirror.measurement_update(H_o, r_o, w.P, s2)
    dx, P = mh.augment_after_update(P_upd, dx_leg, Qu.T @ H_1, Qu.T @ H_2, Qu.T @ r_1, s2)
    assert rel(dx, ref['dx']) < 1e-9
    assert rel(P, ref['P_new']) < 1e-9
    assert np.linalg.eigvalsh(ref['P_new']).min() > -1e-12
    assert ref['P_new'].shape[0] == w.n + idp * len(acc)


@pytest.mark.parametrize('idp', [3, 1])
def test_library_host_arithmetic_for_entering_features(built, idp):
    """orcvio_msckf_new_feature_rows / orcvio_msckf_augment_state (host arithmetic inside the library, no device) against
    the restatement: bases differ (one Householder QR per feature there, one complete QR of the stacked H_f here), so the
    comparison is on what the update depends on -- the Gram data of the V part, H_2^-1 H_1, H_2^-1 r_1, (H_2^T H_2)^-1 --
    and on the augmented state itself."""
    from orcvio_amd import capi
    from scipy.linalg import block_diag
    w0 = synth.make_window(N=9, F=30, seed=5, track_len=(3, 9), flags=synth.Flags(use_larvio=1, estimate_td=1))
    slam = synth.make_slam_features(w0, 5, seed=1)
    w = synth.with_extra_states(w0, idp * len(slam), seed=2)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 4, seed=3)]
    acc, H_top, r_top, H_1, H_2, r_1 = mh.split_new_rows(w, new, idp)
    feats = [new[i] for i in acc]
    gH_top, gr_top, gH_1, gH_2, gr_1 = capi.new_feature_rows(w, idp, feats)
    assert gH_top.shape == H_top.shape
    assert rel(gH_top.T @ gH_top, H_top.T @ H_top) < 1e-10
    assert rel(gH_top.T @ gr_top, H_top.T @ r_top) < 1e-9
    assert abs(gr_top @ gr_top - r_top @ r_top) < 1e-10 * max(1.0, r_top @ r_top)
    G2 = block_diag(*gH_2)
    assert np.allclose(G2, np.triu(G2))
    assert rel(np.linalg.solve(G2, gH_1), np.linalg.solve(H_2, H_1)) < 1e-9
    assert rel(np.linalg.solve(G2, gr_1), np.linalg.solve(H_2, r_1)) < 1e-9
    assert rel(np.linalg.inv(G2.T @ G2), np.linalg.inv(H_2.T @ H_2)) < 1e-9
    # the augmentation, from the restatement's update of the legacy state
    ref = mh.hybrid_update_full(w, slam, new, idp)
    dx_new, P_aug = capi.augment_state(idp, gH_1, gH_2, gr_1, w.flags.noise_feature ** 2, ref['dx_leg'], ref['P_upd'])
    assert rel(np.concatenate([ref['dx_leg'], dx_new]), ref['dx']) < 1e-9
    assert rel(P_aug, ref['P_new']) < 1e-9


# ---- Schmidt nuisance states (use_schmidt) --------------------------------------------------------------------------------
def test_nuisance_anchor_rows_against_central_differences():
    """A SLAM feature anchored at a Schmidt nuisance state (anchor index N + j): the same four blocks with the nuisance pose
    as anchor pose (src/orcvio.cpp:1247-1256); H_a lands in the state's columns of the nuisance block (:1591-1606)."""
    w0 = synth.make_window(N=6, F=2, seed=3, track_len=6, flags=synth.Flags(use_larvio=1))
    w = synth.with_nuisance_states(synth.with_extra_states(w0, 3 * 4, seed=1), 2, seed=2)
    slam = synth.make_slam_features(w, 6, seed=4, nui_frac=1.0)
    assert all(ft.anchor >= w.N for ft in slam)
    eps = 1e-6
    for idx, ft in enumerate(slam[:4]):
        j = ft.anchor - w.N
        Ra, ta = w.nui['R_b2w'][j], w.nui['t_b_w'][j]
        R_c2w = Ra @ w.R_b2c[0].T
        t_c_w = ta + Ra @ w.t_c_b[0]
        pc = np.array([ft.inv_param[0] / ft.inv_param[2], ft.inv_param[1] / ft.inv_param[2], 1 / ft.inv_param[2]])
        ft = dataclasses.replace(ft, p_w=R_c2w @ pc + t_c_w)
        H_f, H_a, H_x, H_e, r = mh.measurement_jacobian_ekf(w, ft, 3)

        def project(d_anchor):
            Rad = mirror.so3_exp(d_anchor[:3]) @ Ra
            p_w = Rad @ (w.R_b2c[0].T @ pc + w.t_c_b[0]) + ta + d_anchor[3:]
            k = ft.state
            q = w.R_b2c[k] @ (w.R_b2w[k].T @ (p_w - w.t_b_w[k]) - w.t_c_b[k])
            return q[:2] / q[2]
        num = np.zeros((2, 6))
        for c in range(6):
            d = np.zeros(6)
            d[c] = eps
            num[:, c] = (project(d) - project(-d)) / (2 * eps)
        assert np.allclose(H_a, num, atol=5e-8)
        H, _ = mh.feature_jacobian_ekf(w, ft, idx, 3)
        c0 = w.n - 6 * w.n_nui + 6 * j
        assert np.array_equal(H[:, c0:c0 + 6], H_a)
        assert np.count_nonzero(H[:, w.flags.leg_dim: w.flags.leg_dim + 6 * (w.N - 1)]) == 0   # no window clone but the observing one


def test_schmidt_update_keeps_the_nuisance_block_and_inserts_new_states_in_front_of_it():
    """:1740-1751 / :1893-1935: the nuisance block of the covariance survives the update unchanged, everything else is
    (I - K H) P; new feature states go in front of the nuisance rows, i.e. the augmented covariance is the no-Schmidt one
    with the nuisance rows / columns moved behind the new states."""
    w0 = synth.make_window(N=8, F=30, seed=11, track_len=(3, 8), flags=synth.Flags(use_larvio=1))
    slam = synth.make_slam_features(w0, 5, seed=2)
    w1 = synth.with_extra_states(w0, 3 * len(slam), seed=3)
    w = synth.with_nuisance_states(w1, 2, seed=4)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 3, seed=6)]
    got = mh.hybrid_update_full(w, slam, new, 3)
    m = 12
    n = w.n
    sz = 3 * len(got['new_accept'])
    assert sz > 0
    assert np.array_equal(got['P_upd'][-m:, -m:], w.P[-m:, -m:])
    plain = mh.hybrid_update_full(dataclasses.replace(w, nui=None), slam, new, 3)      # same numbers, no Schmidt treatment
    assert rel(got['P_upd'][:n - m, :], plain['P_upd'][:n - m, :]) < 1e-12
    order = list(range(n - m)) + list(range(n, n + sz)) + list(range(n - m, n))
    # (the plain augmentation starts from the plain P_upd: compare against the one built from the Schmidt P_upd)
    dx_ref, P_ref = mh.augment_after_update(got['P_upd'], got['dx_leg'], *_blocks(w, new, got), w.flags.noise_feature ** 2)
    assert rel(got['P_new'], P_ref[np.ix_(order, order)]) < 1e-12
    assert rel(got['dx'], dx_ref) < 1e-12


def _blocks(w, new, got):
    acc, H_top, r_top, H_1, H_2, r_1 = mh.split_new_rows(w, new, 3)
    assert acc == got['new_accept']
    return H_1, H_2, r_1


@pytest.mark.parametrize('idp', [3, 1])
def test_library_augmentation_in_front_of_the_nuisance_block(built, idp):
    """orcvio_msckf_augment_state_nuisance (host arithmetic of the library, no device) against the literal block moves of
    src/orcvio.cpp:1920-1935 in the restatement."""
    from orcvio_amd import capi
    w0 = synth.make_window(N=8, F=30, seed=11, track_len=(3, 8), flags=synth.Flags(use_larvio=1))
    slam = synth.make_slam_features(w0, 5, seed=2)
    w = synth.with_nuisance_states(synth.with_extra_states(w0, idp * len(slam), seed=3), 2, seed=4)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 3, seed=6)]
    ref = mh.hybrid_update_full(w, slam, new, idp)
    assert len(ref['new_accept']) > 0
    gH_top, gr_top, gH_1, gH_2, gr_1 = capi.new_feature_rows(w, idp, [new[i] for i in ref['new_accept']])
    dx_new, P_aug = capi.augment_state_nuisance(idp, 6 * w.n_nui, gH_1, gH_2, gr_1, w.flags.noise_feature ** 2, ref['dx_leg'], ref['P_upd'])
    assert rel(np.concatenate([ref['dx_leg'], dx_new]), ref['dx']) < 1e-9
    assert rel(P_aug, ref['P_new']) < 1e-9


def test_reference_literal_h2_ldlt_differs_only_for_three_parameters():
    """`H_2.ldlt().solve(..)` of the reference (src/orcvio.cpp:1826-1827) factors the LOWER triangle of an upper-triangular H_2 --
    its diagonal.  For feature_idp_dim = 1 (every shipped configuration) H_2 is a stack of 1 x 1 blocks and nothing changes; for
    3 parameters the literal tail is another update of the entering states (mirror: ref_ldlt; library:
    orcvio_msckf_augment_state_ref_ldlt, host arithmetic)."""
    from orcvio_amd import capi
    for idp in (1, 3):
        w0 = synth.make_window(N=8, F=30, seed=3, track_len=(3, 8), flags=synth.Flags(use_larvio=1))
        w = synth.with_extra_states(w0, 0, seed=1)
        new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 3, seed=2)]
        acc, H_top, r_top, H_1, H_2, r_1 = mh.split_new_rows(w, new, idp)
        assert len(acc) > 0
        s2 = w.flags.noise_feature ** 2
        dx_leg = np.linspace(-1e-2, 1e-2, w.n)
        a_dx, a_P = mh.augment_after_update(w.P, dx_leg, H_1, H_2, r_1, s2)
        b_dx, b_P = mh.augment_after_update(w.P, dx_leg, H_1, H_2, r_1, s2, ref_ldlt=True)
        blocks = np.stack([H_2[idp * j:idp * j + idp, idp * j:idp * j + idp] for j in range(len(acc))])
        c_dx, c_P = capi.augment_state_nuisance(idp, 0, H_1, blocks, r_1, s2, dx_leg, w.P, ref_ldlt=True)
        d_dx, d_P = capi.augment_state(idp, H_1, blocks, r_1, s2, dx_leg, w.P)
        e = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)
        assert e(np.concatenate([dx_leg, c_dx]), b_dx) < 1e-12 and e(c_P, b_P) < 1e-12
        assert e(np.concatenate([dx_leg, d_dx]), a_dx) < 1e-12 and e(d_P, a_P) < 1e-12
        if idp == 1:
            assert e(b_dx, a_dx) < 1e-14 and e(b_P, a_P) < 1e-14
        else:
            assert e(b_dx, a_dx) > 1e-3

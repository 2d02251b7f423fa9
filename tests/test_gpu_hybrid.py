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
                                  dict(N=11, F=30, nf=9, flags=dict(if_fej=1, estimate_td=1)),
                                  # the OrcVIO Jacobians of the MSCKF rows beside the (always LARVIO-form) rows of the in-state features:
                                  # kitti_raw.yaml's shipped set (:143-148: right perturbation, sigma 1) and the left-perturbation variant
                                  dict(N=20, F=120, nf=12, flags=dict(use_larvio=0, use_left_perturbation=0, noise_feature=1.0, discard_large_update=1), sigma_px=0.008),
                                  dict(N=12, F=60, nf=8, flags=dict(use_larvio=0, use_left_perturbation=1)),
                                  dict(N=10, F=40, nf=7, flags=dict(use_larvio=0, use_left_perturbation=0, estimate_td=1)),
                                  dict(N=10, F=40, nf=7, flags=dict(use_larvio=0, use_left_perturbation=1, if_fej=1))],
                         ids=lambda c: 'N%d_F%d_%s' % (c['N'], c['F'], '_'.join('%s%s' % (k[:6], v) for k, v in c['flags'].items()) or 'larvio'))
def test_joint_update_with_slam_rows(upd, idp, case, on_device):
    fl = synth.Flags(**dict(dict(use_larvio=1), **case['flags']))
    w0 = synth.make_window(N=case['N'], F=case['F'], seed=31 + case['nf'], track_len=None if case['F'] == 400 else (3, min(case['N'], 9)),
                           flags=fl, sigma_px=case.get('sigma_px'))
    slam = synth.make_slam_features(w0, case['nf'], seed=idp, outlier_frac=0.25, sigma_px=case.get('sigma_px'))
    w = synth.with_extra_states(w0, idp * len(slam), seed=7)
    ref = mh.hybrid_update(w, slam, idp)
    got = run(upd, w, slam, idp, on_device)
    assert np.array_equal(got['ekf_accept'], ref['ekf_accept'])
    assert 0 < ref['ekf_accept'].sum() < len(slam) or case['nf'] < 8 or fl.noise_feature == 1.0   # the gate does both (kitti_raw's sigma = 1 passes everything)
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


JAC = {'larvio': dict(use_larvio=1), 'orcvio_right': dict(use_larvio=0, use_left_perturbation=0), 'orcvio_left': dict(use_larvio=0, use_left_perturbation=1)}


@pytest.mark.parametrize('jac', list(JAC))
@pytest.mark.parametrize('idp', [3, 1])
def test_frame_with_new_slam_features(upd, idp, jac):
    """A frame in which NEW SLAM features enter the state, either parametrisation, through library calls only
    (INTEGRATION.md 7b): orcvio_msckf_gate_tracks (their MSCKF gate on the device), orcvio_msckf_new_feature_rows
    (featureJacobian_ekf_new and the W = [V | U] split, host arithmetic), the V-part rows as dense rows under the MSCKF
    tracks and the rows of the existing SLAM features in ONE device update, orcvio_msckf_augment_state for the H_1 / H_2
    tail (src/orcvio.cpp:1811-1947).  Result: the reference's full hybrid update (oracle.hybrid_update_full).  Under each of the
    three Jacobian conventions of the MSCKF rows (config/euroc.yaml:114-118, config/kitti_raw.yaml:143-148)."""
    w0 = synth.make_window(N=10, F=60, seed=17, track_len=(3, 10), flags=synth.Flags(estimate_td=1, **JAC[jac]))
    slam = synth.make_slam_features(w0, 7, seed=5, outlier_frac=0.25)
    w = synth.with_extra_states(w0, idp * len(slam), seed=4)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 5, seed=9, outlier_frac=0.5)]
    ref = mh.hybrid_update_full(w, slam, new, idp)
    assert 0 < len(ref['new_accept']) < len(new)
    # --- the MSCKF gate of the features that want to enter (:2361-2367), on the device: their tracks against the prior
    import dataclasses
    ptr, cl, zz, zv = [0], [], [], []
    for ft in new:
        for (k, z, v) in ft.obs:
            cl.append(k); zz.append(z); zv.append(v)
        ptr.append(len(cl))
    wg = dataclasses.replace(w, p_w=np.ascontiguousarray([ft.p_w for ft in new]), obs_ptr=np.asarray(ptr, dtype=np.int32),
                             obs_clone=np.asarray(cl, dtype=np.int32), obs_z=np.ascontiguousarray(zz).reshape(-1, 2),
                             obs_zvel=np.ascontiguousarray(zv).reshape(-1, 2))
    upd.set_extra_states(w.n_extra)
    try:
        g_gamma, g_accept = upd.gate_tracks(wg)
    finally:
        upd.set_extra_states(0)
    assert [i for i in range(len(new)) if g_accept[i]] == ref['new_accept']
    for i, ft in enumerate(new):
        assert abs(g_gamma[i] - mh.msckf_gate_of_feature(w, ft)[0]) < 1e-9 * max(1.0, g_gamma[i])
    # --- rows of the features that passed and their W split: host arithmetic of the library (the restatement's own
    #     split is compared with it in tests/test_oracle_hybrid.py)
    acc = [i for i in range(len(new)) if g_accept[i]]
    H_top, r_top, H_1, H_2, r_1 = capi.new_feature_rows(w, idp, [new[i] for i in acc])
    # --- device: one joint update with everything that has no column in the new states
    upd.set_extra_states(w.n_extra)
    upd.set_ekf_rows_mode(True)
    try:
        upd.upload(w)
        upd.upload_slam_features(idp, slam)
        upd.upload_dense_rows(H_top, r_top)
        upd.run_update()
        upd.sync()
        got = upd.download()
        _, ekf_accept = upd.download_ekf()
    finally:
        upd.set_ekf_rows_mode(False)
        upd.set_extra_states(0)
    assert np.array_equal(ekf_accept, ref['ekf_accept'])
    assert np.array_equal(got['accept'], ref['accept'])
    assert rel(got['dx'], ref['dx_leg']) < TOL
    assert rel(got['P_new'], ref['P_upd']) < TOL
    # --- the new states and the augmented covariance (host arithmetic of the library again)
    dx_new, P = capi.augment_state(idp, H_1, H_2, r_1, w.flags.noise_feature ** 2, got['dx'], got['P_new'])
    dx = np.concatenate([got['dx'], dx_new])
    assert dx.shape[0] == w.n + idp * len(acc)
    assert rel(dx, ref['dx']) < TOL
    assert rel(P, ref['P_new']) < TOL


@pytest.mark.parametrize('fej', [0, 1], ids=['nofej', 'fej'])
@pytest.mark.parametrize('idp', [3, 1])
def test_entering_features_rows_on_the_device(upd, idp, fej):
    """orcvio_msckf_upload_new_features: featureJacobian_ekf_new and the W = [V | U] split of the entering features evaluated by
    k_ekf_new from the window poses already in HBM (either parametrisation, FEJ, td) -- the U parts H_1, H_2, r_1 equal the
    library's host arithmetic, and the joint update with the V parts stacked on the device equals the one with the host's
    V-part rows handed over as dense rows; the tail (orcvio_msckf_augment_state) then gives the reference's full hybrid update."""
    w0 = synth.make_window(N=10, F=60, seed=17, track_len=(3, 10), flags=synth.Flags(use_larvio=1, estimate_td=1, if_fej=fej))
    slam = synth.make_slam_features(w0, 7, seed=5, outlier_frac=0.25)
    w = synth.with_extra_states(w0, idp * len(slam), seed=4)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 5, seed=9, outlier_frac=0.0)]
    if fej:   # first-estimate positions that differ from the current ones
        for i, ft in enumerate(new):
            ft.p_fej = ft.p_w + 0.01 * np.array([1.0, -0.5, 0.3]) * (i + 1)
    H_top, r_top, H_1, H_2, r_1 = capi.new_feature_rows(w, idp, new)
    upd.set_extra_states(w.n_extra)
    upd.set_ekf_rows_mode(True)
    try:
        upd.upload(w)
        upd.upload_slam_features(idp, slam)
        upd.upload_dense_rows(H_top, r_top)
        upd.run_update()
        upd.sync()
        ref = upd.download()
        upd.upload(w)
        upd.upload_slam_features(idp, slam)
        upd.upload_new_features(w, idp, new)
        upd.run_update()
        upd.sync()
        got = upd.download()
        g1, g2, gr = upd.download_new_feature_blocks()
    finally:
        upd.set_ekf_rows_mode(False)
        upd.set_extra_states(0)
    assert rel(g1, H_1) < 1e-12 and rel(g2, H_2) < 1e-12 and rel(gr, r_1) < 1e-12
    assert np.array_equal(got['accept'], ref['accept'])
    assert rel(got['dx'], ref['dx']) < 1e-10
    assert rel(got['P_new'], ref['P_new']) < 1e-10
    if not fej:   # ... and against the restatement of the reference's whole hybrid update
        full = mh.hybrid_update_full(w, slam, new, idp)
        if full['new_accept'] == list(range(len(new))):
            dx_new, P = capi.augment_state(idp, g1, g2, gr, w.flags.noise_feature ** 2, got['dx'], got['P_new'])
            assert rel(np.concatenate([got['dx'], dx_new]), full['dx']) < TOL
            assert rel(P, full['P_new']) < TOL


def test_dense_rows_alone(upd):
    """orcvio_msckf_upload_dense_rows without SLAM features or extra states: arbitrary caller-projected rows stacked
    under the MSCKF blocks."""
    from oracle import mirror
    w = synth.make_window(N=7, F=30, seed=8, track_len=(3, 7))
    rng = np.random.default_rng(0)
    H = np.zeros((11, w.n))
    H[:, 15:] = 0.3 * rng.standard_normal((11, w.n - 15)) * (rng.random((11, w.n - 15)) < 0.3)
    r = 0.01 * rng.standard_normal(11)
    base = mirror.msckf_update(w)
    Hs = np.vstack([b for b, a in zip(base['blocks'], base['accept']) if a] + [H])
    rs = np.concatenate([b for b, a in zip(base['rs'], base['accept']) if a] + [r])
    dx, _, Pn = mirror.measurement_update(*mirror.qr_compress(Hs, rs), w.P, w.flags.noise_feature ** 2)
    upd.upload(w)
    upd.upload_dense_rows(H, r)
    upd.run_update()
    upd.sync()
    got = upd.download()
    assert rel(got['dx'], dx) < TOL
    assert rel(got['P_new'], Pn) < TOL


def test_new_3d_features_are_msckf_tracks_in_the_joint_update(upd):
    """For the 3-parameter inverse-depth form, the V part of a NEW feature's rows (src/orcvio.cpp:2416-2436) is its
    MSCKF block: H_f(idp) = H_f(xyz) J with J invertible (same column space, same left null space), and
    featureJacobian_ekf_new differs from featureJacobian_msckf only by terms H_f(xyz) X with X common to all rows
    (anchor-pose and extrinsic dependence of the world point) -- which V^T removes.  The reference even gates the new
    feature with the MSCKF test (:2361-2367).  So the joint update needs no special input for them: list them as
    tracks.  (Not so for the 1-parameter form, which fixes the bearing in the anchor frame.)"""
    w0 = synth.make_window(N=10, F=60, seed=17, track_len=(3, 10), flags=synth.Flags(use_larvio=1, estimate_td=1))
    slam = synth.make_slam_features(w0, 7, seed=5, outlier_frac=0.25)
    w = synth.with_extra_states(w0, 3 * len(slam), seed=4)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 5, seed=9, outlier_frac=0.5)]
    ref = mh.hybrid_update_full(w, slam, new, 3)
    assert 0 < len(ref['new_accept']) < len(new)
    # the window again, with the new features appended as ordinary tracks
    import dataclasses
    obs_ptr = list(w.obs_ptr)
    obs_clone, obs_z, obs_zvel = list(w.obs_clone), list(w.obs_z), list(w.obs_zvel)
    p_w = list(w.p_w)
    for ft in new:
        p_w.append(ft.p_w)
        for (k, z, zv) in ft.obs:
            obs_clone.append(k); obs_z.append(z); obs_zvel.append(zv)
        obs_ptr.append(len(obs_clone))
    w2 = dataclasses.replace(w, p_w=np.ascontiguousarray(p_w), obs_ptr=np.asarray(obs_ptr, dtype=np.int32),
                             obs_clone=np.asarray(obs_clone, dtype=np.int32), obs_z=np.ascontiguousarray(obs_z).reshape(-1, 2),
                             obs_zvel=np.ascontiguousarray(obs_zvel).reshape(-1, 2))
    upd.set_extra_states(w2.n_extra)
    upd.set_ekf_rows_mode(True)
    try:
        upd.upload(w2)
        upd.upload_slam_features(3, slam)
        upd.run_update()
        upd.sync()
        got = upd.download()
        acc_new = [i for i in range(len(new)) if got['accept'][w.F + i]]
        # ... and the augmentation from what the feature kernel left for those tracks (T3 and the R factor of H_f):
        # measurementUpdate_hybrid's tail (:1811-1821, :1904-1947) without any Jacobian code on the caller's side
        dx_new, P_aug = upd.augment_new_features(w2, [w.F + i for i in acc_new], [new[i].anchor for i in acc_new],
                                                 [new[i].inv_param for i in acc_new], got['dx'], got['P_new'])
    finally:
        upd.set_ekf_rows_mode(False)
        upd.set_extra_states(0)
    assert acc_new == ref['new_accept']          # the device gate of the track IS the reference's gate of the new feature
    assert rel(got['dx'], ref['dx_leg']) < TOL
    assert rel(got['P_new'], ref['P_upd']) < TOL
    assert rel(np.concatenate([got['dx'], dx_new]), ref['dx']) < TOL
    assert rel(P_aug, ref['P_new']) < TOL
    assert rel(P_aug[w.n:, :], ref['P_new'][w.n:, :]) < TOL   # the new rows on their own (cross terms and P22)


def test_captured_graph_survives_a_reallocation_of_the_ekf_rows():
    """a launch graph captured for shape X must not be replayed after the EKF-row buffers it points into were
    freed and reallocated (grow to a larger SLAM-feature count, then return to shape X).  Every run is compared with a
    fresh handle that never captured anything."""
    u = capi.MsckfUpdater(device=0, max_clones=32, max_features=256, max_observations=4096)
    try:
        w0 = synth.make_window(N=8, F=30, seed=3, track_len=(3, 8))
        small = synth.make_slam_features(w0, 40, seed=1)
        big = synth.make_slam_features(w0, 70, seed=2)
        ws = synth.with_extra_states(w0, len(small), seed=1)
        wb = synth.with_extra_states(w0, len(big), seed=2)
        ref_s = mh.hybrid_update(ws, small, 1)
        ref_b = mh.hybrid_update(wb, big, 1)
        for _ in range(3):   # same signature three times: captured on the second, replayed on the third
            got = run(u, ws, small, 1, on_device=True)
            assert rel(got['dx'], ref_s['dx']) < TOL
        got = run(u, wb, big, 1, on_device=True)          # reallocates d_ekf_* (70 > capacity 64)
        assert rel(got['dx'], ref_b['dx']) < TOL and np.array_equal(got['ekf_accept'], ref_b['ekf_accept'])
        for _ in range(3):                                # back to the earlier shape
            got = run(u, ws, small, 1, on_device=True)
            assert rel(got['dx'], ref_s['dx']) < TOL and rel(got['P_new'], ref_s['P_new']) < TOL
            assert np.array_equal(got['ekf_accept'], ref_s['ekf_accept'])
        # a change of the gate probability between replays of one shape must reach the SLAM-row gate
        import dataclasses
        ws2 = dataclasses.replace(ws, flags=dataclasses.replace(ws.flags, chi2_prob=0.5))
        ref2 = mh.hybrid_update(ws2, small, 1)
        got = run(u, ws2, small, 1, on_device=True)
        assert np.array_equal(got['ekf_accept'], ref2['ekf_accept']) and rel(got['dx'], ref2['dx']) < TOL
    finally:
        u.close()


# ---- Schmidt nuisance states (use_schmidt) --------------------------------------------------------------------------------
@pytest.mark.parametrize('on_device', [False, True], ids=['rows', 'features'])
@pytest.mark.parametrize('idp', [3, 1])
def test_schmidt_update_with_nuisance_anchors(upd, idp, on_device):
    """ORCVIO_OPT_SCHMIDT_STATES: SLAM features anchored at nuisance states (their poses from orcvio_msckf_upload_nuisance_poses,
    their Jacobian block in the nuisance columns), the nuisance block of the covariance untouched by the update
    (src/orcvio.cpp:1740-1751, :1893-1902) -- against the restatement of the reference's hybrid update."""
    w0 = synth.make_window(N=10, F=60, seed=21, track_len=(3, 10), flags=synth.Flags(use_larvio=1, estimate_td=1))
    nf = 8
    w = synth.with_nuisance_states(synth.with_extra_states(w0, idp * nf, seed=4), 3, seed=5)
    slam = synth.make_slam_features(w, nf, seed=6, outlier_frac=0.2, nui_frac=0.5)
    assert any(ft.anchor >= w.N for ft in slam) and any(ft.anchor < w.N for ft in slam)
    ref = mh.hybrid_update(w, slam, idp)
    assert 0 < ref['ekf_accept'].sum()
    upd.set_extra_states(w.n_extra)
    upd.set_schmidt_states(w.n_nui)
    upd.set_ekf_rows_mode(True)
    try:
        upd.upload(w)
        upd.upload_nuisance_poses(w.nui)
        if on_device:
            upd.upload_slam_features(idp, slam)
        else:
            He, Ha, Hx, Hf, r = compact_rows(w, slam, idp)
            upd.upload_ekf_rows(idp, [f.anchor for f in slam], [f.state for f in slam], list(range(len(slam))), He, Ha, Hx, Hf, r,
                                z_vel=np.array([f.z_vel for f in slam]))
        upd.run_update()
        upd.sync()
        got = upd.download()
        g, a = upd.download_ekf()
    finally:
        upd.set_ekf_rows_mode(False)
        upd.set_schmidt_states(0)
        upd.set_extra_states(0)
    assert np.array_equal(a, ref['ekf_accept'])
    assert rel(g, ref['ekf_gamma']) < 1e-9
    assert np.array_equal(got['accept'], ref['accept'])
    assert rel(got['dx'], ref['dx']) < TOL
    assert rel(got['P_new'], ref['P_new']) < TOL
    m = 6 * w.n_nui
    assert np.array_equal(got['P_new'][-m:, -m:], w.P[-m:, -m:])      # the prior's block, bit for bit


@pytest.mark.parametrize('idp', [3, 1])
def test_schmidt_frame_with_entering_features(upd, idp):
    """The whole Schmidt frame: MSCKF tracks, SLAM features (some anchored at nuisance states), features entering the state --
    their rows on the device, their states inserted IN FRONT of the nuisance block (:1920-1935, orcvio_msckf_augment_state_nuisance)."""
    w0 = synth.make_window(N=10, F=60, seed=17, track_len=(3, 10), flags=synth.Flags(use_larvio=1, estimate_td=1))
    slam = synth.make_slam_features(w0, 7, seed=5, outlier_frac=0.25)
    w = synth.with_nuisance_states(synth.with_extra_states(w0, idp * len(slam), seed=4), 2, seed=8)
    slam = synth.make_slam_features(w, 7, seed=5, outlier_frac=0.25, nui_frac=0.4)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 4, seed=9)]
    ref = mh.hybrid_update_full(w, slam, new, idp)
    acc = ref['new_accept']
    assert len(acc) > 0
    upd.set_extra_states(w.n_extra)
    upd.set_schmidt_states(w.n_nui)
    upd.set_ekf_rows_mode(True)
    try:
        upd.upload(w)
        upd.upload_nuisance_poses(w.nui)
        upd.upload_slam_features(idp, slam)
        upd.upload_new_features(w, idp, [new[i] for i in acc])
        upd.run_update()
        upd.sync()
        got = upd.download()
        H_1, H_2, r_1 = upd.download_new_feature_blocks()
    finally:
        upd.set_ekf_rows_mode(False)
        upd.set_schmidt_states(0)
        upd.set_extra_states(0)
    assert rel(got['dx'], ref['dx_leg']) < TOL
    assert rel(got['P_new'], ref['P_upd']) < TOL
    dx_new, P = capi.augment_state_nuisance(idp, 6 * w.n_nui, H_1, H_2, r_1, w.flags.noise_feature ** 2, got['dx'], got['P_new'])
    assert rel(np.concatenate([got['dx'], dx_new]), ref['dx']) < TOL
    assert rel(P, ref['P_new']) < TOL


@pytest.mark.parametrize('nui', [0, 2], ids=['plain', 'schmidt'])
@pytest.mark.parametrize('idp', [3, 1])
def test_entering_features_with_the_covariance_resident(upd, idp, nui):
    """orcvio_msckf_cov_commit_new_features: the covariance never leaves the device in a frame in which features enter the state --
    prior = resident covariance, rows of the entering features and the joint update on the device, the H_1 / H_2 tail
    (src/orcvio.cpp:1818-1821, :1904-1947; :1920-1935 with nuisance states) on the device too; tracks in, dx and dx_new out."""
    w0 = synth.make_window(N=10, F=60, seed=17, track_len=(3, 10), flags=synth.Flags(use_larvio=1, estimate_td=1))
    slam = synth.make_slam_features(w0, 7, seed=5, outlier_frac=0.25)
    w = synth.with_extra_states(w0, idp * len(slam), seed=4)
    if nui:
        w = synth.with_nuisance_states(w, nui, seed=8)
        slam = synth.make_slam_features(w, 7, seed=5, outlier_frac=0.25, nui_frac=0.4)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 4, seed=9)]
    ref = mh.hybrid_update_full(w, slam, new, idp)
    acc = ref['new_accept']
    assert len(acc) > 0
    upd.cov_set(w.P)
    upd.set_extra_states(w.n_extra)
    upd.set_schmidt_states(w.n_nui)
    upd.set_ekf_rows_mode(True)
    try:
        upd.upload(w, resident_cov=True)
        if nui:
            upd.upload_nuisance_poses(w.nui)
        upd.upload_slam_features(idp, slam)
        upd.upload_new_features(w, idp, [new[i] for i in acc])
        upd.run_update()
        upd.sync()
        got = upd.download()
        dx_new = upd.cov_commit_new_features()
    finally:
        upd.set_ekf_rows_mode(False)
        upd.set_schmidt_states(0)
        upd.set_extra_states(0)
    assert rel(np.concatenate([got['dx'], dx_new]), ref['dx']) < TOL
    assert rel(upd.cov_get(), ref['P_new']) < TOL


def test_entering_features_argument_checks(upd):
    """orcvio_msckf_upload_new_features: one call per upload, anchors inside the window (or the nuisance states), enough
    observations; ORCVIO_OPT_SCHMIDT_STATES must fit the extra states and the pose capacity."""
    w0 = synth.make_window(N=10, F=20, seed=17, track_len=(3, 10), flags=synth.Flags(use_larvio=1))
    w = synth.with_extra_states(w0, 3, seed=4)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 2, seed=9)]
    upd.set_extra_states(w.n_extra)
    upd.set_ekf_rows_mode(True)
    try:
        upd.upload(w)
        upd.upload_new_features(w, 3, new)
        with pytest.raises(capi.MsckfError):
            upd.upload_new_features(w, 3, new)              # a second call for the same upload
        upd.upload(w)
        import dataclasses
        bad = [dataclasses.replace(new[0], anchor=w.N)]     # no nuisance states: index N is outside
        with pytest.raises(capi.MsckfError):
            upd.upload_new_features(w, 3, bad)
        short = [dataclasses.replace(new[0], obs=new[0].obs[:1])]
        with pytest.raises(capi.MsckfError):
            upd.upload_new_features(w, 3, short)
        upd.set_schmidt_states(1)                            # 6 nuisance columns do not fit 3 extra states
        with pytest.raises(capi.MsckfError):
            upd.upload(w)
    finally:
        upd.set_schmidt_states(0)
        upd.set_ekf_rows_mode(False)
        upd.set_extra_states(0)


@pytest.mark.parametrize('nui', [0, 2], ids=['plain', 'schmidt'])
def test_reference_literal_h2_ldlt_tail_on_the_device(upd, nui):
    """ORCVIO_OPT_REF_H2_LDLT (VERDICT r2 'missing' 1): for feature_idp_dim = 3 the reference's `H_2.ldlt().solve(..)`
    (src/orcvio.cpp:1826-1827) reads the lower triangle of the upper-triangular H_2, i.e. its diagonal.  The option reproduces that
    (mirror_hybrid.augment_after_update(ref_ldlt=True)); the default solves the triangular system.  The two differ for d = 3."""
    idp = 3
    w0 = synth.make_window(N=10, F=60, seed=17, track_len=(3, 10), flags=synth.Flags(use_larvio=1, estimate_td=1))
    slam = synth.make_slam_features(w0, 7, seed=5, outlier_frac=0.25)
    w = synth.with_extra_states(w0, idp * len(slam), seed=4)
    if nui:
        w = synth.with_nuisance_states(w, nui, seed=8)
        slam = synth.make_slam_features(w, 7, seed=5, outlier_frac=0.25, nui_frac=0.4)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 4, seed=9)]
    lit = mh.hybrid_update_full(w, slam, new, idp, ref_ldlt=True)
    cor = mh.hybrid_update_full(w, slam, new, idp)
    acc = lit['new_accept']
    assert len(acc) > 0 and rel(lit['dx'], cor['dx']) > 1e-3   # the literal tail is a different update of the new states
    upd.cov_set(w.P)
    upd.set_extra_states(w.n_extra)
    upd.set_schmidt_states(w.n_nui)
    upd.set_ekf_rows_mode(True)
    upd.set_ref_h2_ldlt(True)
    try:
        upd.upload(w, resident_cov=True)
        if nui:
            upd.upload_nuisance_poses(w.nui)
        upd.upload_slam_features(idp, slam)
        upd.upload_new_features(w, idp, [new[i] for i in acc])
        upd.run_update()
        upd.sync()
        got = upd.download()
        H_1, H_2, r_1 = upd.download_new_feature_blocks()
        dx_new = upd.cov_commit_new_features()
    finally:
        upd.set_ref_h2_ldlt(False)
        upd.set_ekf_rows_mode(False)
        upd.set_schmidt_states(0)
        upd.set_extra_states(0)
    assert rel(np.concatenate([got['dx'], dx_new]), lit['dx']) < TOL
    assert rel(upd.cov_get(), lit['P_new']) < TOL
    # the handle-less twin on the same blocks
    dxn, P = capi.augment_state_nuisance(idp, 6 * w.n_nui, H_1, H_2, r_1, w.flags.noise_feature ** 2, got['dx'], got['P_new'], ref_ldlt=True)
    assert rel(np.concatenate([got['dx'], dxn]), lit['dx']) < TOL and rel(P, lit['P_new']) < TOL


def test_reference_literal_h2_ldlt_for_new_features_listed_as_tracks(upd):
    """orcvio_msckf_augment_new_features under ORCVIO_OPT_REF_H2_LDLT: its H_2 = R(xyz) J is not triangular (U = Q_1 of H_f in
    Cartesian coordinates), so the literal mode first moves to the reference's basis (the R factor of H_f in the inverse-depth
    parametrisation, a 3 x 3 QR) and then divides by the diagonal as `H_2.ldlt()` does (src/orcvio.cpp:1826-1827)."""
    import dataclasses
    w0 = synth.make_window(N=10, F=60, seed=17, track_len=(3, 10), flags=synth.Flags(use_larvio=1, estimate_td=1))
    slam = synth.make_slam_features(w0, 7, seed=5, outlier_frac=0.25)
    w = synth.with_extra_states(w0, 3 * len(slam), seed=4)
    new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, 5, seed=9, outlier_frac=0.5)]
    lit = mh.hybrid_update_full(w, slam, new, 3, ref_ldlt=True)
    obs_ptr = list(w.obs_ptr)
    obs_clone, obs_z, obs_zvel, p_w = list(w.obs_clone), list(w.obs_z), list(w.obs_zvel), list(w.p_w)
    for ft in new:
        p_w.append(ft.p_w)
        for (k, z, zv) in ft.obs:
            obs_clone.append(k); obs_z.append(z); obs_zvel.append(zv)
        obs_ptr.append(len(obs_clone))
    w2 = dataclasses.replace(w, p_w=np.ascontiguousarray(p_w), obs_ptr=np.asarray(obs_ptr, dtype=np.int32),
                             obs_clone=np.asarray(obs_clone, dtype=np.int32), obs_z=np.ascontiguousarray(obs_z).reshape(-1, 2),
                             obs_zvel=np.ascontiguousarray(obs_zvel).reshape(-1, 2))
    upd.set_extra_states(w2.n_extra)
    upd.set_ekf_rows_mode(True)
    upd.set_ref_h2_ldlt(True)
    try:
        upd.upload(w2)
        upd.upload_slam_features(3, slam)
        upd.run_update()
        upd.sync()
        got = upd.download()
        acc_new = [i for i in range(len(new)) if got['accept'][w.F + i]]
        dx_new, P_aug = upd.augment_new_features(w2, [w.F + i for i in acc_new], [new[i].anchor for i in acc_new],
                                                 [new[i].inv_param for i in acc_new], got['dx'], got['P_new'])
    finally:
        upd.set_ref_h2_ldlt(False)
        upd.set_ekf_rows_mode(False)
        upd.set_extra_states(0)
    assert acc_new == lit['new_accept']
    assert rel(np.concatenate([got['dx'], dx_new]), lit['dx']) < TOL
    assert rel(P_aug, lit['P_new']) < TOL


@pytest.mark.parametrize('jac', ['larvio', 'orcvio_right', 'orcvio_left', 'kitti_raw'])
@pytest.mark.parametrize('idp', [1, 3])
def test_hybrid_update_through_the_in_place_call(upd, idp, jac):
    """The rows of the in-state features between orcvio_msckf_io_begin and orcvio_msckf_io_update: the hybrid update with the
    window written in place, the commit inside the launch and the results through the flag word -- equal to the staged form."""
    # (kitti_raw: config/kitti_raw.yaml's shipped set -- OrcVIO right perturbation, noise_feature 1, discard flag on, :103, :143-158)
    fl = synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=1.0, discard_large_update=1) if jac == 'kitti_raw' else synth.Flags(**JAC[jac])
    w0 = synth.make_window(N=20, F=120, seed=41, track_len=(3, 6), flags=fl, outlier_frac=0.05, sigma_px=0.008 if jac == 'kitti_raw' else None)
    slam = synth.make_slam_features(w0, 12, seed=idp, outlier_frac=0.1, sigma_px=0.008 if jac == 'kitti_raw' else None)
    w = synth.with_extra_states(w0, idp * len(slam), seed=7)
    ref = mh.hybrid_update(w, slam, idp)
    staged = run(upd, w, slam, idp, True)
    upd.set_extra_states(w.n_extra)
    upd.set_ekf_rows_mode(True)
    try:
        upd.cov_set(w.P)
        upd.cov_prefactor()
        for _ in range(3):   # (three times: direct launches, the capture, the replay)
            upd.cov_set(w.P)
            upd.cov_prefactor()
            io = upd.io_begin(w.flags, w.N, w.F, int(w.obs_ptr[-1]), with_P=False)
            upd.io_fill(io, w, with_P=False)
            upd.upload_slam_features(idp, slam)
            upd.io_update(want_P=False, commit=True)
            assert np.array_equal(io['accept'], ref['accept'])
            assert rel(io['dx'], ref['dx']) < TOL and rel(io['dx'], staged['dx']) < 1e-12
            g, a = upd.download_ekf()
            assert np.array_equal(a, ref['ekf_accept']) and rel(g, ref['ekf_gamma']) < 1e-9
            assert rel(upd.cov_get(), ref['P_new']) < TOL
    finally:
        upd.set_ekf_rows_mode(False)
        upd.set_extra_states(0)

"""CPU tests of the oracle itself: golden vectors, Jacobians against central differences,
C restatement against the numpy mirror, chi-square quantiles."""
import numpy as np
import pytest

from orcvio_amd import synth
from oracle import mirror, oracle
from helpers import rel, golden_files, window_from_golden, subset_window, GOLDEN


@pytest.mark.parametrize('path', golden_files(), ids=lambda p: p.split('feat_')[-1][:-4])
def test_c_oracle_matches_golden(built, path):
    w, g = window_from_golden(path)
    o = oracle.msckf_update(w)
    assert np.array_equal(o['accept'], g['exp_accept'])
    assert rel(o['gamma'], g['exp_gamma']) < 1e-9
    assert rel(o['dx'], g['exp_dx']) < 1e-9
    assert rel(o['P_new'], g['exp_P']) < 1e-9
    assert rel(o['G'], g['exp_G']) < 1e-9
    # per-observation Jacobians
    k = 0
    for j in range(w.F):
        for o_ in range(w.obs_ptr[j], w.obs_ptr[j + 1]):
            Hx, He, Hf, r = oracle.measurement_jacobian(w, int(w.obs_clone[o_]), w.p_w[j], w.obs_z[o_])
            assert np.allclose(Hx, g['exp_Hx'][k], rtol=1e-9, atol=1e-12)
            assert np.allclose(He, g['exp_He'][k], rtol=1e-9, atol=1e-12)
            assert np.allclose(Hf, g['exp_Hf'][k], rtol=1e-9, atol=1e-12)
            assert np.allclose(r, g['exp_r'][k], rtol=1e-9, atol=1e-12)
            k += 1
    # basis-invariant block data (nullspace basis differs: SVD in the mirror, Householder here)
    bp = o['block_ptr']
    for j in range(w.F):
        H = o['H_all'][bp[j]:bp[j + 1]]
        r = o['r_all'][bp[j]:bp[j + 1]]
        assert rel(H.T @ H, g['exp_block_gram'][j]) < 1e-9
        assert rel(H.T @ r, g['exp_block_Htr'][j]) < 1e-8
        assert abs(r @ r - g['exp_block_rr'][j]) <= 1e-9 * max(1.0, g['exp_block_rr'][j])


def _project(win, i, p_w, dtheta, dp, dth_e, dt_e):
    """pi(p_c) with the error-state increments of SURVEY.md 8a row 11 applied to clone i."""
    f = win.flags
    R = win.R_b2w[i]
    Rt = mirror.so3_exp(dtheta)
    left = bool(f.use_larvio or f.use_left_perturbation)
    R_b2w = Rt @ R if left else R @ Rt
    t_b_w = win.t_b_w[i] + dp
    q = mirror.small_angle_quaternion(dth_e)
    R_b2c = win.R_b2c[i] @ mirror.quat_to_rot_hamilton(q).T
    t_c_b = win.t_c_b[i] + dt_e
    R_w2c = R_b2c @ R_b2w.T
    t_c_w = t_b_w + R_b2w @ t_c_b
    pc = R_w2c @ (p_w - t_c_w)
    return pc[:2] / pc[2]


@pytest.mark.parametrize('larvio,left', [(1, 0), (0, 0), (0, 1)])
def test_jacobians_against_central_differences(built, larvio, left):
    """r = z - pi(.), so dr/dx = -d pi/dx; the reference stacks H with r = H dx + n (residual
    defined as z - zhat with H = d zhat/dx).  All three H_x variants and H_e, H_f."""
    w = synth.make_window(N=4, F=5, seed=21, track_len=4, flags=synth.Flags(use_larvio=larvio, use_left_perturbation=left))
    eps = 1e-6
    for j in range(w.F):
        for k in range(w.obs_ptr[j], w.obs_ptr[j + 1]):
            i = int(w.obs_clone[k])
            Hx, He, Hf, _ = oracle.measurement_jacobian(w, i, w.p_w[j], w.obs_z[k])
            num = np.zeros((2, 15))
            for c in range(15):
                d = np.zeros(15)
                d[c] = eps
                zp = _project(w, i, w.p_w[j] + d[12:15], d[0:3], d[3:6], d[6:9], d[9:12])
                zm = _project(w, i, w.p_w[j] - d[12:15], -d[0:3], -d[3:6], -d[6:9], -d[9:12])
                num[:, c] = (zp - zm) / (2 * eps)
            assert np.allclose(Hx, num[:, 0:6], atol=2e-8), (larvio, left)
            assert np.allclose(He, num[:, 6:12], atol=2e-8)
            assert np.allclose(Hf, num[:, 12:15], atol=2e-8)


def test_c_oracle_matches_mirror_ragged(built):
    w = synth.make_window(N=10, F=30, seed=5, track_len=(2, 10), outlier_frac=0.3)
    m = mirror.msckf_update(w)
    o = oracle.msckf_update(w)
    assert np.array_equal(o['accept'], m['accept'])
    assert 0 < o['accept'].sum() < w.F
    assert rel(o['gamma'], m['gamma']) < 1e-10
    assert rel(o['dx'], m['dx']) < 1e-9
    assert rel(o['P_new'], m['P_new']) < 1e-10
    assert rel(o['G'], m['G']) < 1e-9


def test_prune_variant_matches_mirror(built):
    """pruneImuStateBuffer (src/orcvio.cpp:2803-2851): only the two removed clones take part."""
    w = synth.make_window(N=8, F=25, seed=9, track_len=(4, 8))
    rm = {1, 2}
    m = mirror.msckf_update(w, clone_subset=rm)
    mask = np.zeros(w.N, dtype=np.int32)
    mask[list(rm)] = 1
    o = oracle.msckf_update(w, clone_mask=mask)
    o2 = oracle.msckf_update(subset_window(w, list(rm)))   # host-side filtered CSR, same thing
    for res in (o, o2):
        assert np.array_equal(res['accept'], m['accept'])
        assert rel(res['dx'], m['dx']) < 1e-9
        assert rel(res['P_new'], m['P_new']) < 1e-10
    assert o['stacked_rows'] == int(m['accept'].sum())   # dof 2*2-3 = 1 row per feature


def test_nullspace_projection_properties(built):
    """A has orthonormal columns spanning null(H_f^T): block Gram equals H^T (I - Q1 Q1^T) H."""
    w = synth.make_window(N=6, F=6, seed=3, track_len=6)
    o = oracle.msckf_update(w)
    bp = o['block_ptr']
    for j in range(w.F):
        Hx, r, Hf = mirror.feature_jacobian_msckf(w, j, project=False)
        Q1, _ = np.linalg.qr(Hf)
        Pn = np.eye(Hf.shape[0]) - Q1 @ Q1.T
        H = o['H_all'][bp[j]:bp[j + 1]]
        assert H.shape[0] == Hf.shape[0] - 3
        assert rel(H.T @ H, Hx.T @ Pn @ Hx) < 1e-10


def test_nullspace_trick_shapes_as_the_reference_test():
    """reference src/tests/test_state_update.cpp:106-... (testObjectResidualNullSpaceTrick): 12 residuals, 5 object-state columns,
    10 sensor-state columns -> the projection succeeds and leaves 12 - 5 = 7 rows; the reference compares its SVD form
    (nullspace_project_inplace_svd, math_utils.hpp:287-312) with the QR twin (:315-344): both are bases of the same left null space,
    so their Gram data agree.  Its second data case (12 x 21 against 12 x 130, rows <= columns) must return false and leave the inputs
    as they are (math_utils.hpp:292)."""
    rng = np.random.default_rng(12)
    Hf, Hx, r = rng.uniform(-1, 1, (12, 5)), rng.uniform(-1, 1, (12, 10)), rng.uniform(-1, 1, 12)   # (Eigen's Random(): uniform in [-1, 1])
    ok, H1, r1 = mirror.nullspace_project_svd(Hf, Hx, r)
    assert ok and H1.shape == (7, 10) and r1.shape == (7,)
    Q, _ = np.linalg.qr(Hf, mode='complete')   # the QR twin: Q2 = the last rows - cols columns of the Householder Q
    H2, r2 = Q[:, 5:].T @ Hx, Q[:, 5:].T @ r
    assert H2.shape == H1.shape
    assert rel(H1.T @ H1, H2.T @ H2) < 1e-12 and rel(H1.T @ r1, H2.T @ r2) < 1e-12 and abs(r1 @ r1 - r2 @ r2) < 1e-12
    Hf2, Hx2, r2 = rng.uniform(-1, 1, (12, 21)), rng.uniform(-1, 1, (12, 130)), rng.uniform(-1, 1, 12)
    ok2, H3, r3 = mirror.nullspace_project_svd(Hf2, Hx2, r2)
    assert not ok2 and H3 is Hx2 and r3 is r2


def test_chi2_quantile_table(built):
    g = np.load(GOLDEN + '/chi2_095.npz')
    t = oracle.chi2_table(0.95, 500)
    assert np.max(np.abs(t[1:] - g['table'][1:]) / g['table'][1:]) < 1e-12
    for dof, val in g['big']:
        assert abs(oracle.chi2_quantile(int(dof)) - val) / val < 1e-12


def test_increment_state_discard(built):
    f = synth.Flags(discard_large_update=1)
    st = dict(R_b2w_imu=np.eye(3), v=np.zeros(3), p=np.zeros(3), bg=np.zeros(3), ba=np.zeros(3), R_b2c=np.eye(3),
              t_c_b=np.zeros(3), td=np.zeros(()), R_b2w=np.tile(np.eye(3), (2, 1, 1)), t_b_w=np.zeros((2, 3)))
    dx = np.zeros(22 + 12)
    dx[6] = 2.0
    _, applied = mirror.increment_state(st, dx, f)
    assert not applied
    dx[6] = 0.1
    s2, applied = mirror.increment_state(st, dx, f)
    assert applied and np.isclose(s2['p'][0], 0.1)


@pytest.mark.parametrize('flags', [dict(use_larvio=1), dict(use_larvio=0, use_left_perturbation=0, estimate_td=1, if_fej=1),
                                   dict(use_larvio=0, use_left_perturbation=1)])
def test_three_independent_restatements_agree(flags):
    """The literal numpy restatement, the literal C port and the minimum-work OpenMP variant (oracle/msckf_fast.c: three
    reflectors instead of a full-U SVD, gate on the touched columns, Gram compression, square-root solve) are written
    separately and must give the same update; so must the 60-digit evaluation of the formula on a small window."""
    from oracle import oracle as orc, mp_reference
    w = synth.make_window(N=7, F=30, seed=5, track_len=(2, 7), outlier_frac=0.2, flags=synth.Flags(**flags))
    a = mirror.msckf_update(w)
    b = orc.msckf_update(w, want_blocks=False, want_K=False)
    c = orc.msckf_update_fast(w)
    assert 0 < a['accept'].sum() < w.F
    for other in (b, c):
        assert np.array_equal(other['accept'], a['accept'])
        fin = np.isfinite(a['gamma'])
        assert rel(other['gamma'][fin], a['gamma'][fin]) < 1e-10
        assert rel(other['dx'], a['dx']) < 1e-9 and rel(other['P_new'], a['P_new']) < 1e-12
    small = synth.make_window(N=4, F=8, seed=6, track_len=(3, 4), flags=synth.Flags(**flags))
    hp = mp_reference.msckf_update_mp(small)
    for upd in (mirror.msckf_update(small), orc.msckf_update(small, want_blocks=False, want_K=False), orc.msckf_update_fast(small)):
        assert np.array_equal(upd['accept'], hp['accept'])
        assert rel(upd['dx'], hp['dx']) < 1e-10 and rel(upd['P_new'], hp['P_new']) < 1e-12


def test_fast_port_with_live_extrinsics_and_team_sizes():
    """oracle/msckf_fast.c (bench.py's all-cores CPU figure) on a window whose extrinsic rows of P are live, one thread and a
    team: same gate decisions and update as the literal port."""
    import ctypes
    from oracle import oracle as orc
    w = synth.make_window(N=9, F=40, seed=11, track_len=(3, 9), outlier_frac=0.15, estimate_extrin=True, flags=synth.Flags(use_larvio=1))
    ref = orc.msckf_update(w, want_blocks=False, want_K=False)
    try:
        for team in (1, 3):
            orc.lib().orc_fast_set_threads(ctypes.c_int(team))
            got = orc.msckf_update_fast(w)
            assert got['threads'] == team
            assert np.array_equal(got['accept'], ref['accept']) and 0 < ref['accept'].sum() < w.F
            assert rel(got['dx'], ref['dx']) < 1e-9 and rel(got['P_new'], ref['P_new']) < 1e-12
    finally:
        orc.lib().orc_fast_set_threads(ctypes.c_int(0))

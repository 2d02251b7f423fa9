"""CPU tests pinning the object-row restatement to the reference's own golden fixtures
(src/tests/data/*.h5 converted by scripts/convert_ref_h5.py) and to central differences."""
import numpy as np
import pytest

from oracle import mirror_objects as mo
from helpers import GOLDEN


def test_keypoint_rows_match_reference_golden():
    """reference src/tests/test_object_lm.cpp:90-152 (error 24, jacobian 24x45, tolerance 1e-6)."""
    g = np.load(GOLDEN + '/ref_test_error_feature_quadric.npz')
    fr = [dict(wTc=np.linalg.inv(g['S']), zs=g['zs'], bbox=np.array([-0.1, -0.1, 0.1, 0.1]))]
    res, Hf, Jc, counts = mo.object_rows(g['T'], np.ones(3), g['M'][:, :3], fr, left=True, new_bbox=False)
    assert counts == [12]
    assert np.abs(res[:24] - g['error'].ravel()).max() < 1e-12
    assert np.abs(Hf[:24] - g['jacobian']).max() < 1e-12


def test_project_object_points_as_the_reference_tests():
    """reference src/tests/test_se3.cpp:60-74 (the known answer f X / Z + x) and :76-90 (test_project_object_points_data: the projection
    of the fixture's keypoints equals its stored error + its observations, 1e-6)."""
    f, x, y, X, Y, Z = 0.25, 2.0, 2.0, 4.0, 4.0, 0.5
    P = np.array([[f, 0, x, 0], [0, f, y, 0], [0, 0, 1.0, 0]])
    uv = mo.project_object_points(P, np.eye(4), np.array([[X, Y, Z, 1.0]]))
    assert np.array_equal(uv, np.array([[f * X / Z + x, f * Y / Z + y]]))
    g = np.load(GOLDEN + '/ref_test_error_feature_quadric.npz')
    uvs = mo.project_object_points(g['S'][:3], g['T'], g['M'])
    assert np.linalg.norm(uvs - (g['error'].reshape(12, 2) + g['zs'])) < 1e-6


def test_old_bbox_rows_match_reference_golden():
    """reference src/tests/test_object_lm.cpp:154-202 (error 4, jacobian 4x45)."""
    g = np.load(GOLDEN + '/ref_test_error_bbox_quadric.npz')
    fr = [dict(wTc=np.linalg.inv(g['S']), zs=np.full((12, 2), np.nan), bbox=g['zb'].ravel())]
    res, Hf, Jc, counts = mo.object_rows(g['T'], g['v'], np.zeros((12, 3)), fr, left=True, new_bbox=False)
    assert counts == [0] and len(res) == 4
    assert np.abs(res - g['error'].ravel()).max() < 1e-12
    assert np.abs(Hf - g['jacobian']).max() < 1e-12


@pytest.mark.parametrize('left', [True, False])
def test_camera_jacobians_against_central_differences(left):
    """reference src/tests/test_object_lm.cpp:482-628: CameraLM Jacobians vs numeric (old bbox form)."""
    g = np.load(GOLDEN + '/ref_test_error_feature_quadric.npz')
    wTc = np.linalg.inv(g['S'])
    v = np.array([0.8, 1.9, 0.5])
    bbox = np.array([-0.3, -0.2, 0.25, 0.3])
    kps = g['M'][:, :3]
    base = mo.object_rows(g['T'], v, kps, [dict(wTc=wTc, zs=g['zs'], bbox=bbox)], left, False)
    eps = 1e-6
    J = np.zeros((len(base[0]), 6))
    for c in range(6):
        d = np.zeros(6)
        d[c] = eps
        Tp = mo.se3_exp(d) @ wTc if left else wTc @ mo.se3_exp(d)
        Tm = mo.se3_exp(-d) @ wTc if left else wTc @ mo.se3_exp(-d)
        rp = mo.object_rows(g['T'], v, kps, [dict(wTc=Tp, zs=g['zs'], bbox=bbox)], left, False)[0]
        rm = mo.object_rows(g['T'], v, kps, [dict(wTc=Tm, zs=g['zs'], bbox=bbox)], left, False)[0]
        J[:, c] = (rp - rm) / (2 * eps)
    assert np.abs(J - base[2]).max() < 1e-8


def test_construct_rows_layout_matches_reference_test():
    """reference src/tests/test_state_update.cpp:16-103: 2 frames x (1 keypoint + bbox), LEG 15, D = I."""
    rng = np.random.default_rng(0)
    F = 2
    r = rng.standard_normal(F * 2 + F * 4)
    Hf = rng.standard_normal((F * 2 + F * 4, 45))
    J = rng.standard_normal((F * 2 + F * 4, 6))
    out = mo.construct_object_residual_jacobians(J, [0, 1], Hf, r, [1, 1], [np.eye(4)] * 2, np.eye(3), np.zeros(3),
                                                 False, 15, 2, fix_D_identity=True)
    Hx, Hf2, r2, rc, hx6 = out
    Hx_true = np.zeros((12, 15 + 12)); Hf_true = np.zeros((12, 45)); r_true = np.zeros(12)
    for i in range(12):
        if i < F * 2:
            nr = (i // 2) * 6 + (i % 2); nc = (i // 2) * 6 + 15
        else:
            j = i - F * 2
            nr = (j // 4) * 6 + (j % 4) + 2; nc = (j // 4) * 6 + 15
        r_true[nr] = r[i]; Hf_true[nr] = Hf[i]; Hx_true[nr, nc:nc + 6] = J[i]
    assert np.allclose(Hx, Hx_true) and np.allclose(Hf2, Hf_true) and np.allclose(r2, r_true)
    # frames outside the window are dropped; none in the window -> None
    assert mo.construct_object_residual_jacobians(J, [-1, -1], Hf, r, [1, 1], [np.eye(4)] * 2, np.eye(3), np.zeros(3),
                                                  False, 15, 2, fix_D_identity=True) is None
    out = mo.construct_object_residual_jacobians(J, [-1, 1], Hf, r, [1, 1], [np.eye(4)] * 2, np.eye(3), np.zeros(3),
                                                 False, 15, 2, fix_D_identity=True)
    assert out[0].shape == (6, 27) and np.all(out[3] == 1)


def test_se3_exp_log_roundtrip():
    rng = np.random.default_rng(1)
    for _ in range(10):
        xi = rng.standard_normal(6) * 0.7
        assert np.allclose(mo.se3_log(mo.se3_exp(xi)), xi, atol=1e-12)


@pytest.mark.parametrize('obj_left,new_bbox,vio_left', [(True, False, 0), (False, False, 0), (True, True, 0), (False, True, 1)])
def test_c_object_oracle_matches_the_mirror_and_the_goldens(built, obj_left, new_bbox, vio_left):
    """oracle/object_oracle.c (the C twin, bench.py's CPU baseline of the object update) against mirror_objects.py: rows of a
    ragged track, and the update of several objects; and against the reference's golden vectors directly."""
    from orcvio_amd import synth
    from oracle import oracle as orc
    from helpers import object_rows_reference, objects_update_reference, rel
    flags = synth.Flags(use_larvio=0, use_left_perturbation=vio_left)
    win = synth.make_window(N=10, F=4, seed=21, flags=flags, track_len=4)
    objs = synth.make_objects(win, n_objects=3, seed=9, sigma_kp=0.004, missing_frac=0.25)
    objs[0].frames[3]['clone'] = -1
    blocks = []
    for ob in objs:
        Hx, Hf, r, rc, hx6 = object_rows_reference(win, ob, obj_left, new_bbox, vio_left)
        got = orc.object_rows_c(ob, win.R_b2c[0], win.t_c_b[0], obj_left, new_bbox, vio_left)
        assert np.array_equal(got['row_clone'], rc)
        assert rel(got['res'], r) < 1e-10 and rel(got['Hx6'], hx6) < 1e-10 and rel(got['Hf'], Hf) < 1e-10
        blocks.append(got)
    ref = objects_update_reference(win, objs, win.P, obj_left, new_bbox, vio_left)
    upd = orc.objects_update_c(flags, win.N, blocks, win.P)
    assert upd['accept'] == ref['accept'] and upd['dof'] == ref['dof']
    assert abs(upd['gamma'] - ref['gamma']) < 1e-8 * abs(ref['gamma'])
    assert rel(upd['dx'], ref['dx']) < 1e-8 or not ref['accept']
    assert rel(upd['P_new'], ref['P_new']) < 1e-10
    if obj_left and not new_bbox and not vio_left:   # the reference's own vectors (test_object_lm.cpp:90-202)
        g = np.load(GOLDEN + '/ref_test_error_feature_quadric.npz')
        o = synth.ObjectTrack(wTo=g['T'], shape=np.ones(3), kps=g['M'][:, :3].copy(),
                              frames=[dict(clone=0, wTc=np.linalg.inv(g['S']), zs=g['zs'], bbox=np.array([-0.1, -0.1, 0.1, 0.1]))])
        c = orc.object_rows_c(o, np.eye(3), np.zeros(3), True, False, 0, fix_D=True)
        assert np.abs(c['res'][:24] - g['error'].ravel()).max() < 1e-12 and np.abs(c['Hf'][:24] - g['jacobian']).max() < 1e-12
        gb = np.load(GOLDEN + '/ref_test_error_bbox_quadric.npz')
        ob = synth.ObjectTrack(wTo=gb['T'], shape=gb['v'].copy(), kps=np.zeros((12, 3)),
                               frames=[dict(clone=0, wTc=np.linalg.inv(gb['S']), zs=np.full((12, 2), np.nan), bbox=gb['zb'].ravel().copy())])
        cb = orc.object_rows_c(ob, np.eye(3), np.zeros(3), True, False, 0, fix_D=True)
        assert np.abs(cb['res'] - gb['error'].ravel()).max() < 1e-12 and np.abs(cb['Hf'] - gb['jacobian']).max() < 1e-12


@pytest.mark.parametrize('left', [True, False])
def test_corrected_new_bbox_jacobians_match_central_differences(left):
    """SURVEY note N8: the reference's Jacobians of the NEW bbox residual take the plane in the world frame and drop -sign(b4);
    mode 1 restates that literally (and is off by O(1) from the numerical derivative), mode 2 is the opt-in corrected form: camera
    pose, object pose and shape Jacobians within 1e-9 of central differences on the reference's bbox fixture inputs."""
    g = np.load(GOLDEN + '/ref_test_error_bbox_quadric.npz')
    wTc, wTo, v, bbox = np.linalg.inv(g['S']), g['T'], g['v'], g['zb'].ravel()
    kps, zs = np.zeros((12, 3)), np.full((12, 2), np.nan)

    def rows(wTc_, wTo_, v_, mode):
        return mo.object_rows(wTo_, v_, kps, [dict(wTc=wTc_, zs=zs, bbox=bbox)], left, mode)
    eps = 1e-6
    for mode, bound in ((2, 1e-9), (1, None)):
        res0, Hf0, Jc0, _ = rows(wTc, wTo, v, mode)
        Jn, Jo, Js = np.zeros((4, 6)), np.zeros((4, 6)), np.zeros((4, 3))
        for c in range(6):
            d = np.zeros(6); d[c] = eps
            pert = (lambda T, s: mo.se3_exp(s * d) @ T) if left else (lambda T, s: T @ mo.se3_exp(s * d))
            Jn[:, c] = (rows(pert(wTc, 1), wTo, v, mode)[0] - rows(pert(wTc, -1), wTo, v, mode)[0]) / (2 * eps)
            Jo[:, c] = (rows(wTc, pert(wTo, 1), v, mode)[0] - rows(wTc, pert(wTo, -1), v, mode)[0]) / (2 * eps)
        for c in range(3):
            d = np.zeros(3); d[c] = eps
            Js[:, c] = (rows(wTc, wTo, v + d, mode)[0] - rows(wTc, wTo, v - d, mode)[0]) / (2 * eps)
        errs = (np.abs(Jn - Jc0).max(), np.abs(Jo - Hf0[:, :6]).max(), np.abs(Js - Hf0[:, 6:9]).max())
        if bound is not None:
            assert max(errs) < bound, errs
        else:
            assert min(errs) > 0.1, errs   # the literal form really is inconsistent: that is what parity means for mode 1
    # the residual itself is the same in both modes
    assert np.array_equal(rows(wTc, wTo, v, 1)[0], rows(wTc, wTo, v, 2)[0])


@pytest.mark.parametrize('seed', [0, 2, 13, 1060])
def test_whole_null_space_reference_equals_the_svd_reference_on_full_rank_blocks(seed):
    """helpers.objects_update_reference(full_nullspace=True) -- what the device's object update is held to when some H_f is rank
    deficient -- is the reference's full-U-SVD projection whenever every H_f has full column rank; on a
    rank-deficient block (seeds 13, 1060: a keypoint never seen / a dependent border column) it stacks rows - rank directions per
    object instead of rows - columns, keeps the reference's degrees of freedom for the gate, and the two differ."""
    from helpers import objects_update_reference, random_object_case, rel
    c = random_object_case(seed)
    win, objs = c['win'], c['objs']
    a = objects_update_reference(win, objs, win.P, c['obj_left'], c['new_bbox'], c['vio_left'])
    b = objects_update_reference(win, objs, win.P, c['obj_left'], c['new_bbox'], c['vio_left'], full_nullspace=True)
    assert a['dof'] == b['dof'] and len(a['blocks']) == len(b['blocks'])
    if b['rank_deficient'] == 0:
        assert a['accept'] == b['accept']
        assert abs(a['gamma'] - b['gamma']) < 1e-7 * abs(a['gamma'])
        assert rel(b['dx'], a['dx']) < 1e-7 or (not a['dx'].any() and not b['dx'].any())
        assert rel(b['P_new'], a['P_new']) < 1e-9
    else:
        assert seed in (13, 1060)
        assert b['gamma'] >= a['gamma'] * (1 - 1e-9)   # more directions of the same residual: the distance cannot shrink


@pytest.mark.parametrize('case', [dict(N=10, nobj=3, missing=0.0, noise=0.004), dict(N=30, nobj=6, missing=0.1, noise=0.004),
                                  dict(N=12, nobj=2, missing=0.0, noise=0.2)])
def test_minimum_work_object_port_agrees_with_the_literal_one(built, case):
    """oracle/object_fast.c (bench.py's all-cores CPU figure for the object update: Schur-complement projection from the 7 non-zeros
    per row, square-root solve) against oracle/object_oracle.c (literal: reflectors applied to dense rows, QR of the stack) and
    the numpy reference: same gate decision, same update.  Full-rank H_f (every keypoint seen): the two projections coincide."""
    import ctypes
    from orcvio_amd import synth
    from oracle import oracle as orc
    from helpers import objects_update_reference, rel
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=case['N'], F=4, seed=23, flags=flags, track_len=4)
    objs = synth.make_objects(win, n_objects=case['nobj'], seed=5, sigma_kp=case['noise'], missing_frac=case['missing'])
    blocks = [orc.object_rows_c(ob, win.R_b2c[0], win.t_c_b[0], True, False, 0) for ob in objs]
    ref = objects_update_reference(win, objs, win.P, True, False, 0)
    lit = orc.objects_update_c(flags, win.N, blocks, win.P)
    for team in (1, 3):
        got = orc.objects_update_fast(flags, win.N, blocks, win.P, threads=team)
        assert got['threads'] == min(team, len(blocks))
        assert got['accept'] == lit['accept'] == ref['accept'] and got['dof'] == lit['dof']
        assert abs(got['gamma'] - lit['gamma']) < 1e-7 * abs(lit['gamma'])
        if ref['accept']:
            assert rel(got['dx'], lit['dx']) < 1e-7 and rel(got['P_new'], lit['P_new']) < 1e-9
            assert rel(got['dx'], ref['dx']) < 1e-7
        else:
            assert not np.any(got['dx']) and np.array_equal(got['P_new'], win.P)
    assert ref['accept'] == (0 if case['noise'] > 0.1 else 1)

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

"""Device twins of the reference's own object tests, on the reference's own fixtures (tests/golden/ref_*.npz, converted from
src/tests/data/*.h5 by scripts/convert_ref_h5.py):

  test_object_lm.cpp:154-202            old bbox residual + 4 x 45 Jacobian       -> k_object_rows, bbox lanes
  test_object_lm_multiframe.cpp:129-570 two duplicated frames, block_start_frame  -> row offsets of a two-frame track
  test_object_lm_multiframe.cpp:61-125  data/one_car/frame_{0..46}.h5             -> rows of all 47 real frames on the device
                                                                                    and the object update built from them
The stored error / Jacobian vectors are the reference's outputs; where the fixture holds inputs only (one_car) the device is
compared with the restatement that those vectors pin (tests/test_oracle_objects.py)."""
import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import mirror_objects as mo
from helpers import GOLDEN, rel, object_rows_reference, objects_update_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=48, max_features=64, max_observations=1024)
    yield u
    u.close()


def test_old_bbox_rows_reference_golden_on_the_device(upd):
    """src/tests/test_object_lm.cpp:154-202: error (4) and Jacobian (4 x 45) of the old bbox residual, left perturbation."""
    g = np.load(GOLDEN + '/ref_test_error_bbox_quadric.npz')
    obj = synth.ObjectTrack(wTo=g['T'], shape=g['v'].copy(), kps=np.zeros((12, 3)),
                            frames=[dict(clone=0, wTc=np.linalg.inv(g['S']), zs=np.full((12, 2), np.nan), bbox=g['zb'].ravel().copy())])
    got = upd.object_rows_eval(obj, np.eye(3), np.zeros(3), True, False, 0, fix_D=True)
    assert got['res'].shape[0] == 4   # no valid keypoint: the four bbox rows only
    assert np.abs(got['res'] - g['error'].ravel()).max() < 1e-12
    assert np.abs(got['Hf'] - g['jacobian']).max() < 1e-12


def test_two_frame_row_layout_on_the_device(upd):
    """src/tests/test_object_lm_multiframe.cpp:129-570: the single frame of both fixtures duplicated; the functors stack
    [kp rows of all frames ; bbox rows of all frames] (block_start_frame), constructObjectResidualJacobians interleaves them
    per frame -- the device writes [kp f0 ; bbox f0 ; kp f1 ; bbox f1] directly."""
    gk = np.load(GOLDEN + '/ref_test_error_feature_quadric.npz')
    gb = np.load(GOLDEN + '/ref_test_error_bbox_quadric.npz')
    # keypoint rows from the keypoint fixture (two frames)
    fr = dict(wTc=np.linalg.inv(gk['S']), zs=gk['zs'], bbox=np.array([-0.1, -0.1, 0.1, 0.1]))
    obj = synth.ObjectTrack(wTo=gk['T'], shape=np.ones(3), kps=gk['M'][:, :3].copy(), frames=[dict(clone=0, **fr), dict(clone=1, **fr)])
    got = upd.object_rows_eval(obj, np.eye(3), np.zeros(3), True, False, 0, fix_D=True)
    assert got['res'].shape[0] == 2 * 28
    for f in range(2):
        assert np.all(got['row_clone'][28 * f: 28 * f + 28] == f)
        assert np.abs(got['res'][28 * f: 28 * f + 24] - gk['error'].ravel()).max() < 1e-12
        assert np.abs(got['Hf'][28 * f: 28 * f + 24] - gk['jacobian']).max() < 1e-12
    # bbox rows from the bbox fixture (two frames, no valid keypoint)
    frb = dict(wTc=np.linalg.inv(gb['S']), zs=np.full((12, 2), np.nan), bbox=gb['zb'].ravel().copy())
    objb = synth.ObjectTrack(wTo=gb['T'], shape=gb['v'].copy(), kps=np.zeros((12, 3)), frames=[dict(clone=0, **frb), dict(clone=1, **frb)])
    gotb = upd.object_rows_eval(objb, np.eye(3), np.zeros(3), True, False, 0, fix_D=True)
    assert gotb['res'].shape[0] == 8
    for f in range(2):
        assert np.abs(gotb['res'][4 * f: 4 * f + 4] - gb['error'].ravel()).max() < 1e-12
        assert np.abs(gotb['Hf'][4 * f: 4 * f + 4] - gb['jacobian']).max() < 1e-12
    # the mirror's functor layout ([kp all frames ; bbox all frames]) re-indexed equals the device's interleaved layout
    res, Hf, Jc, counts = mo.object_rows(obj.wTo, obj.shape, obj.kps, obj.frames, True, False)
    ref = mo.construct_object_residual_jacobians(Jc, [0, 1], Hf, res, counts, [f_['wTc'] for f_ in obj.frames], np.eye(3), np.zeros(3),
                                                 False, 22, 2, fix_D_identity=True)
    assert rel(got['res'], ref[2]) < 1e-12 and rel(got['Hf'], ref[1]) < 1e-12 and rel(got['Hx6'], ref[4]) < 1e-12


def _one_car_track(N):
    g = np.load(GOLDEN + '/ref_one_car.npz')
    nf = g['zs'].shape[0]
    frames = []
    for i in range(nf):
        x, y, w, h = g['zb'][i].ravel()
        # the fixture stores the camera poses in float32: their rotation blocks are orthonormal to 1e-7 only.  The reference
        # passes them through Sophus SE3 (a proper rotation); project onto SO(3) so that the rigid inverse the device takes
        # (R^T, -R^T t) and the general inverse of the restatement are the same matrix
        T = g['wTo'][i].astype(np.float64)
        U, _, Vt = np.linalg.svd(T[:3, :3])
        T[:3, :3] = U @ np.diag([1.0, 1.0, np.linalg.det(U @ Vt)]) @ Vt
        T[3] = [0.0, 0.0, 0.0, 1.0]
        frames.append(dict(clone=i if i < N else -1, wTc=T, zs=g['zs'][i].astype(np.float64),
                           bbox=np.array([x, y, x + w, y + h], dtype=np.float64)))   # load_multi_frame_test_data: xywh -> xyminmax
    # the state the rows are evaluated at: ground-truth object pose, mean keypoints, mean ellipsoid (test_utils.cpp:73-132)
    return synth.ObjectTrack(wTo=g['wTq'][0].astype(np.float64), shape=g['ellipsoid_shape'][0].ravel().astype(np.float64),
                             kps=g['mean_shape'][0].astype(np.float64), frames=frames)


@pytest.mark.parametrize('obj_left,new_bbox,vio_left', [(True, False, 0), (False, False, 0), (True, True, 0), (False, True, 1)])
def test_one_car_47_frames_rows_on_the_device(upd, obj_left, new_bbox, vio_left):
    """All 47 real frames of src/tests/data/one_car (test_object_lm_multiframe.cpp:61-125): device rows vs the restatement."""
    N = 47
    obj = _one_car_track(N)
    flags = synth.Flags(use_larvio=0, use_left_perturbation=vio_left)
    win = synth.make_window(N=N, F=2, seed=1, flags=flags, track_len=3)
    Hx, Hf, r, rc, hx6 = object_rows_reference(win, obj, obj_left, new_bbox, vio_left)
    got = upd.object_rows_eval(obj, win.R_b2c[0], win.t_c_b[0], obj_left, new_bbox, vio_left)
    assert got['res'].shape[0] == r.shape[0] and np.array_equal(got['row_clone'], rc)
    assert rel(got['res'], r) < 1e-9 and rel(got['Hx6'], hx6) < 1e-9 and rel(got['Hf'], Hf) < 1e-9


def test_one_car_object_update_from_the_real_frames(upd):
    """The object update (removeLostObjects) built from the 47 real frames, 30 of them in the window, against the mirror's
    literal update; tracks in, dx / P+ out."""
    N = 30
    obj = _one_car_track(N)
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=0.05)   # (pixel-level detections: looser sigma)
    win = synth.make_window(N=N, F=2, seed=2, flags=flags, track_len=3)
    ref = objects_update_reference(win, [obj], win.P, True, False, 0)
    got = upd.update_object_tracks(flags, win.N, [obj], win.P, win.R_b2c[0], win.t_c_b[0], True, False, 0)
    assert got['accept'] == ref['accept']
    assert abs(got['gamma'] - ref['gamma']) < 1e-6 * abs(ref['gamma'])
    if ref['accept']:
        assert got['stats'][0] == ref['dof']
        assert rel(got['dx'], ref['dx']) < 1e-6 and rel(got['P_new'], ref['P_new']) < 1e-6


def test_gram_route_loses_the_weak_direction_of_the_real_object(upd):
    """Why the object path factors Hf by Householder QR: on the real frames cond(Hf) = 2.5e8 (the keypoint rows' gauge, pinned
    only by the bbox rows), chol(Hf^T Hf) drops that direction and the update is a different one (gamma off by percents), while
    the structured QR reproduces the reference's full-U-SVD projection to 1e-6."""
    N = 30
    obj = _one_car_track(N)
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=0.05)
    win = synth.make_window(N=N, F=2, seed=2, flags=flags, track_len=3)
    Hx, Hf, r, rc, hx6 = object_rows_reference(win, obj, True, False, 0)
    sv = np.linalg.svd(Hf, compute_uv=False)
    assert sv[0] / sv[-1] > 1e8
    ref = objects_update_reference(win, [obj], win.P, True, False, 0)
    args = (flags, win.N, [obj], win.P, win.R_b2c[0], win.t_c_b[0], True, False, 0)
    upd._chk(upd.lib.orcvio_msckf_set_option(upd.h, 8, 0), 'set_option')   # ORCVIO_OPT_OBJECT_QR = 0: the Gram route
    try:
        gram = upd.update_object_tracks(*args)
    finally:
        upd._chk(upd.lib.orcvio_msckf_set_option(upd.h, 8, 1), 'set_option')
    qr = upd.update_object_tracks(*args)
    assert abs(qr['gamma'] - ref['gamma']) < 1e-6 * abs(ref['gamma']) and rel(qr['dx'], ref['dx']) < 1e-6
    assert abs(gram['gamma'] - ref['gamma']) > 1e-3 * abs(ref['gamma'])   # a different update
    assert gram['stats'][7] >= 1                                              # ... and it says so: dropped pivot(s) of F
    # the same through pre-evaluated rows (orcvio_msckf_update_objects detects the arrow shape of Hf)
    rows = upd.update_objects(flags, win.N, [dict(row_clone=rc, Hx6=hx6, Hf=Hf, res=r)], win.P)
    assert abs(rows['gamma'] - ref['gamma']) < 1e-6 * abs(ref['gamma']) and rel(rows['dx'], ref['dx']) < 1e-6

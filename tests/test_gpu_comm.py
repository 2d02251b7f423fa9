"""The handle's own RCCL communicator (include/orcvio_msckf.h "Multi-GPU"): on the one GPU of the test box the sharded
entry points run with world size 1 -- communicator, in-place all-gather, rank-ordered sum, replicated solve -- and must
equal the one-shot calls bit for bit on dx / P+ (same kernels, one block).  The N > 1 algebra is covered on CPU by
tests/test_distributed_cpu.py (gloo, world size 2) and on the device by the multi-block finish of test_gpu_parity.py."""
import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import oracle
from helpers import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    u.comm_init(capi.comm_unique_id(), 0, 1)
    assert u.comm_info() == (0, 1)
    yield u
    u.close()


@pytest.mark.parametrize('shape', [dict(N=8, F=40, track_len=(3, 8)), dict(N=30, F=400, track_len=None), dict(N=30, F=700, track_len=None)])
def test_sharded_feature_update_world_1(upd, shape):
    win = synth.make_window(seed=3, outlier_frac=0.1, **shape)
    one = upd.update_features(win)
    got = upd.update_features_sharded(win)
    assert np.array_equal(got['accept'], one['accept'])
    assert rel(got['dx'], one['dx']) < 1e-12 and rel(got['P_new'], one['P_new']) < 1e-12
    if shape['F'] <= 400:
        ref = oracle.msckf_update(win, want_blocks=False, want_K=False)
        assert np.array_equal(got['accept'], ref['accept'])
        assert rel(got['dx'], ref['dx']) < 1e-6 and rel(got['P_new'], ref['P_new']) < 1e-6
    # staged form, repeated (graph capture / replay of the two halves around the collective)
    upd.upload(win)
    for _ in range(4):
        upd.run_update_sharded()
    upd.sync()
    st = upd.download()
    assert rel(st['dx'], one['dx']) < 1e-12 and rel(st['P_new'], one['P_new']) < 1e-12


def test_sharded_object_update_world_1(upd):
    oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=12, F=4, seed=0, flags=oflags, track_len=4)
    objs = synth.make_objects(win, n_objects=5, seed=2, sigma_kp=0.004)
    args = (oflags, win.N, objs, win.P, win.R_b2c[0], win.t_c_b[0], True, True, 0)
    one = upd.update_object_tracks(*args)
    got = upd.update_object_tracks_sharded(*args)
    assert got['accept'] == one['accept'] and abs(got['gamma'] - one['gamma']) <= 1e-9 * abs(one['gamma'])
    assert rel(got['dx'], one['dx']) < 1e-12 and rel(got['P_new'], one['P_new']) < 1e-12
    assert np.array_equal(got['stats'], one['stats'])


def test_sharded_calls_need_a_communicator(built):
    u = capi.MsckfUpdater(device=0, max_clones=8, max_features=64, max_observations=1024)
    try:
        assert u.comm_info() == (0, 0)
        win = synth.make_window(N=6, F=10, seed=1, track_len=(3, 6))
        with pytest.raises(capi.MsckfError):
            u.update_features_sharded(win)
    finally:
        u.close()

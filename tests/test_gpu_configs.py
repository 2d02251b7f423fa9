"""BASELINE.json configurations at FULL size on the device against the oracle (config 4 untested, config 5 only
against itself, config 3's feature half only at reduced size).

  config 3   30 clones, 400 features (OrcVIO right-perturbation rows)                 -- tests/test_gpu_sequence.py (+ 20 objects)
  config 4   30 clones, 2000 features + 100 objects, dealt 4 ways: on the one GPU of the test box the four shards run through
             the staged entry points one after the other (run_local_to x 4 -> run_finish on the four blocks; objects_local_tracks
             x 4 -> objects_finish), exactly what four ranks do around the all-gather
  config 5   kitti_raw.yaml flags (right perturbation, sigma = 1, discard flag), 2000 features
The C oracle needs about 6 s per 2000-feature update on one core."""
import ctypes as C

import numpy as np
import pytest

from orcvio_amd import capi, synth, sharding
from oracle import oracle
from helpers import rel, objects_update_reference

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    yield u
    u.close()


def _d2d(dst, src, nbytes):
    hip = C.CDLL('libamdhip64.so')
    assert hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 3) == 0


@pytest.mark.parametrize('cfg', [4, 5])
def test_2000_features_against_the_oracle(upd, cfg):
    win = synth.config_window(cfg)
    assert win.F == 2000 and win.N == 30
    ref = oracle.msckf_update(win, want_blocks=False, want_K=False)
    got = upd.update_features(win, want_G=True)
    assert np.array_equal(got['accept'], ref['accept'])
    assert rel(got['gamma'], ref['gamma']) < 1e-9
    assert rel(got['dx'], ref['dx']) < TOL and rel(got['P_new'], ref['P_new']) < TOL and rel(got['G'], ref['G']) < TOL
    assert rel(got['P_new'] - win.P, ref['P_new'] - win.P) < TOL
    if cfg == 5:   # discard_large_update_flag: the flag the host applies is reported, P is updated all the same (:4479-4494)
        nv, npos = np.linalg.norm(ref['dx'][3:6]), np.linalg.norm(ref['dx'][6:9])
        assert got['stats'][4] == int(nv > 1.0 or npos > 1.5)


@pytest.mark.parametrize('case', [dict(F=300, track_len=(3, 30), outlier_frac=0.2, seed=3),
                                  dict(F=700, track_len=None, outlier_frac=0.05, seed=4, estimate_extrin=True),
                                  dict(F=1, track_len=5, seed=5)])
def test_split_tracks_front_end_against_the_fused_one(built, monkeypatch, case):
    """k_feature_e + k_feature_gate (the opt-in form for F >= ORCVIO_SPLIT_TRACKS; round 4 took it from 1 800 tracks by default)
    forced on small ragged windows with rejected tracks, against k_feature on the same window: same gate decisions, same update."""
    win = synth.make_window(N=30, flags=synth.Flags(use_larvio=1), **case)
    out = {}
    for name, thr in (('fused', '0'), ('split', '1')):
        monkeypatch.setenv('ORCVIO_SPLIT_TRACKS', thr)   # read when the handle is created (a switch of the diagnostics build)
        u = capi.MsckfUpdater(device=0, max_clones=32, max_features=1024, max_observations=32768, debug_hooks=True)
        try:
            out[name] = u.update_features(win)
        finally:
            u.close()
    a, b = out['fused'], out['split']
    assert np.array_equal(a['accept'], b['accept']) and (case['F'] == 1 or 0 < a['accept'].sum() < win.F)
    assert rel(b['gamma'], a['gamma']) < 1e-9
    assert rel(b['dx'], a['dx']) < 1e-9 and rel(b['P_new'] - win.P, a['P_new'] - win.P) < 1e-9


def test_config4_four_shards_features_and_objects(upd):
    """Config 4 dealt four ways.  Features: each shard's block through run_local_to, the four blocks side by side (what the
    all-gather leaves on every rank), run_finish: equal to the oracle's single 2000-feature update.  Objects: 100 cars dealt
    round-robin, objects_local_tracks x 4, objects_finish on the four blocks with the summed dof: equal to the mirror's update
    of all 100 objects on the P+ the feature update left."""
    import torch
    world = 4
    win = synth.config_window(4)
    ref = oracle.msckf_update(win, want_blocks=False, want_K=False)
    upd.upload(win)
    _, ne = upd.block_ptr()
    gathered = torch.zeros(world * ne, dtype=torch.float64, device='cuda:0')
    accept = np.zeros(win.F, dtype=np.int32)
    for rank in range(world):
        sub, idx = sharding.shard_window(win, rank, world)
        assert 400 <= sub.F <= 600
        upd.upload(sub)
        upd.run_local_to(gathered.data_ptr() + rank * ne * 8)
        upd.sync()
        # the rank's own gate decisions (read through a finish on its own block alone)
        upd.run_finish(gathered.data_ptr() + rank * ne * 8, 1)
        upd.sync()
        accept[idx] = upd.download()['accept']
    assert np.array_equal(accept, ref['accept'])
    upd.run_finish(gathered.data_ptr(), world)   # (the last shard's upload is still the current problem: same window, same P)
    upd.sync()
    got = upd.download()
    assert rel(got['dx'], ref['dx']) < TOL and rel(got['P_new'], ref['P_new']) < TOL

    # ---- 100 objects on the covariance the feature update left ---------------------------------------------------------
    oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    owin = synth.make_window(N=30, F=4, seed=0, flags=oflags, track_len=4)
    objs = synth.make_objects(owin, n_objects=100, seed=11, sigma_kp=0.004)
    P1 = ref['P_new']
    oref = objects_update_reference(owin, objs, P1, True, False, 0)   # left perturbation, old bbox residual (consistent Jacobians: accepted)
    assert oref['accept'] == 1
    blocks = torch.zeros(world * ne, dtype=torch.float64, device='cuda:0')
    fl = capi.make_flags(oflags)
    dof = 0
    Pc = np.ascontiguousarray(P1)
    for rank in range(world):
        mine = objs[rank::world]
        ef, arr, keep = upd._object_tracks(mine, owin.R_b2c[0], owin.t_c_b[0], True, False, 0, False)
        d = C.c_int32(0)
        rc = upd.lib.orcvio_msckf_objects_local_tracks(upd.h, C.byref(fl), C.byref(ef), owin.N, arr, len(mine), capi._d(Pc),
                                                       C.c_void_p(blocks.data_ptr() + rank * ne * 8), C.byref(d), None)
        assert rc == 0
        upd.sync()
        dof += d.value
    upd.n = owin.n
    upd.objects_finish(blocks.data_ptr(), world, dof)
    ogot = upd.objects_download()
    assert dof == oref['dof']
    assert ogot['accept'] == oref['accept']
    assert abs(ogot['gamma'] - oref['gamma']) < 1e-6 * abs(oref['gamma'])
    assert rel(ogot['dx'], oref['dx']) < TOL and rel(ogot['P_new'], oref['P_new']) < TOL

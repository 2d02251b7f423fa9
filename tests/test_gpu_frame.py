"""orcvio_msckf_io_update_frame: the feature update and the object update of one frame in one call (System::imageCallback,
ros_wrapper/src/orcvio/src/System.cpp:548-554), the object tracks' compression running beside the feature update's solve.
It must equal the two calls in sequence -- orcvio_msckf_io_update(commit) then orcvio_msckf_update_object_tracks on the
resident covariance -- bit for bit (same kernels, same arguments), and the oracle run step by step within the tolerance."""
import dataclasses
import os
import subprocess
import sys

import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import oracle
from helpers import rel, objects_update_reference

pytestmark = pytest.mark.gpu
TOL = 1e-6
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    yield u
    u.close()


def _two_calls(upd, win, objs, new_bbox=False):
    """the reference form: io_update with its commit, then the object update on the resident covariance, then its commit"""
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=False)
    upd.io_fill(io, win, with_P=False)
    stats = upd.io_update(want_P=False, commit=True)
    f = dict(dx=io['dx'].copy(), gamma=io['gamma'].copy(), accept=io['accept'].copy(), stats=stats)
    o = upd.update_object_tracks(win.flags, win.N, objs, None, win.R_b2c[0], win.t_c_b[0], True, new_bbox, 0)
    upd.cov_commit()
    return f, o


def _frame(upd, win, objs, new_bbox=False):
    return upd.update_frame(win, win.flags, objs, win.R_b2c[0], win.t_c_b[0], True, new_bbox, 0)


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def _chained(upd):
    return upd.counters()['chained_frames']


def _obj_agree(chained, a, b):
    """the object half of the one-call form against the two calls: bit for bit -- unless the frame's object solve ran CHAINED to the
    feature update's factor (windows from six block steps: another factor of the same matrix), then to rounding"""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if not chained:
        return np.array_equal(a, b, equal_nan=True)
    return float(np.linalg.norm(a - b)) <= 1e-9 * max(float(np.linalg.norm(b)), 1e-300)


@pytest.mark.parametrize('N,F,nobj,new_bbox', [(10, 60, 3, False), (30, 400, 20, False), (30, 400, 20, True), (20, 120, 1, False)])
def test_frame_equals_the_two_calls_and_the_oracle(upd, N, F, nobj, new_bbox):
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=N, F=F, seed=4, flags=flags, track_len=None if N == 30 else (3, N), outlier_frac=0.05)
    objs = synth.make_objects(win, n_objects=nobj, seed=2, sigma_kp=0.004)
    upd.cov_set(win.P)
    f0, o0 = _two_calls(upd, win, objs, new_bbox)
    P0 = upd.cov_get()
    upd.cov_set(win.P)
    c0 = _chained(upd)
    f1, o1 = _frame(upd, win, objs, new_bbox)
    ch = _chained(upd) > c0
    assert ch == (N >= 20)   # (the chained object solve: windows from six block steps)
    P1 = upd.cov_get()
    # the two forms: identical (the object half to rounding where its solve ran chained)
    assert _same(f1['dx'], f0['dx']) and _same(f1['gamma'], f0['gamma']) and _same(f1['accept'], f0['accept'])
    assert _same(f1['stats'][:5], f0['stats'][:5])
    assert o1['accept'] == o0['accept'] and _obj_agree(ch, o1['gamma'], o0['gamma']) and _obj_agree(ch, o1['dx'], o0['dx']) and _same(o1['stats'], o0['stats'])
    assert _obj_agree(ch, P1, P0)
    # ... and the oracle, step by step
    ref1 = oracle.msckf_update(win, want_blocks=False, want_K=False)
    ref2 = objects_update_reference(win, objs, ref1['P_new'], True, new_bbox, 0)
    assert np.array_equal(f1['accept'], ref1['accept']) and rel(f1['dx'], ref1['dx']) < TOL
    assert o1['accept'] == ref2['accept'] and abs(o1['gamma'] - ref2['gamma']) < 1e-6 * abs(ref2['gamma'])
    assert rel(o1['dx'], ref2['dx']) < TOL and rel(P1, ref2['P_new']) < TOL


@pytest.mark.parametrize('case', ['unfused_front', 'many_tracks', 'leg46_wide', 'no_tracks'])
def test_frame_when_the_feature_half_assembles_A_in_memory(upd, case):
    """the overlapped frame compresses the objects into the handle's A block on a second stream, which is a race
    whenever the feature half itself writes and reads that block (ORCVIO_OPT_FUSED_FRONT = 0, more tracks than the fused front end
    holds, NA > 192, no tracks at all).  Those frames now run their halves in sequence: bit for bit the two calls, ten times over."""
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0, leg_dim=46 if case == 'leg46_wide' else 22)
    N, F = {'unfused_front': (12, 90), 'many_tracks': (12, 1100), 'leg46_wide': (28, 60), 'no_tracks': (10, 0)}[case]
    win = synth.make_window(N=N, F=max(F, 4), seed=5, flags=flags, track_len=(3, min(N, 8)), outlier_frac=0.05)
    objs = synth.make_objects(win, n_objects=3, seed=2, sigma_kp=0.004)
    if F == 0:
        win = dataclasses.replace(win, p_w=win.p_w[:0], obs_ptr=win.obs_ptr[:1], obs_clone=win.obs_clone[:0], obs_z=win.obs_z[:0], obs_zvel=win.obs_zvel[:0])
    if case == 'unfused_front':
        upd.set_fused_front(False)
    try:
        upd.cov_set(win.P)
        f0, o0 = _two_calls(upd, win, objs)
        P0 = upd.cov_get()
        for _ in range(10):
            upd.cov_set(win.P)
            f1, o1 = _frame(upd, win, objs)
            assert _same(f1['dx'], f0['dx']) and _same(f1['accept'], f0['accept'])
            assert o1['accept'] == o0['accept'] and _same(o1['dx'], o0['dx']) and _same(o1['gamma'], o0['gamma'])
            assert _same(upd.cov_get(), P0)
    finally:
        upd.set_fused_front(True)
    if F > 0:
        ref1 = oracle.msckf_update(win, want_blocks=False, want_K=False)
        ref2 = objects_update_reference(win, objs, ref1['P_new'], True, False, 0)
        assert rel(f0['dx'], ref1['dx']) < TOL and o0['accept'] == ref2['accept'] and rel(o0['dx'], ref2['dx']) < TOL


def test_frames_in_a_row(upd):
    """six frames on the covariance the previous one left (the launch-graph cache replays the feature half; the spare factor
    buffer alternates): every frame equals the two calls"""
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=12, F=90, seed=7, flags=flags, track_len=(3, 12), outlier_frac=0.05)
    wins = [dataclasses.replace(synth.make_window(N=12, F=90, seed=70 + k, flags=flags, track_len=(3, 12)), P=win.P) for k in range(6)]
    objs = [synth.make_objects(win, n_objects=2 + (k % 3), seed=30 + k, sigma_kp=0.004) for k in range(6)]
    upd.cov_set(win.P)
    seq = [_two_calls(upd, w, o) for w, o in zip(wins, objs)]
    P_seq = upd.cov_get()
    upd.cov_set(win.P)
    for k, (w, o) in enumerate(zip(wins, objs)):
        f, ob = _frame(upd, w, o)
        assert _same(f['dx'], seq[k][0]['dx']) and _same(f['accept'], seq[k][0]['accept']), k
        assert ob['accept'] == seq[k][1]['accept'] and _same(ob['dx'], seq[k][1]['dx']), k
    assert _same(upd.cov_get(), P_seq)


def test_frame_without_objects_and_with_a_rejected_object_update(upd):
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=10, F=40, seed=4, flags=flags, track_len=(3, 10))
    ref = oracle.msckf_update(win, want_blocks=False, want_K=False)
    upd.cov_set(win.P)
    f, o = _frame(upd, win, [])
    assert rel(f['dx'], ref['dx']) < TOL and o['accept'] == 0 and not np.any(o['dx'])
    assert rel(upd.cov_get(), ref['P_new']) < TOL   # the feature update stands, nothing else happened
    bad = synth.make_objects(win, n_objects=2, seed=6, sigma_kp=0.2)   # 25 sigma keypoint noise: the gate fails
    upd.cov_set(win.P)
    f, o = _frame(upd, win, bad)
    assert o['accept'] == 0 and rel(f['dx'], ref['dx']) < TOL
    assert rel(upd.cov_get(), ref['P_new']) < TOL
    # the factor the feature half committed is still the resident one: the next update on it equals the update on P+
    win2 = dataclasses.replace(win, P=ref['P_new'])
    again = upd.update_features(win2, resident_cov=True, want_P=True)
    ref2 = oracle.msckf_update(win2, want_blocks=False, want_K=False)
    assert rel(again['dx'], ref2['dx']) < TOL and rel(again['P_new'], ref2['P_new']) < TOL


def test_refused_feature_half_applies_nothing(upd):
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=10, F=40, seed=4, flags=flags, track_len=(3, 10))
    objs = synth.make_objects(win, n_objects=2, seed=2, sigma_kp=0.004)
    bad = win.P.copy()
    bad[20, 20] = np.nan   # a non-finite prior: the device refuses the feature update and its commit
    upd.cov_set(bad)
    with pytest.raises(capi.MsckfError) as e:
        _frame(upd, win, objs)
    assert e.value.code == 6   # ORCVIO_ERR_NOT_SPD
    assert np.array_equal(upd.cov_get(), bad, equal_nan=True)   # neither half was applied
    # the handle goes on: the same frame on a sound prior equals the two calls
    upd.cov_set(win.P)
    f0, o0 = _two_calls(upd, win, objs)
    upd.cov_set(win.P)
    f1, o1 = _frame(upd, win, objs)
    assert _same(f1['dx'], f0['dx']) and _same(o1['dx'], o0['dx']) and o1['accept'] == o0['accept']


def test_refused_feature_half_with_a_finite_prior_refuses_the_object_commit_too(upd):
    """a FINITE prior on which only the feature half fails -- its noise so small that sigma^2 is lost beside
    L^T A L and chol(M1) meets a non-positive pivot -- while the object half, with a sane noise of its own, would go through.  The
    object half's commit rides in its own epilogue launch, enqueued before the host has seen the feature half's outcome: it must
    refuse itself on the feature half's status words (kept by the feature half's epilogue), or the resident covariance would hold an
    object update applied to a prior the call reports as untouched."""
    import dataclasses
    oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    fflags = dataclasses.replace(oflags, noise_feature=1e-40)
    win = synth.make_window(N=10, F=40, seed=4, flags=oflags, track_len=(3, 10))
    objs = synth.make_objects(win, n_objects=2, seed=2, sigma_kp=0.004)
    wf = dataclasses.replace(win, flags=fflags)
    upd.cov_set(win.P)
    try:
        upd.update_frame(wf, oflags, objs, win.R_b2c[0], win.t_c_b[0], True, False, 0)
    except capi.MsckfError as e:
        assert e.code == 6, e
        assert np.array_equal(upd.cov_get(), win.P)   # nothing of the frame was applied: the object half's commit refused itself
    else:
        pytest.skip('chol(M1) went through at sigma = 1e-40 on this window: the refusal could not be provoked')
    # the handle goes on
    upd.cov_set(win.P)
    f0, o0 = _two_calls(upd, win, objs)
    upd.cov_set(win.P)
    f1, o1 = _frame(upd, win, objs)
    assert _same(f1['dx'], f0['dx']) and _same(o1['dx'], o0['dx'])


def test_refused_object_half_keeps_the_feature_update(upd):
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=10, F=40, seed=4, flags=flags, track_len=(3, 10))
    objs = synth.make_objects(win, n_objects=2, seed=2, sigma_kp=0.004)
    objs[1].frames[0]['clone'] = 99   # a frame outside the window
    ref = oracle.msckf_update(win, want_blocks=False, want_K=False)
    upd.cov_set(win.P)
    with pytest.raises(capi.MsckfError) as e:
        _frame(upd, win, objs)
    assert e.value.code == 1   # ORCVIO_ERR_INVALID from the object half
    assert rel(upd.cov_get(), ref['P_new']) < TOL   # the feature update was applied and committed
    with pytest.raises(capi.MsckfError):
        upd.cov_commit()                             # ... and nothing else is left to commit
    assert rel(upd.cov_get(), ref['P_new']) < TOL


def test_frame_needs_the_resident_covariance_in_the_arena(upd):
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=8, F=20, seed=4, flags=flags, track_len=(3, 8))
    upd.cov_set(win.P)
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=True)
    upd.io_fill(io, win, with_P=True)
    call, _ = upd.make_frame_call(win, win.flags, [], win.R_b2c[0], win.t_c_b[0], True, False, 0)
    with pytest.raises(capi.MsckfError) as e:
        call()
    assert e.value.code == 1


def test_the_sequential_switch_gives_the_same_frame(built):
    """ORCVIO_FRAME_OVERLAP=0 (read once per process): the two halves one behind the other -- same numbers"""
    code = (
        "import sys, json\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "import numpy as np\n"
        "from orcvio_amd import capi, synth\n"
        "u = capi.MsckfUpdater(device=0, max_clones=32, max_features=512, max_observations=16384)\n"
        "fl = synth.Flags(use_larvio=0, use_left_perturbation=0)\n"
        "w = synth.make_window(N=10, F=60, seed=4, flags=fl, track_len=(3, 10), outlier_frac=0.05)\n"
        "ob = synth.make_objects(w, n_objects=3, seed=2, sigma_kp=0.004)\n"
        "u.cov_set(w.P)\n"
        "f, o = u.update_frame(w, w.flags, ob, w.R_b2c[0], w.t_c_b[0], True, False, 0)\n"
        "print('OUT', json.dumps(dict(fdx=f['dx'].tolist(), odx=o['dx'].tolist(), acc=int(o['accept']), P=u.cov_get().tolist())))\n")
    outs = []
    for mode in ('1', '0'):
        p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ, ORCVIO_FRAME_OVERLAP=mode), timeout=600)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
        import json
        outs.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('OUT')][-1][4:]))
    assert outs[0] == outs[1] and outs[0]['acc'] == 1


def _fallbacks(u):
    import ctypes as C
    u.lib.orcvio_msckf_debug_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    v = C.c_int32(0)
    assert u.lib.orcvio_msckf_debug_read(u.h, 10, C.byref(v), 4) == 0
    return v.value


def test_frame_survives_a_kernel_that_holds_half_the_device(built):
    """The feature half's fused front end needs its workgroups resident at once; with 128 CUs held by somebody else it loses its
    in-launch hand-off -- the frame is then run again as the two calls, inside the same call, and the object half that ran
    meanwhile on the un-committed prior is discarded.  In normal frames the object kernels beside the feature update never
    cause that (fall-back counter stays at 0)."""
    import ctypes as C
    import time
    u = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536, debug_hooks=True)
    try:
        flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
        win = synth.make_window(N=30, F=400, seed=4, flags=flags, track_len=None, outlier_frac=0.05)
        objs = synth.make_objects(win, n_objects=20, seed=2, sigma_kp=0.004)
        u.cov_set(win.P)
        f0, o0 = _two_calls(u, win, objs)
        P0 = u.cov_get()
        for _ in range(5):   # ordinary frames: the objects' kernels beside the feature update never strand its front end
            u.cov_set(win.P)
            f1, o1 = _frame(u, win, objs)
            assert _same(f1['dx'], f0['dx']) and _obj_agree(True, o1['dx'], o0['dx'])   # (30 clones: the object solve runs chained)
        assert _fallbacks(u) == 0 and _chained(u) == 5
        u.cov_set(win.P)
        u.lib.orcvio_msckf_debug_occupy.argtypes = [C.c_void_p, C.c_int32, C.c_double]
        assert u.lib.orcvio_msckf_debug_occupy(u.h, 128, 400.0) == 0   # 128 CUs held for 0.4 s on another stream
        t0 = time.perf_counter()
        f2, o2 = _frame(u, win, objs)
        dt = time.perf_counter() - t0
        assert _fallbacks(u) == 1 and dt < 2.5
        assert np.array_equal(f2['accept'], f0['accept']) and rel(f2['dx'], f0['dx']) < 1e-9
        assert o2['accept'] == o0['accept'] and rel(o2['dx'], o0['dx']) < 1e-9
        assert rel(u.cov_get(), P0) < 1e-9
        u.sync()
        time.sleep(0.5)
        u.cov_set(win.P)
        f3, o3 = _frame(u, win, objs)   # the occupying kernel is gone: the overlapped form again
        assert _same(f3['dx'], f0['dx']) and _obj_agree(True, o3['dx'], o0['dx']) and _fallbacks(u) == 1
    finally:
        u.close()


def test_frame_without_feature_tracks(upd):
    """no lost feature in this frame, objects all the same: the feature half is an update with no rows (P+ = P, committed), the
    object half runs on it"""
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=10, F=20, seed=4, flags=flags, track_len=(3, 10))
    none = dataclasses.replace(win, p_w=win.p_w[:0].copy(), obs_ptr=np.zeros(1, dtype=np.int32), obs_clone=win.obs_clone[:0].copy(),
                               obs_z=win.obs_z[:0].copy(), obs_zvel=win.obs_zvel[:0].copy())
    objs = synth.make_objects(win, n_objects=2, seed=2, sigma_kp=0.004)
    ref = objects_update_reference(win, objs, win.P, True, False, 0)
    upd.cov_set(win.P)
    f, o = _frame(upd, none, objs)
    assert not f['stats'][3] and not np.any(f['dx'])
    assert o['accept'] == ref['accept'] == 1 and rel(o['dx'], ref['dx']) < TOL and rel(upd.cov_get(), ref['P_new']) < TOL


@pytest.mark.parametrize('N,F,nobj,new_bbox', [(30, 400, 20, False), (30, 400, 20, True), (20, 120, 1, False), (12, 90, 3, False)])
def test_frame_with_the_chained_object_solve(built, monkeypatch, N, F, nobj, new_bbox):
    """The default from six block steps (ORCVIO_FRAME_CHAIN=0 at create: off): the object solve of the frame runs from the FEATURE update's prior factor and its M
    (M12 = M1 + L_a^T A' L_a: the sequential update of the reference by Woodbury), on a stream and solve buffers of its own, beside
    the feature half's solve and commit.  Another factor of the same matrix: the results agree with the two calls to rounding, not
    bit for bit; the gate decision, the committed covariance and the oracle step by step are checked, five frames in a row."""
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=N, F=F, seed=4, flags=flags, track_len=None if N == 30 else (3, N), outlier_frac=0.05)
    objs = synth.make_objects(win, n_objects=nobj, seed=2, sigma_kp=0.004)
    monkeypatch.setenv('ORCVIO_FRAME_CHAIN', '0')
    plain = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    try:
        plain.cov_set(win.P)
        f0, o0 = _frame(plain, win, objs, new_bbox)
        P0 = plain.cov_get()
        assert plain.counters()['chained_frames'] == 0
        plain.cov_set(win.P)   # ... and with the chain off the one-call form IS the two calls, bit for bit
        f00, o00 = _two_calls(plain, win, objs, new_bbox)
        assert _same(o00['dx'], o0['dx']) and _same(plain.cov_get(), P0) and _same(f00['dx'], f0['dx'])
    finally:
        plain.close()
    monkeypatch.delenv('ORCVIO_FRAME_CHAIN')
    upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    try:
        ref1 = oracle.msckf_update(win, want_blocks=False, want_K=False)
        ref2 = objects_update_reference(win, objs, ref1['P_new'], True, new_bbox, 0)
        for it in range(5):
            upd.cov_set(win.P)
            f1, o1 = _frame(upd, win, objs, new_bbox)
            P1 = upd.cov_get()
            assert _same(f1['dx'], f0['dx']) and _same(f1['gamma'], f0['gamma']) and _same(f1['accept'], f0['accept'])   # the feature half is the same launches
            assert o1['accept'] == o0['accept'] and _same(o1['stats'], o0['stats'])
            assert abs(o1['gamma'] - o0['gamma']) < 1e-9 * abs(o0['gamma'])
            assert rel(o1['dx'], o0['dx']) < 1e-9 and rel(P1, P0) < 1e-10
            assert o1['accept'] == ref2['accept'] and rel(o1['dx'], ref2['dx']) < TOL and rel(P1, ref2['P_new']) < TOL
        assert upd.counters()['chained_frames'] == (5 if N >= 20 else 0)
        # the factor the chained commit left is a factor of the committed covariance: the next frame runs on it
        f2, o2 = _frame(upd, win, objs, new_bbox)
        assert np.isfinite(o2['dx']).all() and np.isfinite(f2['dx']).all()
    finally:
        upd.close()


def test_frame_runs_again_as_two_calls_when_an_in_launch_hand_off_is_lost(built, monkeypatch):
    """ORCVIO_LA_SPIN=0: every wait on another workgroup inside the look-ahead solve launches (the far workgroups' block rows, the finish
    workgroups' counter) gives up at once.  The frame call notices on the feature half's status word, drains, rolls its book-keeping
    back and runs the frame again as the two calls, with plain launches and the one-workgroup factorisation -- the caller gets the
    frame's results (equal to the two calls'), one fall-back counted per frame, and the handle stays usable."""
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=30, F=400, seed=4, flags=flags, outlier_frac=0.05)
    objs = synth.make_objects(win, n_objects=20, seed=2, sigma_kp=0.004)
    ref = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    try:
        ref.cov_set(win.P)
        f0, o0 = _two_calls(ref, win, objs)
        P0 = ref.cov_get()
    finally:
        ref.close()
    monkeypatch.setenv('ORCVIO_LA_SPIN', '0')
    upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    try:
        for it in range(3):
            upd.cov_set(win.P)
            f1, o1 = _frame(upd, win, objs)
            P1 = upd.cov_get()
            assert np.array_equal(f1['accept'], f0['accept']) and rel(f1['dx'], f0['dx']) < 1e-9
            assert o1['accept'] == o0['accept'] == 1 and rel(o1['dx'], o0['dx']) < 1e-9 and rel(P1, P0) < 1e-10
            assert upd.counters()['front_fallbacks'] >= it + 1
    finally:
        upd.close()


@pytest.mark.parametrize('what', ['chi2_prob', 'noise_feature'])
def test_frame_with_object_flags_that_differ_from_the_feature_flags(built, what):
    """The object update of the frame takes its OWN flags: a chi-square probability of its own changes the gate's threshold (the chained
    solve's in-launch gate must use it, not the feature half's), a measurement noise of its own takes the frame off the chain
    (M12 = M1 + L_a^T A' L_a holds for one sigma): either way the one-call form agrees with the two calls."""
    import dataclasses
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=30, F=400, seed=4, flags=flags, outlier_frac=0.05)
    objs = synth.make_objects(win, n_objects=20, seed=2, sigma_kp=0.004)
    upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    try:
        # the threshold that sits just below / above this frame's gamma: the decision flips with the object flags' probability
        upd.cov_set(win.P)
        _, o_ref = _two_calls(upd, win, objs)
        assert o_ref['accept'] == 1
        for val in ((1e-9, 0.95) if what == 'chi2_prob' else (0.008 * 1.5,)):
            oflags = dataclasses.replace(flags, **{what: val})
            upd.cov_set(win.P)
            io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=False)
            upd.io_fill(io, win, with_P=False)
            upd.io_update(want_P=False, commit=True)
            o0 = upd.update_object_tracks(oflags, win.N, objs, None, win.R_b2c[0], win.t_c_b[0], True, False, 0)
            upd.cov_commit()
            P0 = upd.cov_get()
            upd.cov_set(win.P)
            c0 = _chained(upd)
            f1, o1 = upd.update_frame(win, oflags, objs, win.R_b2c[0], win.t_c_b[0], True, False, 0)
            P1 = upd.cov_get()
            ch = _chained(upd) > c0
            assert ch == (what == 'chi2_prob')
            assert o1['accept'] == o0['accept'] and _obj_agree(ch, o1['gamma'], o0['gamma']) and _obj_agree(ch, o1['dx'], o0['dx'])
            assert _obj_agree(ch, P1, P0)
            if what == 'chi2_prob':
                assert o1['accept'] == (1 if val > 0.5 else 0)   # (a probability of 1e-9 rejects everything: P++ = P+, dx = 0)
    finally:
        upd.close()


def test_frame_with_the_object_tracks_staged_ahead(built):
    """orcvio_msckf_io_stage_object_tracks between io_fill and the frame call: the frame call finds the scan and the pinned staging arena
    done and goes straight to the launches -- the same results bit for bit, counted in counters()['prestaged_frames']; a frame call with
    OTHER tracks than the staged ones (or a second call on one staging) stages for itself."""
    flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    win = synth.make_window(N=30, F=400, seed=4, flags=flags, outlier_frac=0.05)
    objs = synth.make_objects(win, n_objects=20, seed=2, sigma_kp=0.004)
    objs2 = synth.make_objects(win, n_objects=7, seed=5, sigma_kp=0.004)
    upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    nobs = int(win.obs_ptr[-1])

    def frame(ob, stage, stage_other=None):
        upd.cov_set(win.P)
        io = upd.io_begin(win.flags, win.N, win.F, nobs, with_P=False)
        upd.io_fill(io, win, with_P=False)
        call, outs = upd.make_frame_call(win, win.flags, ob, win.R_b2c[0], win.t_c_b[0], True, False, 0)
        keep = None
        if stage_other is not None:   # stage OTHER tracks: the call below must not use them
            keep = upd.make_frame_call(win, win.flags, stage_other, win.R_b2c[0], win.t_c_b[0], True, False, 0)
            keep[0].stage()
        if stage:
            call.stage()
        call()
        f, o = outs()
        return f, o, upd.cov_get()
    try:
        f0, o0, P0 = frame(objs, False)
        n0 = upd.counters()['prestaged_frames']
        for it in range(3):
            f1, o1, P1 = frame(objs, True)
            assert upd.counters()['prestaged_frames'] == n0 + it + 1
            assert _same(f1['dx'], f0['dx']) and _same(o1['dx'], o0['dx']) and _same(o1['gamma'], o0['gamma']) and o1['accept'] == o0['accept'] == 1
            assert _same(P1, P0) and _same(o1['stats'], o0['stats'])
        n1 = upd.counters()['prestaged_frames']
        f2, o2, P2 = frame(objs, False, stage_other=objs2)   # the staging is of other tracks: ignored
        assert upd.counters()['prestaged_frames'] == n1
        assert _same(o2['dx'], o0['dx']) and _same(P2, P0)
        f3, o3, P3 = frame(objs2, False)
        f4, o4, P4 = frame(objs2, True)
        assert upd.counters()['prestaged_frames'] == n1 + 1
        assert _same(o4['dx'], o3['dx']) and _same(P4, P3) and o4['accept'] == o3['accept']
    finally:
        upd.close()

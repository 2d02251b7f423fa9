"""Robustness of the fused front end (VERDICT r1 "weak: robustness"): k_front meets at an in-launch device-wide counter and
therefore needs all its workgroups resident at once.  A kernel of somebody else holding compute units (RCCL on another stream,
a second handle, a torch op) can strand part of them: the wait is bounded, and the update is then RE-RUN on the forked path
inside the same call instead of being reported as an error."""
import ctypes as C
import time

import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import oracle
from helpers import rel

pytestmark = pytest.mark.gpu


def _fallbacks(upd):
    upd.lib.orcvio_msckf_debug_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    v = C.c_int32(0)
    assert upd.lib.orcvio_msckf_debug_read(upd.h, 10, C.byref(v), 4) == 0
    return v.value


def test_update_survives_a_kernel_that_holds_half_the_device(built):
    upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536, debug_hooks=True)   # (occupy / counter hooks)
    try:
        win = synth.config_window(2)
        ref = oracle.msckf_update(win, want_blocks=False, want_K=False)
        good = upd.update_features(win)                      # warm: code objects loaded, graph not yet captured
        assert rel(good['dx'], ref['dx']) < 1e-6 and _fallbacks(upd) == 0
        upd.lib.orcvio_msckf_debug_occupy.argtypes = [C.c_void_p, C.c_int32, C.c_double]
        assert upd.lib.orcvio_msckf_debug_occupy(upd.h, 128, 400.0) == 0   # 128 CUs held for 0.4 s on another stream
        t0 = time.perf_counter()
        got = upd.update_features(win)                       # k_front cannot get its 201 workgroups resident
        dt = time.perf_counter() - t0
        assert np.array_equal(got['accept'], ref['accept'])
        assert rel(got['dx'], ref['dx']) < 1e-6 and rel(got['P_new'], ref['P_new']) < 1e-6
        assert _fallbacks(upd) == 1, 'the stranded launch should have been re-run on the forked path'
        assert upd.counters()['front_fallbacks'] == 1   # ... and the caller can see it through the product ABI (orcvio_msckf_counters)
        upd.set_fused_front(False)                           # bit-identical to the unfused form (the re-run IS the unfused form)
        unf = upd.update_features(win)
        upd.set_fused_front(True)
        assert np.array_equal(unf['dx'], got['dx']) and np.array_equal(unf['P_new'], got['P_new']) and np.array_equal(unf['gamma'], got['gamma'])
        assert dt < 2.0
        upd.sync()
        time.sleep(0.5)                                      # the occupying kernel is gone: the fused path again, no fallback
        again = upd.update_features(win)
        assert rel(again['dx'], ref['dx']) < 1e-6 and _fallbacks(upd) == 1
    finally:
        upd.close()


def test_lookahead_solve_is_bit_identical_and_falls_back_when_a_hand_off_is_lost(built, monkeypatch):
    """k_potrf_solve_la (far workgroups bring the block rows forward) against k_potrf_solve (one workgroup): the same panels in the
    same order, bit-identical updates.  With ORCVIO_LA_SPIN = 0 every wait on another workgroup gives up at once: the launch flags
    itself, and the update is run again through k_potrf_solve inside the same call -- same results, one fall-back counted."""
    win = synth.config_window(2)
    ref = oracle.msckf_update(win, want_blocks=False, want_K=False)
    upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=512, max_observations=16384)
    try:
        la = upd.update_features(win)
        upd.set_lookahead_solve(0)
        one = upd.update_features(win)
        upd.set_lookahead_solve(2)
        la2 = upd.update_features(win)
        for got in (la, one, la2):
            assert rel(got['dx'], ref['dx']) < 1e-6 and rel(got['P_new'], ref['P_new']) < 1e-6
        assert np.array_equal(la['dx'], one['dx']) and np.array_equal(la['P_new'], one['P_new'])
        assert np.array_equal(la2['dx'], one['dx']) and np.array_equal(la2['P_new'], one['P_new'])
        assert upd.counters()['front_fallbacks'] == 0
        upd.set_fused_front(False)   # (the re-run of a lost hand-off is the forked form, with k_potrf_solve)
        upd.set_lookahead_solve(0)
        unf = upd.update_features(win)
    finally:
        upd.close()
    monkeypatch.setenv('ORCVIO_LA_SPIN', '0')
    upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=512, max_observations=16384)
    try:
        for it in range(3):   # (plain launches, then the captured graph)
            got = upd.update_features(win)
            assert np.array_equal(got['dx'], unf['dx']) and np.array_equal(got['P_new'], unf['P_new'])
            assert rel(got['dx'], one['dx']) < 1e-10 and rel(got['P_new'], one['P_new']) < 1e-12
            assert upd.counters()['front_fallbacks'] == it + 1
    finally:
        upd.close()


@pytest.mark.parametrize('cfg', [1, 2])
def test_finish_inside_the_solve_launch_is_bit_identical_to_the_finish_launch(built, monkeypatch, cfg):
    """P+ = s2 Z^T Z, dx (and an object update's chi-square gate) by finish workgroups of k_potrf_solve_la (LaFin: they wait for the solver
    workgroups' counter and read Z past the caches; ORCVIO_FUSE_FINISH=2: every update -- by default the chained frame call only, where
    it pays) against the k_finish_sqrt launch behind it (ORCVIO_FUSE_FINISH=0): the same tiles, the same split of K, the same order --
    the same bits, for the feature update (plain launches and the captured graph) and for the object update on the covariance it leaves."""
    win = synth.config_window(cfg)
    ref = oracle.msckf_update(win, want_blocks=False, want_K=False)
    oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    owin = synth.make_window(N=win.N, F=4, seed=0, flags=oflags, track_len=4)
    objs = synth.make_objects(owin, n_objects=6, seed=1, sigma_kp=0.004)

    def run(mode):
        monkeypatch.setenv('ORCVIO_FUSE_FINISH', mode)   # (a switch of the diagnostics build)
        upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=512, max_observations=16384, debug_hooks=True)
        try:
            feats = [upd.update_features(win) for _ in range(4)]
            upd.cov_set(win.P)
            obj = upd.update_object_tracks(oflags, owin.N, objs, None, owin.R_b2c[0], owin.t_c_b[0], True, False, 0)
            return feats, obj
        finally:
            upd.close()

    fused, ofused = run('2')
    plain, oplain = run('0')
    assert rel(plain[0]['dx'], ref['dx']) < 1e-6 and rel(plain[0]['P_new'], ref['P_new']) < 1e-6
    for got in fused + plain[1:]:
        assert np.array_equal(got['dx'], plain[0]['dx']) and np.array_equal(got['P_new'], plain[0]['P_new'])
        assert np.array_equal(got['accept'], plain[0]['accept'])
    assert oplain['accept'] == 1 and ofused['accept'] == oplain['accept']
    assert np.array_equal(ofused['gamma'], oplain['gamma']) and np.array_equal(ofused['dx'], oplain['dx'])
    if 'P_new' in oplain:
        assert np.array_equal(ofused['P_new'], oplain['P_new'])


@pytest.mark.parametrize('cfg', [1, 2])
def test_U_inside_k_front_is_bit_identical_to_the_product_launch(built, monkeypatch, cfg):
    """U = [A; b^T] L_a by the feature workgroups of k_front, behind the Grams, each tile as soon as the prior's factorisation has
    published the block row of the factor it needs (FrontUArgs; ORCVIO_FRONT_U=1: opt-in, it measured slower) against the k_gemm_asmA launch
    behind k_front (the default): the same body, the same bits -- with the prior factored inside the launch (plain launches, then the captured
    graph) and with its factor resident (cov_prefactor: nothing to wait for)."""
    win = synth.config_window(cfg)
    ref = oracle.msckf_update(win, want_blocks=False, want_K=False)
    nobs = int(win.obs_ptr[-1])

    def run(mode):
        monkeypatch.setenv('ORCVIO_FRONT_U', mode)   # (a switch of the diagnostics build)
        upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=512, max_observations=16384, debug_hooks=True)
        try:
            feats = [upd.update_features(win) for _ in range(4)]
            upd.cov_set(win.P)
            upd.cov_prefactor()
            io = upd.io_begin(win.flags, win.N, win.F, nobs, with_P=False)
            upd.io_fill(io, win, with_P=False)
            upd.io_update(want_P=False, commit=True)
            return feats, io['dx'].copy(), upd.cov_get(), upd.counters()['front_fallbacks']
        finally:
            upd.close()

    inside, dx_in, P_in, fb_in = run('1')
    launch, dx_l, P_l, fb_l = run('0')
    assert fb_in == 0 and fb_l == 0
    assert rel(launch[0]['dx'], ref['dx']) < 1e-6 and rel(launch[0]['P_new'], ref['P_new']) < 1e-6
    for got in inside + launch[1:]:
        assert np.array_equal(got['dx'], launch[0]['dx']) and np.array_equal(got['P_new'], launch[0]['P_new'])
        assert np.array_equal(got['accept'], launch[0]['accept']) and np.array_equal(got['gamma'], launch[0]['gamma'])
    assert np.array_equal(dx_in, dx_l) and np.array_equal(P_in, P_l)
    assert rel(dx_in, ref['dx']) < 1e-6 and rel(P_in, ref['P_new']) < 1e-6


@pytest.mark.parametrize('F', [509, 510, 511, 2000])
def test_track_counts_around_the_co_residency_limit(built, F):
    """Up to 2 (CUs - 1) = 510 tracks the front end is ONE co-resident launch (k_front); beyond that the update takes the forked
    form (k_feature, k_gram_pair, k_assemble_A with chol(P) on a side stream).  Same results either way."""
    upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    try:
        win = synth.make_window(N=30, F=F, seed=F, flags=synth.Flags(use_larvio=1), outlier_frac=0.05)
        fused = upd.update_features(win)
        upd.set_fused_front(False)
        forked = upd.update_features(win)
        upd.set_fused_front(True)
        assert np.array_equal(fused['accept'], forked['accept'])
        assert rel(fused['gamma'], forked['gamma']) < 1e-12
        assert rel(fused['dx'], forked['dx']) < 1e-10 and rel(fused['P_new'], forked['P_new']) < 1e-12
    finally:
        upd.close()


def test_malformed_and_degenerate_inputs_never_crash_or_poison_the_handle(built):
    """scripts/gpu_fuzz_inputs.py in a child process (a crash of the library would take the test runner with it): malformed index
    arrays come back as status codes; empty / one-observation / one-clone windows, NaN / Inf in a track, a zero prior are 'no
    update' or the update without the offending track; a prior that is not PSD is flagged in stats[6]; a prior or a noise
    value that makes M = s2 I + L^T A L lose positive definiteness (scale 1e30, s2 = 0, NaN) is ORCVIO_ERR_NOT_SPD with P and x
    left alone ON THE DEVICE (the resident covariance and its factor survive, cov_commit refuses); after every case the same
    handle reproduces the oracle on a well-formed window."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'gpu_fuzz_inputs.py')], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rep = json.loads(out.stdout[out.stdout.index('{\n'):])
    assert len(rep['cases']) >= 30
    assert rep['problems'] == []

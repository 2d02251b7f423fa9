"""One filter frame in one call (orcvio_msckf_io_step_frame: propagation, augmentation, the update, the prune update and the
marginalisation of reference src/orcvio.cpp:567-594 enqueued at once on the resident covariance) against the separate calls it
replaces -- bit for bit -- and against the oracle; its refusals; and the C++ stream harness (tests/cpp/stream_bench.cpp), which
replays the same frames through the C-ABI without Python."""
import json
import os
import subprocess

import numpy as np
import pytest

from orcvio_amd import capi, synth
from helpers import rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEG, IDP, NSLAM = 22, 1, 12


def _handle(debug_hooks=False):
    u = capi.MsckfUpdater(device=0, max_clones=24, max_features=256, max_observations=4096, debug_hooks=debug_hooks)
    u.set_extra_states(IDP * NSLAM)
    u.set_ekf_rows_mode(True)
    return u


def _frame_by_calls(u, fr, prune_poses=None):
    """The separate calls of the round-5 ABI for one frame; returns (dx1, gamma1, accept1, dx2 | None)."""
    w = fr['w']
    u.cov_propagate(fr['Phi'], fr['Q'])
    u.cov_augment()
    io = u.io_begin(w.flags, w.N, w.F, int(w.obs_ptr[-1]), with_P=False)
    u.io_fill(io, w, with_P=False)
    u.make_slam_call(IDP, fr['slam'])()
    u.io_update(want_P=False, commit=True)
    out = [io['dx'].copy(), io['gamma'].copy(), io['accept'].copy(), None]
    if fr['prune'] is not None:
        p = fr['prune'] if prune_poses is None else prune_poses(fr['prune'], out[0])
        io = u.io_begin(p.flags, p.N, p.F, int(p.obs_ptr[-1]), with_P=False)
        u.io_fill(io, p, with_P=False)
        u.io_update(want_P=False, commit=True)
        out[3] = io['dx'].copy()
    if fr['remove']:
        u.cov_remove_clones(LEG, fr['remove'])
    return out


@pytest.mark.parametrize('which', ['euroc', 'kitti'])
def test_step_frame_equals_the_separate_calls_bit_for_bit(built, which):
    fl = synth.Flags(use_larvio=1) if which == 'euroc' else synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=1.0, discard_large_update=1)
    frames, P0 = synth.make_stream(fl, sigma_px=None if which == 'euroc' else 0.008)
    a, b = _handle(), _handle()
    try:
        a.cov_set(P0); b.cov_set(P0)
        for it in range(24):
            fr = frames[it % len(frames)]
            ref = _frame_by_calls(a, fr)
            got = b.io_step_frame(fr['w'], fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], False, fr['remove'])
            assert got['repaired'] == 0 and got['status_first'] == 0 and got['status_prune'] == 0
            assert np.array_equal(got['dx'], ref[0]), it
            assert np.array_equal(got['gamma'], ref[1], equal_nan=True) and np.array_equal(got['accept'], ref[2])
            assert got['stats'][3] == 1 and got['stats'][2] == int(ref[2].sum())
            if fr['prune'] is not None:
                assert np.array_equal(got['prune_dx'], ref[3]), it
            else:
                assert got['prune_dx'] is None
            Pa, Pb = a.cov_get(), b.cov_get()
            assert got['n_after'] == Pb.shape[0] == Pa.shape[0]
            assert np.array_equal(Pa, Pb), it
    finally:
        a.close(); b.close()


def test_cpp_stream_harness_both_modes_agree(built, tmp_path):
    """tests/cpp/stream_bench.cpp: the stream through the C-ABI from C++ -- separate calls against one call per frame, same dx and
    same final covariance (hashes), and a sane figure."""
    fl = synth.Flags(use_larvio=1)
    frames, P0 = synth.make_stream(fl)
    path = str(tmp_path / 'config1.bin')
    synth.write_stream(path, frames, P0, fl, IDP)
    exe = str(tmp_path / 'stream_bench')
    lib = os.path.join(ROOT, 'orcvio_amd', 'lib')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-DORCVIO_HAVE_STEP_FRAME', '-o', exe, os.path.join(ROOT, 'tests', 'cpp', 'stream_bench.cpp'),
                           '-L', lib, '-lorcvio_msckf', f'-Wl,-rpath,{lib}'])
    outs = {}
    for mode in ('calls', 'step'):
        r = subprocess.run([exe, '--stream', path, '--mode', mode, '--frames', '96', '--warmup', '16'], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    assert outs['calls']['dx_hash'] == outs['step']['dx_hash']
    assert outs['calls']['P_hash'] == outs['step']['P_hash']
    assert outs['step']['frames_per_s'] > outs['calls']['frames_per_s']
    assert outs['step']['front_fallbacks'] == 0


def test_step_frame_with_the_state_increment_on_the_device(built):
    """prune_apply_dx: the second update's window = the first update's poses incremented by its dx (incrementState_IMUCam,
    reference src/orcvio.cpp:4468-4567) on the device -- against the separate calls with the host's increment between them."""
    import dataclasses
    for fl in (synth.Flags(use_larvio=1), synth.Flags(use_larvio=0, use_left_perturbation=0), synth.Flags(use_larvio=0, use_left_perturbation=1)):
        frames, P0 = synth.make_stream(fl)
        a, b = _handle(), _handle()

        def incremented(p, dx):
            return capi.increment_window(p, dx)[0]
        try:
            a.cov_set(P0); b.cov_set(P0)
            for it in range(8):
                fr = frames[it % len(frames)]
                ref = _frame_by_calls(a, fr, prune_poses=incremented)
                got = b.io_step_frame(fr['w'], fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], True, fr['remove'])
                assert rel(got['dx'], ref[0]) < 1e-9   # (not bit for bit: the device's sin / cos against the host's, carried by the covariance from the first prune update on)
                if fr['prune'] is not None:
                    assert rel(got['prune_dx'], ref[3]) < 1e-9, (it, rel(got['prune_dx'], ref[3]))
                assert rel(b.cov_get(), a.cov_get()) < 1e-10
        finally:
            a.close(); b.close()


def test_step_frame_validation_leaves_nothing_done(built):
    import dataclasses
    fl = synth.Flags(use_larvio=1)
    frames, P0 = synth.make_stream(fl)
    u = _handle()
    try:
        u.cov_set(P0)
        fr0, fr = frames[0], frames[1]
        got = u.io_step_frame(fr0['w'], fr0['Phi'], fr0['Q'], True, fr0['slam'], IDP, None, False, [])
        assert got['rc'] == 0 and got['stats'][3] == 1 and got['n_after'] == P0.shape[0] + 6
        P1 = u.cov_get()
        bad = synth.subset_tracks(fr['w'], [0, 1], min_obs=2)
        bad.obs_clone[0] = 99
        with pytest.raises(capi.MsckfError) as e:   # the prune tracks are checked before anything is enqueued
            u.io_step_frame(fr['w'], fr['Phi'], fr['Q'], True, fr['slam'], IDP, bad, False, fr['remove'])
        assert e.value.code == 1
        assert np.array_equal(u.cov_get(), P1)
        w_bad = dataclasses.replace(fr['w'], obs_clone=fr['w'].obs_clone.copy())
        w_bad.obs_clone[3] = 77
        with pytest.raises(capi.MsckfError):   # ... and so are the first update's
            u.io_step_frame(w_bad, fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], False, fr['remove'])
        assert np.array_equal(u.cov_get(), P1)
        with pytest.raises(capi.MsckfError):   # the window of io_begin must be the augmented one
            u.io_step_frame(fr['w'], fr['Phi'], fr['Q'], False, fr['slam'], IDP, None, False, [])
        with pytest.raises(capi.MsckfError):   # descending removal list
            u.io_step_frame(fr['w'], fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], False, [1, 0])
        assert np.array_equal(u.cov_get(), P1)
        # ... and the frame goes through afterwards
        got = u.io_step_frame(fr['w'], fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], False, fr['remove'])
        assert got['rc'] == 0 and got['stats'][3] == 1 and got['n_after'] == P0.shape[0]
    finally:
        u.close()


def test_step_frame_refusal_of_the_first_update_refuses_the_second(built):
    """A prior 1e30 times the noise in scale: sigma^2 is lost beside L^T A L and M is not positive definite in double -- the device
    refuses the first update (ORCVIO_ERR_NOT_SPD, nothing applied) and the second with it; the covariance bookkeeping of the frame
    stands: the result equals propagate + augment + remove alone."""
    fl = synth.Flags(use_larvio=1)
    frames, P0 = synth.make_stream(fl)
    a, b = _handle(), _handle()
    try:
        a.cov_set(P0 * 1e30); b.cov_set(P0 * 1e30)
        for k in (0, 1):
            fr = frames[k]
            got = a.io_step_frame(fr['w'], fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], False, fr['remove'], raise_on_refusal=False)
            assert got['rc'] == 6 and got['status_first'] == 6, got
            assert got['stats'][3] == 0
            if fr['prune'] is not None:
                assert got['status_prune'] == 6 and got['prune_stats'][3] == 0
            b.cov_propagate(fr['Phi'], fr['Q']); b.cov_augment()
            if fr['remove']:
                b.cov_remove_clones(LEG, fr['remove'])
            Pb = b.cov_get()
            assert got['n_after'] == Pb.shape[0]
            assert np.array_equal(a.cov_get(), Pb)
        # the handle goes on with a sane covariance
        a.cov_set(P0)
        fr0 = frames[0]
        got = a.io_step_frame(fr0['w'], fr0['Phi'], fr0['Q'], True, fr0['slam'], IDP, None, False, [])
        assert got['rc'] == 0 and got['stats'][3] == 1
    finally:
        a.close(); b.close()


@pytest.mark.parametrize('spin', ['ORCVIO_LA_SPIN', 'ORCVIO_FRONT_SPIN'])
def test_step_frame_repairs_a_lost_hand_off(built, monkeypatch, spin):
    """ORCVIO_LA_SPIN = 0: every wait of the look-ahead factorisation on another workgroup gives up at once -- each update of each
    frame flags itself and refuses its commit on the device (and the second update with the first).  The call takes the
    marginalisation back, runs the lost updates again in separate launches (the one-workgroup factorisation, the outcome checked by
    the host) and marginalises: the same filter as the separate calls on an undisturbed handle.  ORCVIO_FRONT_SPIN = 0: the waits of
    k_front's device-wide counter and the two words that join the in-state rows' side stream (k_wait_word, k_gemm_asmA_w) give up instead."""
    fl = synth.Flags(use_larvio=1)
    frames, P0 = synth.make_stream(fl)
    a = _handle()
    monkeypatch.setenv(spin, '0')
    b = _handle()
    monkeypatch.delenv(spin)
    try:
        a.cov_set(P0); b.cov_set(P0)
        repaired = 0
        for it in range(10):
            fr = frames[it % len(frames)]
            ref = _frame_by_calls(a, fr)
            got = b.io_step_frame(fr['w'], fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], False, fr['remove'])
            assert got['rc'] == 0 and got['stats'][3] == 1
            repaired += got['repaired']
            assert got['repaired'] == (2 if fr['prune'] is not None else 1), (it, got['repaired'])
            assert rel(got['dx'], ref[0]) < 1e-9 and np.array_equal(got['accept'], ref[2])
            if fr['prune'] is not None:
                assert rel(got['prune_dx'], ref[3]) < 1e-9
            Pa, Pb = a.cov_get(), b.cov_get()
            assert Pa.shape == Pb.shape and rel(Pb, Pa) < 1e-10
        c = b.counters()
        assert c['step_frames'] == 10 and c['step_repairs'] == repaired and c['front_fallbacks'] >= 10
    finally:
        a.close(); b.close()


@pytest.mark.parametrize('off', ['ORCVIO_STEP_FUSED', 'ORCVIO_FINISH_PUB', 'ORCVIO_EKF_ONE_LAUNCH', 'ORCVIO_STEP_EKF_SIDE', 'all', 'ORCVIO_THIN_UPDATE'])
def test_the_folded_launches_equal_the_separate_ones_bit_for_bit(built, monkeypatch, off):
    """k_frame_head / k_cov_remove_fac (ORCVIO_STEP_FUSED), k_finish_pub (ORCVIO_FINISH_PUB) and k_ekf_evalgate (ORCVIO_EKF_ONE_LAUNCH) each
    switched off in the diagnostics build -- the frame then runs the separate launches and copies of the round-5 calls, enqueued at
    once -- against the default: the same dx, the same covariance, bit for bit.  (The direct form of a thin stack rides on
    k_finish_pub's publication, so both sides run without it where that launch is switched off.)  ORCVIO_STEP_EKF_SIDE = 0: the in-state
    features' rows in front of k_front on the update's own stream instead of beside it on the side stream.  ORCVIO_THIN_UPDATE = 0 alone: the
    prune update through the general square-root path instead of the direct form -- another algorithm for the same update: equal to
    rounding."""
    fl = synth.Flags(use_larvio=1)
    frames, P0 = synth.make_stream(fl)
    thin_cmp = off == 'ORCVIO_THIN_UPDATE'
    if off in ('ORCVIO_FINISH_PUB', 'all'):
        monkeypatch.setenv('ORCVIO_THIN_UPDATE', '0')
    a = _handle(debug_hooks=True)
    for name in (['ORCVIO_STEP_FUSED', 'ORCVIO_FINISH_PUB', 'ORCVIO_EKF_ONE_LAUNCH', 'ORCVIO_STEP_EKF_SIDE'] if off == 'all' else [off]):
        monkeypatch.setenv(name, '0')
    b = _handle(debug_hooks=True)
    same = (lambda x, y: rel(x, y) < 1e-9) if thin_cmp else (lambda x, y: np.array_equal(x, y, equal_nan=True))
    try:
        a.cov_set(P0); b.cov_set(P0)
        for it in range(16):
            fr = frames[it % len(frames)]
            ra = a.io_step_frame(fr['w'], fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], False, fr['remove'])
            rb = b.io_step_frame(fr['w'], fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], False, fr['remove'])
            assert same(ra['dx'], rb['dx']) and np.array_equal(ra['accept'], rb['accept']), it
            if fr['prune'] is not None:
                assert same(ra['prune_dx'], rb['prune_dx']), it
            assert same(a.cov_get(), b.cov_get()), it
    finally:
        a.close(); b.close()


def test_step_frame_without_a_first_update_and_without_propagation(built):
    """Frames the reference meets at start-up and on quiet images: no lost feature and no in-state feature (removeLostFeatures returns
    before any arithmetic) -- the covariance bookkeeping and the prune update are the whole frame; and a frame without propagation
    (Phi == NULL: augmentation alone, the resident factor's rows copied along).  Against the separate calls, bit for bit."""
    import dataclasses
    fl = synth.Flags(use_larvio=1)
    frames, P0 = synth.make_stream(fl)
    a, b = _handle(), _handle()
    try:
        a.cov_set(P0); b.cov_set(P0)
        fr0, fr1 = frames[0], frames[1]
        empty = lambda w: dataclasses.replace(w, p_w=w.p_w[:0].copy(), obs_ptr=np.zeros(1, np.int32), obs_clone=w.obs_clone[:0].copy(),
                                              obs_z=w.obs_z[:0].copy(), obs_zvel=w.obs_zvel[:0].copy())
        # frame 0: nothing to update at all
        a.cov_propagate(fr0['Phi'], fr0['Q']); a.cov_augment()
        got = b.io_step_frame(empty(fr0['w']), fr0['Phi'], fr0['Q'], True, None, IDP, None, False, [])
        assert got['rc'] == 0 and got['stats'][3] == 0 and not np.any(got['dx']) and got['n_after'] == P0.shape[0] + 6
        assert np.array_equal(a.cov_get(), b.cov_get())
        # frame 1: no first update, but the prune update and the marginalisation
        a.cov_propagate(fr1['Phi'], fr1['Q']); a.cov_augment()
        p = fr1['prune']
        io = a.io_begin(p.flags, p.N, p.F, int(p.obs_ptr[-1]), with_P=False)
        a.io_fill(io, p, with_P=False)
        a.io_update(want_P=False, commit=True)
        ref_dx = io['dx'].copy()
        a.cov_remove_clones(LEG, fr1['remove'])
        got = b.io_step_frame(empty(fr1['w']), fr1['Phi'], fr1['Q'], True, None, IDP, p, True, fr1['remove'])
        assert got['rc'] == 0 and got['stats'][3] == 0 and got['prune_stats'][3] == 1
        assert np.array_equal(got['prune_dx'], ref_dx)
        assert np.array_equal(a.cov_get(), b.cov_get())
        # frame 2: augmentation without propagation, behind an update that left its factor resident (frame 0's window, 19 clones)
        fr2 = frames[2]
        for u in (a, b):   # (an update that commits a square-root factor: the tracks of frame 2 on the 18-clone window would not fit; use the augment-only call first)
            pass
        a.cov_augment()
        io = a.io_begin(fr2['w'].flags, fr2['w'].N, fr2['w'].F, int(fr2['w'].obs_ptr[-1]), with_P=False)
        a.io_fill(io, fr2['w'], with_P=False)
        a.make_slam_call(IDP, fr2['slam'])()
        a.io_update(want_P=False, commit=True)
        ref_dx = io['dx'].copy()
        got = b.io_step_frame(fr2['w'], None, None, True, fr2['slam'], IDP, None, False, [])
        assert got['rc'] == 0 and np.array_equal(got['dx'], ref_dx)
        assert np.array_equal(a.cov_get(), b.cov_get())
        # ... and once more without propagation, now WITH the resident factor of that update: its rows are copied along
        fr3 = frames[3]
        a.cov_augment()
        io = a.io_begin(fr3['w'].flags, fr3['w'].N, fr3['w'].F, int(fr3['w'].obs_ptr[-1]), with_P=False)
        a.io_fill(io, fr3['w'], with_P=False)
        a.make_slam_call(IDP, fr3['slam'])()
        a.io_update(want_P=False, commit=True)
        ref_dx = io['dx'].copy()
        got = b.io_step_frame(fr3['w'], None, None, True, fr3['slam'], IDP, None, False, [])
        assert got['rc'] == 0 and np.array_equal(got['dx'], ref_dx)
        assert np.array_equal(a.cov_get(), b.cov_get())
    finally:
        a.close(); b.close()


def test_step_frame_with_imu_intrinsics_in_the_state(built):
    """leg_dim 46 (calib_imu_instrinsic, reference src/orcvio.cpp:196-199): k_frame_head's other instantiation -- against the
    separate calls, bit for bit."""
    fl = synth.Flags(use_larvio=1, leg_dim=46)
    frames, P0 = synth.make_stream(fl, leg=46)
    a, b = _handle(), _handle()
    try:
        a.cov_set(P0); b.cov_set(P0)
        for it in range(6):
            fr = frames[it % len(frames)]
            w = fr['w']
            a.cov_propagate(fr['Phi'], fr['Q']); a.cov_augment()
            io = a.io_begin(w.flags, w.N, w.F, int(w.obs_ptr[-1]), with_P=False)
            a.io_fill(io, w, with_P=False)
            a.make_slam_call(IDP, fr['slam'])()
            a.io_update(want_P=False, commit=True)
            ref = io['dx'].copy()
            ref2 = None
            if fr['prune'] is not None:
                p = fr['prune']
                io = a.io_begin(p.flags, p.N, p.F, int(p.obs_ptr[-1]), with_P=False)
                a.io_fill(io, p, with_P=False)
                a.io_update(want_P=False, commit=True)
                ref2 = io['dx'].copy()
            if fr['remove']:
                a.cov_remove_clones(46, fr['remove'])
            got = b.io_step_frame(w, fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], False, fr['remove'])
            assert got['rc'] == 0 and np.array_equal(got['dx'], ref), it
            if ref2 is not None:
                assert np.array_equal(got['prune_dx'], ref2), it
            assert np.array_equal(a.cov_get(), b.cov_get()), it
    finally:
        a.close(); b.close()


def test_a_slice_of_the_randomised_soak(built):
    """scripts/gpu_soak_step.py for a few seconds: random flags / leg_dim / window sizes / track counts (none .. 250, sometimes all
    outliers), prune updates in the direct form and through the square-root path, frames without propagation or augmentation,
    the state increment on the device -- every dx, accept mask and covariance against the host chain (C oracle + numpy mirrors)."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'gpu_soak_step.py'), '8', '4242'], capture_output=True, text=True, timeout=300)
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and rec['failures'] == 0 and rec['frames'] > 50, rec
    assert rec['worst']['dx'] < 1e-6 and rec['worst']['P'] < 1e-6

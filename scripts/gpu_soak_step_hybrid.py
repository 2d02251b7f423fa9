"""Randomised soak of orcvio_msckf_io_step_frame on HYBRID-filter frames (lost tracks + in-state features, the stream of
synth.make_stream with random seeds, flag sets and pixel noise): the frame in one call -- the in-state features' rows on the side
stream beside k_front, joined by polled words -- against the separate calls of the round-5 ABI on a second handle, BIT FOR BIT (dx, gamma,
accept masks, the prune update's dx, the covariance after every frame).  The separate calls are what the other soaks hold against the
oracle (gpu_soak_hybrid*.py, gpu_soak_loop.py).  usage: python scripts/gpu_soak_step_hybrid.py [seconds] [first_seed]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import capi, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LEG, IDP, NSLAM = 22, 1, 12


def handle():
    u = capi.MsckfUpdater(device=0, max_clones=24, max_features=256, max_observations=4096)
    u.set_extra_states(IDP * NSLAM)
    u.set_ekf_rows_mode(True)
    return u


def by_calls(u, fr, apply_dx):
    w = fr['w']
    u.cov_propagate(fr['Phi'], fr['Q'])
    u.cov_augment()
    io = u.io_begin(w.flags, w.N, w.F, int(w.obs_ptr[-1]), with_P=False)
    u.io_fill(io, w, with_P=False)
    u.make_slam_call(IDP, fr['slam'])()
    u.io_update(want_P=False, commit=True)
    out = [io['dx'].copy(), io['gamma'].copy(), io['accept'].copy(), None]
    if fr['prune'] is not None:
        p = fr['prune']
        if apply_dx:
            p = capi.increment_window(p, out[0])
        io = u.io_begin(p.flags, p.N, p.F, int(p.obs_ptr[-1]), with_P=False)
        u.io_fill(io, p, with_P=False)
        u.io_update(want_P=False, commit=True)
        out[3] = io['dx'].copy()
    if fr['remove']:
        u.cov_remove_clones(LEG, fr['remove'])
    return out


a, b = handle(), handle()
fails, n_frames, n_streams, n_refused = [], 0, 0, 0
t_end = time.time() + budget
seed = seed0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    euroc = bool(rng.integers(0, 2))
    fl = synth.Flags(use_larvio=1) if euroc else synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=1.0, discard_large_update=1)
    frames, P0 = synth.make_stream(fl, sigma_px=None if euroc else 0.008, seed=seed, cycle=int(rng.choice([4, 6, 8])))
    a.cov_set(P0); b.cov_set(P0)
    n_streams += 1
    try:
        for it in range(2 * len(frames)):
            fr = frames[it % len(frames)]
            try:
                ref = by_calls(a, fr, False)
            except capi.MsckfError:
                n_refused += 1
                break
            got = b.io_step_frame(fr['w'], fr['Phi'], fr['Q'], True, fr['slam'], IDP, fr['prune'], False, fr['remove'], raise_on_refusal=False)
            n_frames += 1
            ok = got['repaired'] == 0 and got['status_first'] == 0 and np.array_equal(got['dx'], ref[0]) and \
                np.array_equal(got['gamma'], ref[1], equal_nan=True) and np.array_equal(got['accept'], ref[2])
            if fr['prune'] is not None:
                ok = ok and got['status_prune'] == 0 and np.array_equal(got['prune_dx'], ref[3])
            ok = ok and np.array_equal(a.cov_get(), b.cov_get())
            if not ok:
                fails.append(dict(seed=seed, frame=it, repaired=int(got['repaired']), status=[int(got['status_first']), int(got['status_prune'])]))
                break
    except Exception as e:   # noqa: BLE001
        fails.append(dict(seed=seed, error=repr(e)))
    seed += 1
print(json.dumps(dict(what='io_step_frame on hybrid frames (in-state rows on the side stream) against the separate calls, bit for bit', seconds=budget,
                      first_seed=seed0, streams=n_streams, frames=n_frames, refused_streams=n_refused, counters=b.counters(), failures=fails)))
a.close(); b.close()

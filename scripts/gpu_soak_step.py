"""Randomised soak of orcvio_msckf_io_step_frame (one filter frame in one call): random runs of frames on the resident covariance,
each with random flags / leg_dim / window size, a random number of lost tracks (0 .. 250, sometimes all outliers), a prune update of
0 .. 40 rows on random leaving clones (so that both the direct form of a thin stack and the square-root path take it), propagation
and augmentation sometimes left out, 0 .. 2 clones marginalised -- against the SAME loop on the host: numpy mirrors for the covariance
bookkeeping (oracle/mirror_cov.py), the C oracle for both updates (on the window incremented by the first update's dx when
prune_apply_dx is drawn).  Every dx, the accept masks and the covariance at the end of every frame, at 1e-6.
usage: python scripts/gpu_soak_step.py [seconds] [first_seed]"""
import sys, os, json, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from orcvio_amd import capi, synth
from oracle import oracle
from oracle import mirror_cov as mc
from helpers import rel

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=512, max_observations=16384)
fails, n_runs, n_frames, n_updates, n_thin, worst = [], 0, 0, 0, 0, dict(dx=0.0, P=0.0)
t_end = time.time() + budget
seed = seed0


def check(tag, par, got_dx, ref_dx, accept_equal=True):
    global n_updates
    n_updates += 1
    e = rel(got_dx, ref_dx) if np.linalg.norm(ref_dx) > 0 else float(np.linalg.norm(got_dx))
    worst['dx'] = max(worst['dx'], e)
    if not (accept_equal and e < 1e-6):
        fails.append(dict(par, step=tag, accept_equal=bool(accept_equal), e_dx=e))
        return False
    return True


while time.time() < t_end:
    rng = np.random.default_rng(770000 + seed)
    leg = int(rng.choice([22, 22, 46]))
    variant = int(rng.integers(0, 3))
    flags = synth.Flags(use_larvio=int(variant == 0), use_left_perturbation=int(variant == 1), if_fej=int(rng.integers(0, 2)),
                        estimate_td=int(rng.integers(0, 2)), leg_dim=leg, noise_feature=float(rng.choice([0.008, 0.05])),
                        discard_large_update=int(rng.integers(0, 2)))
    N = int(rng.integers(3, 22))
    frames = int(rng.integers(3, 8))
    par = dict(seed=seed, leg=leg, variant=variant, N0=N, frames=frames)
    try:
        P = synth.make_window(N=N, F=1, seed=seed, flags=flags).P.copy()
        upd.cov_set(P)
        good = True
        for fr in range(frames):
            par['frame'] = fr
            prop = rng.integers(0, 5) > 0
            aug = N < 24 and rng.integers(0, 6) > 0
            Phi = Q = None
            if prop:
                Phi = np.eye(leg) + 0.01 * rng.standard_normal((leg, leg))
                G = rng.standard_normal((leg, 12))
                Q = 1e-6 * G @ G.T
                P = mc.propagate(P, Phi, Q)
            if aug:
                P = mc.augment(P)
            N = (P.shape[0] - leg) // 6
            F = int(rng.choice([0, rng.integers(1, 30), rng.integers(30, 250)], p=[0.1, 0.45, 0.45]))
            out_frac = float(rng.choice([0.0, 0.2, 1.0], p=[0.5, 0.4, 0.1]))
            w = synth.make_window(N=N, F=max(F, 1), seed=1000 * seed + fr, flags=flags, track_len=(min(3, N), min(N, 8)), outlier_frac=out_frac, sigma_px=0.008)
            if F == 0:
                w = dataclasses.replace(w, p_w=w.p_w[:0].copy(), obs_ptr=np.zeros(1, np.int32), obs_clone=w.obs_clone[:0].copy(),
                                        obs_z=w.obs_z[:0].copy(), obs_zvel=w.obs_zvel[:0].copy())
            w.P[:] = P
            ref = oracle.msckf_update(w, want_blocks=False, want_K=False) if F > 0 else None
            P1 = ref['P_new'] if ref is not None else P
            dx1 = ref['dx'] if ref is not None else np.zeros(P.shape[0])
            # the prune update: tracks of another draw, restricted to two or three leaving clones
            prune, ref2, apply_dx = None, None, bool(rng.integers(0, 2))
            if N >= 4 and rng.integers(0, 3) > 0:
                leave = sorted(rng.choice(N - 1, int(rng.integers(2, 4)), replace=False).tolist())
                wp = synth.make_window(N=N, F=int(rng.integers(1, 40)), seed=2000 * seed + fr, flags=flags, track_len=(min(3, N), min(N, 8)), sigma_px=0.008)
                wp = dataclasses.replace(wp, R_b2w=w.R_b2w, t_b_w=w.t_b_w, t_fej=w.t_fej, R_b2c=w.R_b2c, t_c_b=w.t_c_b)
                sub = synth.subset_tracks(wp, leave, min_obs=2)
                if int(sub.obs_ptr[-1]) > 0:
                    prune = sub
                    host_win = sub
                    if apply_dx and ref is not None:
                        host_win, applied = capi.increment_window(sub, dx1)
                    host_win = dataclasses.replace(host_win, P=P1)
                    ref2 = oracle.msckf_update(host_win, want_blocks=False, want_K=False)
                    rows = int(sum(max(2 * int(m) - 3, 0) for m in np.diff(sub.obs_ptr)))
                    n_thin += rows <= 16
            P2 = ref2['P_new'] if ref2 is not None else P1
            remove = []
            if N >= 3 and rng.integers(0, 2):
                remove = sorted(rng.choice(N, int(rng.integers(1, 3)), replace=False).tolist())
                P2 = mc.remove_clones(P2, leg, remove)
            got = upd.io_step_frame(w, Phi, Q, aug, None, 1, prune, apply_dx, remove)
            if ref is not None:
                good &= check('first', par, got['dx'], ref['dx'], np.array_equal(got['accept'], ref['accept']))
            if ref2 is not None:
                good &= check('prune', par, got['prune_dx'], ref2['dx'], np.array_equal(got['prune_accept'], ref2['accept']))
            eP = rel(upd.cov_get(), P2)
            worst['P'] = max(worst['P'], eP)
            n_frames += 1
            if eP > 1e-6 or got['repaired'] != 0:
                fails.append(dict(par, step='covariance at the end of the frame', e_P=eP, repaired=got['repaired']))
                good = False
            P = P2
            if not good:
                break
    except Exception as e:
        fails.append(dict(par, error=repr(e)[:300]))
    n_runs += 1
    seed += 1
print(json.dumps(dict(soak='io_step_frame', seconds=budget, first_seed=seed0, runs=n_runs, frames=n_frames, updates=n_updates,
                      prune_updates_in_the_direct_form=int(n_thin), worst=worst, failures=len(fails), first_failures=fails[:5])))
upd.close()
sys.exit(1 if fails else 0)

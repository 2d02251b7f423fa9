"""Randomised soak of the filter LOOP on the device-resident covariance (the part with state across calls: the resident P, its
square-root factor kept / dropped / permuted, the captured graphs): random runs of frames, each
propagate -> augment -> [prefactor] -> feature update -> commit -> [prune update on the two oldest clones -> commit] ->
[object update -> commit] -> [marginalise random clones] (one frame in four: feature update and object update in ONE call,
orcvio_msckf_io_update_frame), P never sent after the first frame; every dx and the covariance at the
end of every frame against the same loop on the host (C oracle for the feature updates, numpy mirrors for the rest).
usage: python scripts/gpu_soak_loop.py [seconds] [first_seed]"""
import sys, os, json, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from orcvio_amd import capi, synth
from oracle import oracle
from oracle import mirror_cov as mc
from helpers import rel, subset_window, objects_update_reference

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
fails, n_runs, n_frames, n_updates, worst = [], 0, 0, 0, dict(dx=0.0, P=0.0)
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
    rng = np.random.default_rng(660000 + seed)
    leg = int(rng.choice([22, 22, 46]))
    variant = int(rng.integers(0, 3))
    flags = synth.Flags(use_larvio=int(variant == 0), use_left_perturbation=int(variant == 1), if_fej=int(rng.integers(0, 2)),
                        estimate_td=int(rng.integers(0, 2)), leg_dim=leg, noise_feature=float(rng.choice([0.008, 0.05])))
    N = int(rng.integers(2, 10)) if not os.environ.get('ORCVIO_FRAME_CHAIN') else int(rng.integers(9, 16))   # (the chained object solve takes windows from six block steps)
    cap = int(rng.integers(N + 1, 25))
    frames = int(rng.integers(3, 9))
    par = dict(seed=seed, leg=leg, variant=variant, N0=N, cap=cap, frames=frames)
    try:
        P = synth.make_window(N=N, F=1, seed=seed, flags=flags).P.copy()
        upd.cov_set(P)
        good = True
        for fr in range(frames):
            par['frame'] = fr
            Phi = np.eye(leg) + 0.01 * rng.standard_normal((leg, leg))
            G = rng.standard_normal((leg, 12))
            Q = 1e-6 * G @ G.T
            upd.cov_propagate(Phi, Q); P = mc.propagate(P, Phi, Q)
            upd.cov_augment(); P = mc.augment(P)
            N = (P.shape[0] - leg) // 6
            if rng.integers(0, 2):
                upd.cov_prefactor()
            # 1: the lost features (sometimes every track is an outlier: no update, the prior and its factor stay)
            F = int(rng.choice([rng.integers(1, 30), rng.integers(30, 200)]))
            out_frac = float(rng.choice([0.0, 0.2, 1.0], p=[0.5, 0.4, 0.1]))
            w = synth.make_window(N=N, F=F, seed=1000 * seed + fr, flags=flags, track_len=(min(3, N), N), outlier_frac=out_frac, sigma_px=0.008)
            w.P[:] = P
            ref = oracle.msckf_update(w, want_blocks=False, want_K=False)
            one_call = N >= 4 and rng.integers(0, 4 if not os.environ.get('ORCVIO_FRAME_CHAIN') else 2) == 0   # features + objects of the frame in ONE call (orcvio_msckf_io_update_frame; every other frame when the chained form is being soaked)
            if one_call:
                objs = synth.make_objects(w, n_objects=int(rng.integers(1, 4)), seed=seed + fr, sigma_kp=float(rng.choice([0.004, 0.1])))
                ol, nb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                ref3 = objects_update_reference(dataclasses.replace(w, P=ref['P_new']), objs, ref['P_new'], ol, nb, flags.use_left_perturbation, full_nullspace=True)
                gf, go = upd.update_frame(w, flags, objs, w.R_b2c[0], w.t_c_b[0], ol, nb, flags.use_left_perturbation)
                good &= check('frame: features', par, gf['dx'], ref['dx'], np.array_equal(gf['accept'], ref['accept']))
                good &= check('frame: objects', par, go['dx'], ref3['dx'], go['accept'] == ref3['accept'])
                P = ref3['P_new']
            else:
                got = upd.update_features(w, resident_cov=True, want_P=False)
                upd.cov_commit()
                good &= check('features', par, got['dx'], ref['dx'], np.array_equal(got['accept'], ref['accept']))
                P = ref['P_new']
            # 2: the prune update on the two oldest clones
            if not one_call and N >= 3 and rng.integers(0, 2):
                sub = subset_window(w, [0, 1])
                both = np.diff(sub.obs_ptr) == 2
                if both.any():
                    keep = np.repeat(both, np.diff(sub.obs_ptr))
                    ptr = np.concatenate([[0], np.cumsum(np.where(both, 2, 0))]).astype(np.int32)
                    sub = dataclasses.replace(sub, obs_ptr=ptr, obs_clone=sub.obs_clone[keep].copy(), obs_z=sub.obs_z[keep].copy(),
                                              obs_zvel=sub.obs_zvel[keep].copy(), P=P)
                    ref2 = oracle.msckf_update(sub, want_blocks=False, want_K=False)
                    g2 = upd.update_features(sub, resident_cov=True, want_P=False)
                    upd.cov_commit()
                    good &= check('prune', par, g2['dx'], ref2['dx'], np.array_equal(g2['accept'], ref2['accept']))
                    P = ref2['P_new']
            # 3: objects
            if not one_call and N >= 4 and rng.integers(0, 3) == 0:
                wo = dataclasses.replace(w, P=P)
                objs = synth.make_objects(wo, n_objects=int(rng.integers(1, 4)), seed=seed + fr, sigma_kp=float(rng.choice([0.004, 0.1])))
                ol, nb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
                ref3 = objects_update_reference(wo, objs, P, ol, nb, flags.use_left_perturbation, full_nullspace=True)
                g3 = upd.update_object_tracks(flags, N, objs, None, wo.R_b2c[0], wo.t_c_b[0], ol, nb, flags.use_left_perturbation)
                upd.cov_commit()
                good &= check('objects', par, g3['dx'], ref3['dx'], g3['accept'] == ref3['accept'])
                P = ref3['P_new']
            # 4: marginalisation
            if N > cap or (N >= 3 and rng.integers(0, 3) == 0):
                k = int(rng.integers(1, min(3, N - 1) + 1))
                ix = sorted(rng.choice(N, k, replace=False).tolist())
                upd.cov_remove_clones(leg, ix); P = mc.remove_clones(P, leg, ix)
            eP = rel(upd.cov_get(), P)
            worst['P'] = max(worst['P'], eP)
            n_frames += 1
            if eP > 1e-6:
                fails.append(dict(par, step='covariance at the end of the frame', e_P=eP))
                good = False
            if not good:
                break
    except Exception as e:
        fails.append(dict(par, error=repr(e)[:300]))
    n_runs += 1
    seed += 1
print(json.dumps(dict(runs=n_runs, frames=n_frames, updates=n_updates, first_seed=seed0, frames_with_a_chained_object_solve=upd.counters().get('chained_frames'),
                      failures=fails, worst=worst), indent=1, default=str))

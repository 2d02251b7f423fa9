"""Randomised parity soak on the GPU box: many random windows (size, ragged and NON-contiguous tracks, Jacobian variant, FEJ,
td, leg_dim, noise, outliers, host / resident / prefactored prior) through the C-ABI against the C oracle.  Prints one JSON
summary; any failing seed is listed with its parameters.  usage: python scripts/gpu_soak.py [seconds] [first_seed]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from orcvio_amd import capi, synth
from oracle import oracle
from helpers import rel, scatter_tracks

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
upd = capi.MsckfUpdater(device=0, max_clones=40, max_features=2560, max_observations=81920)


fails, n_done, worst = [], 0, dict(dx=0.0, P=0.0, gamma=0.0)
t_end = time.time() + budget
seed = seed0
while time.time() < t_end:
    rng = np.random.default_rng(770000 + seed)
    N = int(rng.integers(2, 41))
    # (1 800 tracks and more: the two-launch tracks front end)
    F = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 720), rng.integers(1800, 2400)], p=[0.4, 0.43, 0.15, 0.02]))
    variant = int(rng.integers(0, 3))
    flags = synth.Flags(use_larvio=int(variant == 0), use_left_perturbation=int(variant == 1), if_fej=int(rng.integers(0, 2)),
                        estimate_td=int(rng.integers(0, 2)), leg_dim=int(rng.choice([22, 22, 46])),
                        noise_feature=float(rng.choice([0.008, 0.05, 1.0])))
    scattered = bool(rng.integers(0, 2))
    lo = int(rng.integers(1, min(N, 6) + 1))
    hi = int(rng.integers(lo, min(N, 32) + 1))
    mode = int(rng.integers(0, 3))   # 0 host P, 1 resident, 2 resident + prefactored
    inplace = bool(rng.integers(0, 2))   # orcvio_msckf_io_begin / _io_update (the caller writes into the arena) or the copying call
    if F >= 1800:
        hi = max(lo, min(hi, 12))   # (bounded oracle time: ~3 ms per track of 30 observations)
    par = dict(seed=seed, N=N, F=F, variant=variant, fej=flags.if_fej, td=flags.estimate_td, leg=flags.leg_dim, s=flags.noise_feature,
               scattered=scattered, lo=lo, hi=hi, mode=mode, inplace=inplace)
    try:
        if scattered:
            w = synth.make_window(N=N, F=F, seed=seed, track_len=None, flags=flags, outlier_frac=float(rng.choice([0.0, 0.3])), sigma_px=0.008)
            w = scatter_tracks(w, rng, lo, hi)
        else:
            w = synth.make_window(N=N, F=F, seed=seed, track_len=(lo, hi), flags=flags, outlier_frac=float(rng.choice([0.0, 0.3])), sigma_px=0.008)
        ref = oracle.msckf_update(w)
        if mode != 0:
            upd.cov_set(w.P)
            if mode == 2:
                upd.cov_prefactor()
        if inplace:
            io = upd.io_begin(w.flags, w.N, w.F, int(w.obs_ptr[-1]), with_P=mode == 0)
            upd.io_fill(io, w, with_P=mode == 0)
            upd.io_update(want_P=True, commit=False)
            got = dict(dx=io['dx'].copy(), gamma=io['gamma'].copy(), accept=io['accept'].copy(), P_new=io['P_out'].copy())
        elif mode == 0:
            got = upd.update_features(w)
        else:
            got = upd.update_features(w, resident_cov=True)
        ok = np.array_equal(got['accept'], ref['accept'])
        fin = np.isfinite(ref['gamma'])
        ok = ok and np.array_equal(np.isfinite(got['gamma']), fin)
        eg = rel(got['gamma'][fin], ref['gamma'][fin]) if fin.any() else 0.0
        ed = rel(got['dx'], ref['dx']) if np.linalg.norm(ref['dx']) > 0 else float(np.linalg.norm(got['dx']))
        eP = rel(got['P_new'], ref['P_new'])
        worst['dx'] = max(worst['dx'], ed); worst['P'] = max(worst['P'], eP); worst['gamma'] = max(worst['gamma'], eg)
        if not (ok and eg < 1e-9 and ed < 1e-6 and eP < 1e-6):
            fails.append(dict(par, accept_equal=bool(ok), e_gamma=eg, e_dx=ed, e_P=eP))
    except Exception as e:   # a status code from the library or a shape the wrapper refuses
        fails.append(dict(par, error=repr(e)[:300]))
    n_done += 1
    seed += 1
print(json.dumps(dict(windows=n_done, first_seed=seed0, failures=fails, worst=worst), indent=1))

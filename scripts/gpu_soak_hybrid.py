"""Randomised parity soak of the hybrid (MSCKF tracks + EKF-SLAM features) update on the GPU box against the literal restatement
oracle/mirror_hybrid.py: random window, number of SLAM features, parametrisation (inverse depth / 3-d), FEJ, td, rows handed
over or evaluated on the device, Schmidt nuisance anchors.  usage: python scripts/gpu_soak_hybrid.py [seconds] [first_seed]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from orcvio_amd import capi, synth
from oracle import mirror_hybrid as mh
from test_gpu_hybrid import run, rel

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
upd = capi.MsckfUpdater(device=0, max_clones=48, max_features=2048, max_observations=65536)   # n up to 22 + 180 + 90: beyond n = 224 the LDS-panel factorisation
fails, n_done, worst = [], 0, dict(dx=0.0, P=0.0, gamma=0.0)
t_end = time.time() + budget
seed = seed0
while time.time() < t_end:
    rng = np.random.default_rng(990000 + seed)
    N = int(rng.integers(4, 31))
    F = int(rng.choice([0, rng.integers(1, 30), rng.integers(30, 90)]))
    nf = int(rng.integers(1, 31))
    idp = int(rng.choice([1, 3]))
    on_device = bool(rng.integers(0, 2))
    variant = int(rng.integers(0, 3))   # Jacobians of the MSCKF rows: LARVIO, OrcVIO left, OrcVIO right (config/euroc.yaml:114-118, kitti_raw.yaml:143-148)
    fl = synth.Flags(use_larvio=int(variant == 0), use_left_perturbation=int(variant == 1), if_fej=int(rng.integers(0, 2)), estimate_td=int(rng.integers(0, 2)))
    par = dict(seed=seed, N=N, F=F, nf=nf, idp=idp, on_device=on_device, fej=fl.if_fej, td=fl.estimate_td)
    try:
        w0 = synth.make_window(N=N, F=F, seed=seed, track_len=(min(3, N), min(N, 9)), flags=fl)
        slam = synth.make_slam_features(w0, nf, seed=seed, outlier_frac=float(rng.choice([0.0, 0.25])))
        w = synth.with_extra_states(w0, idp * len(slam), seed=seed + 1)
        ref = mh.hybrid_update(w, slam, idp)
        got = run(upd, w, slam, idp, on_device)
        ok = np.array_equal(got['ekf_accept'], ref['ekf_accept']) and np.array_equal(got['accept'], ref['accept'])
        eg = rel(got['ekf_gamma'], ref['ekf_gamma'])
        ed = rel(got['dx'], ref['dx']) if np.linalg.norm(ref['dx']) > 0 else float(np.linalg.norm(got['dx']))
        eP = rel(got['P_new'], ref['P_new'])
        worst['dx'] = max(worst['dx'], ed); worst['P'] = max(worst['P'], eP); worst['gamma'] = max(worst['gamma'], eg)
        if not (ok and eg < 1e-9 and ed < 1e-6 and eP < 1e-6):
            fails.append(dict(par, accept_equal=bool(ok), e_gamma=eg, e_dx=ed, e_P=eP))
    except Exception as e:
        fails.append(dict(par, error=repr(e)[:300]))
    n_done += 1
    seed += 1
print(json.dumps(dict(windows=n_done, first_seed=seed0, failures=fails, worst=worst), indent=1))

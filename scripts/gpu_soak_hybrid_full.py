"""Randomised parity soak of the WHOLE hybrid frame on the resident covariance (GPU box) against oracle/mirror_hybrid.py: MSCKF
tracks + SLAM features (some anchored at Schmidt nuisance states) + features entering the state, prior = resident covariance,
rows of the entering features, the joint update and the H_1 / H_2 tail on the device (orcvio_msckf_cov_commit_new_features).
usage: python scripts/gpu_soak_hybrid_full.py [seconds] [first_seed]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from orcvio_amd import capi, synth
from oracle import mirror_hybrid as mh
from helpers import rel

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
fails, n_done, n_skipped, worst = [], 0, 0, dict(dx=0.0, P=0.0)
t_end = time.time() + budget
seed = seed0
while time.time() < t_end:
    rng = np.random.default_rng(440000 + seed)
    N = int(rng.integers(8, 21))
    F = int(rng.choice([0, rng.integers(1, 40), rng.integers(40, 120)]))
    ns = int(rng.integers(0, 11)); nn = int(rng.integers(1, 7)); idp = int(rng.choice([1, 3])); nui = int(rng.choice([0, 0, 1, 3]))
    variant = int(rng.integers(0, 3))   # Jacobians of the MSCKF rows: LARVIO, OrcVIO left, OrcVIO right (config/euroc.yaml:114-118, kitti_raw.yaml:143-148)
    fl = synth.Flags(use_larvio=int(variant == 0), use_left_perturbation=int(variant == 1), if_fej=int(rng.integers(0, 2)), estimate_td=int(rng.integers(0, 2)))
    par = dict(seed=seed, N=N, F=F, slam=ns, new=nn, idp=idp, nui=nui, fej=fl.if_fej, td=fl.estimate_td)
    try:
        w0 = synth.make_window(N=N, F=F, seed=seed, track_len=(3, min(N, 10)), flags=fl)
        w = synth.with_extra_states(w0, idp * ns, seed=seed + 1)
        if nui:
            w = synth.with_nuisance_states(w, nui, seed=seed + 2)
        slam = synth.make_slam_features(w, ns, seed=seed, outlier_frac=0.25, nui_frac=0.4 if nui else 0.0)
        new = [mh.NewSlamFeature(**d) for d in synth.make_new_slam_features(w, nn, seed=seed + 3, outlier_frac=float(rng.choice([0.0, 0.3])))]
        ref = mh.hybrid_update_full(w, slam, new, idp)
        acc = ref['new_accept']
        if len(acc) == 0:
            n_skipped += 1
        else:
            upd.cov_set(w.P)
            upd.set_extra_states(w.n_extra); upd.set_schmidt_states(w.n_nui); upd.set_ekf_rows_mode(True)
            try:
                upd.upload(w, resident_cov=True)
                if nui:
                    upd.upload_nuisance_poses(w.nui)
                if ns:
                    upd.upload_slam_features(idp, slam)
                upd.upload_new_features(w, idp, [new[i] for i in acc])
                upd.run_update(); upd.sync()
                got = upd.download()
                dx_new = upd.cov_commit_new_features()
            finally:
                upd.set_ekf_rows_mode(False); upd.set_schmidt_states(0); upd.set_extra_states(0)
            ed = rel(np.concatenate([got['dx'], dx_new]), ref['dx'])
            eP = rel(upd.cov_get(), ref['P_new'])
            worst['dx'] = max(worst['dx'], ed); worst['P'] = max(worst['P'], eP)
            if not (ed < 1e-6 and eP < 1e-6):
                fails.append(dict(par, entering=len(acc), e_dx=ed, e_P=eP))
    except Exception as e:
        fails.append(dict(par, error=repr(e)[:300]))
    n_done += 1
    seed += 1
print(json.dumps(dict(frames=n_done, without_entering_features=n_skipped, first_seed=seed0, failures=fails, worst=worst), indent=1))

"""Randomised parity soak of the triangulation kernel on the GPU box against oracle/mirror_triangulate.py: random windows, ragged
and scattered tracks, spoiled tracks (mismatched / mirrored observations), random thresholds and iteration limits, tracks that
start from a prior position.  The Levenberg-Marquardt loop branches on cost comparisons, so a track whose decision sits on a
rounding error can legitimately take another path: failures are listed per TRACK with how far the decision was from its threshold.
usage: python scripts/gpu_soak_triangulate.py [seconds] [first_seed]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from orcvio_amd import capi, synth
from oracle import mirror_triangulate as mt
from helpers import scatter_tracks
from make_golden_tri import spoil

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
upd = capi.MsckfUpdater(device=0, max_clones=40, max_features=2048, max_observations=65536)
fails, n_win, n_tracks, n_valid, worst = [], 0, 0, 0, 0.0
t_end = time.time() + budget
seed = seed0
while time.time() < t_end:
    rng = np.random.default_rng(550000 + seed)
    N = int(rng.integers(3, 33))
    F = int(rng.integers(1, 80))
    lo = int(rng.integers(2, min(N, 5) + 1)); hi = int(rng.integers(lo, N + 1))
    par = dict(seed=seed, N=N, F=F, lo=lo, hi=hi)
    try:
        w = synth.make_window(N=N, F=F, seed=seed, track_len=None if rng.integers(0, 2) else (lo, hi), outlier_frac=float(rng.choice([0.0, 0.2])))
        if w.obs_ptr[1] - w.obs_ptr[0] == N and rng.integers(0, 2):
            w = scatter_tracks(w, rng, lo, hi)
        if rng.integers(0, 2):
            w = spoil(w, seed)
        cfg = mt.OptimizationConfig() if rng.integers(0, 2) else mt.OptimizationConfig(
            translation_threshold=float(rng.choice([0.2, 0.05, 0.5])), cost_threshold=float(rng.choice([1e-6, 1e-5, 1e-4])) if rng.integers(0, 2) else mt.OptimizationConfig().cost_threshold,
            outer_loop_max_iteration=int(rng.choice([1, 3, 10])), inner_loop_max_iteration=int(rng.choice([2, 5, 10])),
            huber_epsilon=float(rng.choice([1e-3, 0.01, 0.1])))
        ini = (rng.random(F) < 0.5).astype(np.int32) if rng.integers(0, 3) == 0 else None
        ref = mt.triangulate_tracks(w, cfg, is_initialized=ini)
        got = upd.triangulate(w, cfg=cfg, is_initialized=ini)
        n_tracks += F
        for j in range(F):
            same = got['valid'][j] == ref['valid'][j] and got['flags'][j] == ref['flags'][j]
            e = 0.0
            if same and ref['valid'][j] == 1:
                n_valid += 1
                e = float(np.linalg.norm(got['p_w'][j] - ref['p_w'][j]) / max(np.linalg.norm(ref['p_w'][j]), 1e-300))
                worst = max(worst, e)
            if not same or e > 1e-6:
                fails.append(dict(par, track=j, M=int(w.obs_ptr[j + 1] - w.obs_ptr[j]), valid=(int(got['valid'][j]), int(ref['valid'][j])),
                                  flags=(int(got['flags'][j]), int(ref['flags'][j])), cost=(float(got['cost'][j]), float(ref['cost'][j])), e_p=e,
                                  cfg=dict(cost_threshold=cfg.cost_threshold, outer=cfg.outer_loop_max_iteration, inner=cfg.inner_loop_max_iteration, huber=cfg.huber_epsilon)))
    except Exception as e:
        fails.append(dict(par, error=repr(e)[:300]))
    n_win += 1
    seed += 1
print(json.dumps(dict(windows=n_win, tracks=n_tracks, valid=n_valid, first_seed=seed0, failures=fails, worst_p=worst), indent=1, default=str))

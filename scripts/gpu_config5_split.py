"""Config 5 on one GPU (2 000 tracks under kitti_raw.yaml's flags): device-resident ms per update with the tracks front end as two
launches over E scratch in HBM (k_feature_e + k_feature_gate, the default from 1 800 tracks) and as one (k_feature, E in LDS:
ORCVIO_SPLIT_TRACKS=0).  Run each in a fresh process (the switch is read once)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import capi, synth
upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
out = {}
for F in (1000, 1500, 1800, 2000):
    win = synth.make_window(N=30, F=F, seed=0, flags=synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=1.0, discard_large_update=1))
    upd.upload(win)
    for _ in range(20): upd.run_update()
    upd.sync()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(100): upd.run_update()
        upd.sync()
        ts.append((time.perf_counter() - t0) / 100 * 1e3)
    out[F] = round(float(np.median(ts)), 4)
print('ORCVIO_SPLIT_TRACKS=%s' % os.environ.get('ORCVIO_SPLIT_TRACKS', '(default 1800)'), out)

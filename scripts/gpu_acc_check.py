import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi
from oracle import oracle
def rel(a, b): return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))
upd = capi.MsckfUpdater(max_clones=40, max_features=2048, max_observations=65536)
for cfg in (1, 2, 5):
    w = synth.config_window(cfg)
    o = oracle.msckf_update(w)
    g = upd.update_features(w, want_G=True)
    fin = np.isfinite(o['gamma'])
    print(cfg, 'gamma', rel(g['gamma'][fin], o['gamma'][fin]), 'dx', rel(g['dx'], o['dx']), 'P', rel(g['P_new'], o['P_new']), 'dP', rel(g['P_new'] - w.P, o['P_new'] - w.P), 'G', rel(g['G'], o['G']))

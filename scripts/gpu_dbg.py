import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi
from oracle import oracle
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536, debug_hooks=True)   # diagnostics build: orcvio_msckf_debug_* hooks
order = sys.argv[1:] or ['small', 'config1']
for nm in order:
    if nm == 'small':
        w = synth.make_window(N=12, F=40, seed=2, track_len=(2, 12), outlier_frac=0.2)
    elif nm == 'config1':
        w = synth.config_window(1)
    else:
        w = synth.config_window(2)
    g = upd.update_features(w, want_G=True, want_thin=True, want_K=True)
    print("   G nan", int(np.isnan(g["G"]).sum()), "K nan", int(np.isnan(g["K"]).sum()), "P nan", int(np.isnan(g["P_new"]).sum()), "Hthin nan", int(np.isnan(g["H_thin"]).sum()))
    d = capi.debug_read(upd, 'dims')
    print(nm, d, 'dx finite', np.isfinite(g['dx']).all())
    for b in ['A', 'RP', 'U', 'M', 'RM', 'Z']:
        X = capi.debug_read(upd, b)
        n = d['n']
        sub = X[:n, :n] if b in ('RP', 'M', 'RM') else X
        print('  ', b, X.shape, 'nan in full', int(np.isnan(X).sum()), 'nan in used', int(np.isnan(sub).sum()))
        if np.isnan(sub).any():
            idx = np.argwhere(np.isnan(sub))
            print('     first nan at', idx[:5].tolist(), 'rows', sorted(set(idx[:, 0].tolist()))[:10], 'cols', sorted(set(idx[:, 1].tolist()))[:10])

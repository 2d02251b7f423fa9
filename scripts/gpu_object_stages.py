"""Per-stage device times of the config-3 object update (HIP events between the stages), with and without the resident prior."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
win = synth.make_window(N=30, F=4, seed=0, flags=oflags, track_len=4)
objs = synth.make_objects(win, n_objects=20, seed=1, sigma_kp=0.004)
args = (oflags, win.N, objs, win.P, win.R_b2c[0], win.t_c_b[0], True, False, 0)
out = {}
for qr in (1, 0):
    upd._chk(upd.lib.orcvio_msckf_set_option(upd.h, 8, qr), 'opt')
    for _ in range(5):
        upd.update_object_tracks(*args)
    upd.set_stage_profile(True)
    runs = []
    for _ in range(20):
        upd.update_object_tracks(*args)
        runs.append(upd.profile_stages())
    upd.set_stage_profile(False)
    out['qr' if qr else 'gram'] = {name: round(float(np.median([r[i][1] for r in runs])) * 1e3, 1) for i, (name, _) in enumerate(runs[0])}
print(json.dumps(out, indent=1))

import ctypes as C, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
win = synth.make_window(N=30, F=4, seed=0, flags=oflags, track_len=4)
objs = synth.make_objects(win, n_objects=20, seed=1, sigma_kp=0.004)
upd.set_object_refine(2)
for i in range(6):
    upd.update_object_tracks(oflags, win.N, objs, win.P, win.R_b2c[0], win.t_c_b[0], True, False, 0)

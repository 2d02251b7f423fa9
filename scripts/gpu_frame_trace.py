"""The config-3 frame in one call (orcvio_msckf_io_update_frame), repeated: run under `rocprofv3 --kernel-trace` to see the object
tracks' compression beside the feature update's solve (scripts/frame_timeline.py turns the trace into a per-kernel timeline)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
fwin = synth.config_window(3)
oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
owin = synth.make_window(N=fwin.N, F=4, seed=0, flags=oflags, track_len=4)
objs = synth.make_objects(owin, n_objects=20, seed=1, sigma_kp=0.004)
for it in range(60):
    upd.cov_set(fwin.P)
    upd.cov_prefactor()
    upd.sync()
    f, o = upd.update_frame(fwin, oflags, objs, owin.R_b2c[0], owin.t_c_b[0], True, False, 0)
assert o['accept'] == 1

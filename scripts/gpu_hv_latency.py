"""Host-visible latency of orcvio_msckf_io_update at config 2 (SURVEY 8d's metric) under the switches of the diagnostics build.
usage: python scripts/gpu_hv_latency.py [ENV=VALUE ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for kv in sys.argv[1:]:
    k, v = kv.split('=')
    os.environ[k] = v
import numpy as np
from orcvio_amd import capi, synth

win = synth.config_window(2)
upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536, debug_hooks=True)


def lat(call, reps=400, after=None):
    for _ in range(30):
        call()
        if after:
            after()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        t.append((time.perf_counter() - t0) * 1e3)
        if after:
            after()
    a = np.sort(t)
    return round(float(np.median(a)), 5), round(float(a[int(0.95 * len(a))]), 5)


call_io, _ = upd.make_io_call(win)
out = dict(env=' '.join(sys.argv[1:]), host_visible=lat(call_io))
upd.cov_set(win.P)
call_res, _ = upd.make_io_call(win, resident_cov=True, want_P=False, commit=True)
out['resident_commit'] = lat(call_res, after=lambda: upd.cov_set(win.P))
out['counters'] = {k: v for k, v in upd.counters().items() if k.startswith('graph') or k == 'plain_runs'}
print(out)
upd.close()

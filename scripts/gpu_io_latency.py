"""Host-visible latency of one update at config 2 through the copying call and through the zero-copy arena (median / p95 in ms)."""
import gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import capi, synth


def timed(fn, reps=300, warm=20, after=None):
    for _ in range(warm):
        fn()
        if after: after()
    gc.collect(); gc.disable()
    out = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); out.append((time.perf_counter() - t) * 1e3)
        if after: after()
    gc.enable()
    a = np.sort(out)
    return dict(median=round(float(np.median(a)), 4), p95=round(float(a[int(0.95 * len(a))]), 4), min=round(float(a[0]), 4))


cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
win = synth.config_window(cfg)
upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
out = {}
call, _ = upd.make_update_call(win)
out['copying host_visible'] = timed(call)
call, io = upd.make_io_call(win)
out['io host_visible (P in, P+ out)'] = timed(call)
call, io = upd.make_io_call(win, want_P=False)
out['io P in, no P+ out'] = timed(call)
upd.cov_set(win.P)
call, io = upd.make_io_call(win, resident_cov=True, want_P=False, commit=True)
out['io resident + commit'] = timed(call, after=lambda: upd.cov_set(win.P))
def pre():
    upd.cov_set(win.P); upd.cov_prefactor(); upd.sync()
pre()
out['io resident prefactored + commit'] = timed(call, after=pre)
call2, _ = upd.make_update_call(win, resident_cov=True, want_P=False, commit=True)
upd.cov_set(win.P)
out['copying resident + commit'] = timed(call2, after=lambda: upd.cov_set(win.P))
upd.upload(win)
def dev():
    upd.run_update(); upd.sync()
out['device_resident (run_update + sync)'] = timed(dev)
print(json.dumps(out, indent=1))

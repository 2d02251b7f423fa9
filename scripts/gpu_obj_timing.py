"""Host-visible latency of config 3's object update (20 cars x 12 keypoints x 30 frames), host buffers and resident prior."""
import ctypes as C, gc, json, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
win = synth.make_window(N=30, F=4, seed=0, flags=oflags, track_len=4)
NOBJ = int(sys.argv[1]) if len(sys.argv) > 1 else 20
objs = synth.make_objects(win, n_objects=NOBJ, seed=1 if NOBJ == 20 else 4, sigma_kp=0.004)
ofl = capi.make_flags(oflags)
ef, arr, keep = upd._object_tracks(objs, win.R_b2c[0], win.t_c_b[0], True, False, 0, False)
Pc = np.ascontiguousarray(win.P)
o, res = upd._result(win.n, 1)
o2, res2 = upd._result(win.n, 1)
res2.P_out = None


def timed(fn, reps=200, warm=20):
    for _ in range(warm): fn()
    gc.collect(); gc.disable()
    out = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); out.append((time.perf_counter() - t) * 1e3)
    gc.enable()
    a = np.sort(out)
    return dict(median=round(float(np.median(a)), 4), p95=round(float(a[int(0.95 * len(a))]), 4), p75=round(float(a[int(0.75 * len(a))]), 4), max=round(float(a[-1]), 4),
                slow_at=[int(i) for i in np.nonzero(np.array(out) > 1.2 * np.median(a))[0][:40]])


def host():
    assert upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(ofl), C.byref(ef), win.N, arr, len(objs), capi._d(Pc), C.byref(res)) == 0
out = {'host buffers': timed(host)}
upd.cov_set(win.P); upd.cov_prefactor(); upd.sync()
def resident():
    assert upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(ofl), C.byref(ef), win.N, arr, len(objs), None, C.byref(res2)) == 0
out['resident prior + factor'] = timed(resident)
print(os.environ.get('ORCVIO_OBJ_INGEST', '1'), os.environ.get('ORCVIO_OBJ_PUBLISH', '1'), json.dumps(out))

"""Device-resident time per update for the BASELINE configs (feature part; config 3 adds the object update)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orcvio_amd import synth, capi
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
upd = capi.MsckfUpdater(max_clones=32, max_features=4096, max_observations=131072)
out = {}
for cfg in (1, 2, 4, 5):
    w = synth.config_window(cfg)
    upd.upload(w)
    for _ in range(20):
        upd.run_update()
    upd.sync()
    t0 = time.perf_counter()
    K = 200
    for _ in range(K):
        upd.run_update()
    upd.sync()
    dt = (time.perf_counter() - t0) / K
    prof = upd.profile(reps=10)
    out[f'config{cfg}'] = dict(N=w.N, F=w.F, rows=int(np.sum(np.maximum(2 * np.diff(w.obs_ptr) - 3, 0))), ms_per_update=round(dt * 1e3, 4),
                               kernel_us={k: round(v * 1e3, 1) for k, v in prof.items()})
    # host-inclusive
    for _ in range(10):   # the first calls at a new shape pay one-off costs (code load, graph capture): several ms
        upd.update_features(w)
    import gc
    gc.collect(); gc.disable()   # (an interpreter collection inside a call costs tens of ms: not the library's)
    t0 = time.perf_counter()
    for _ in range(20):
        upd.update_features(w)
    out[f'config{cfg}']['host_inclusive_ms'] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    gc.enable()
# config 3: 20 objects
from oracle import mirror_objects as mo
flags = synth.Flags(use_larvio=0, use_left_perturbation=0)
win = synth.make_window(N=30, F=4, seed=0, flags=flags, track_len=4)
objs = synth.make_objects(win, n_objects=20, seed=1, sigma_kp=0.004)
blocks = []
for ob in objs:
    res, Hf, Jc, counts = mo.object_rows(ob.wTo, ob.shape, ob.kps, ob.frames, True, False)
    Hx, Hf2, r, rc, hx6 = mo.construct_object_residual_jacobians(Jc, [fr['clone'] for fr in ob.frames], Hf, res, counts,
                                                                  [fr['wTc'] for fr in ob.frames], win.R_b2c[0], win.t_c_b[0], 0, 22, win.N)
    blocks.append(dict(row_clone=rc, Hx6=hx6, Hf=Hf2, res=r))
import ctypes as C
fl = capi.make_flags(flags)
Pc = np.ascontiguousarray(win.P)
n = win.n
arr_rows, keep1 = upd._object_blocks(blocks)
ef, arr_tr, keep2 = upd._object_tracks(objs, win.R_b2c[0], win.t_c_b[0], True, False, 0, False)
def call_rows():
    out, res = upd._result(n, 1)
    rc = upd.lib.orcvio_msckf_update_objects(upd.h, C.byref(fl), win.N, arr_rows, len(blocks), capi._d(Pc), C.byref(res))
    assert rc == 0
    return int(out['accept'][0])
def call_tracks():
    out, res = upd._result(n, 1)
    rc = upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(fl), C.byref(ef), win.N, arr_tr, len(objs), capi._d(Pc), C.byref(res))
    assert rc == 0
    return int(out['accept'][0])
res3 = {}
for name, fn in (('from_rows', call_rows), ('from_tracks', call_tracks)):
    for _ in range(10):
        fn()
    t0 = time.perf_counter()
    for _ in range(20):
        acc = fn()
    res3[name + '_host_inclusive_ms'] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    res3[name + '_accept'] = acc
out['config3_objects'] = dict(objects=20, rows=int(sum(len(b['res']) for b in blocks)), **res3)
# triangulation
w = synth.config_window(2)
for _ in range(3):
    upd.triangulate(w)
t0 = time.perf_counter()
for _ in range(20):
    upd.triangulate(w)
out['triangulate_config2_host_inclusive_ms'] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
print(json.dumps(out, indent=1))

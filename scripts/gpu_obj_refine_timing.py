"""Config 3's object update (20 cars x 12 keypoints x 30 frames) with ORCVIO_OPT_OBJECT_REFINE = 0 / 1 / 2: host-visible latency with
the prior and its factor resident, modes interleaved (five rounds of 200 calls each), and the stage profile of each mode."""
import ctypes as C, gc, json, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
win = synth.make_window(N=30, F=4, seed=0, flags=oflags, track_len=4)
objs = synth.make_objects(win, n_objects=20, seed=1, sigma_kp=0.004)
ofl = capi.make_flags(oflags)
ef, arr, keep = upd._object_tracks(objs, win.R_b2c[0], win.t_c_b[0], True, False, 0, False)
o2, res2 = upd._result(win.n, 1)
res2.P_out = None
upd.cov_set(win.P); upd.cov_prefactor(); upd.sync()


def resident():
    assert upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(ofl), C.byref(ef), win.N, arr, len(objs), None, C.byref(res2)) == 0


def timed(fn, reps=200, warm=20):
    for _ in range(warm): fn()
    gc.collect(); gc.disable()
    out = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); out.append((time.perf_counter() - t) * 1e3)
    gc.enable()
    return float(np.median(out))


med = {0: [], 1: [], 2: []}
for rnd in range(5):
    for mode in (0, 1, 2):
        upd.set_object_refine(mode)
        med[mode].append(timed(resident))
out = {'median_ms_by_mode': {str(m): [round(v, 4) for v in med[m]] for m in med}, 'refined': {}}
for mode in (0, 1, 2):
    upd.set_object_refine(mode)
    resident()
    out['refined'][str(mode)] = upd.objects_refined()
    upd.set_stage_profile(True)
    runs = []
    for _ in range(20):
        resident(); runs.append(upd.profile_stages())
    upd.set_stage_profile(False)
    out['stage_ms_mode%d' % mode] = {name: round(float(np.median([r[i][1] for r in runs])), 5) for i, (name, _) in enumerate(runs[0])}
print(json.dumps(out, indent=1))

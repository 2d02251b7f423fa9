"""Experiment: config 3's frame with the object COMPRESSION (tracks -> rows -> structured QR -> A') running beside the feature
update's solve, emulated with two handles (B compresses on its own stream; A runs the feature update, then the object solve on
B's block).  Sequential frame (bench.py's) against the overlapped one; the two must give the same results."""
import ctypes as C, gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi

A = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
B = capi.MsckfUpdater(max_clones=32, max_features=64, max_observations=4096)
lib = A.lib
fwin = synth.config_window(3)
oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
owin = synth.make_window(N=fwin.N, F=4, seed=0, flags=oflags, track_len=4)
objs = synth.make_objects(owin, n_objects=20, seed=1, sigma_kp=0.004)
ofl = capi.make_flags(oflags)
ef, arr, keep = A._object_tracks(objs, owin.R_b2c[0], owin.t_c_b[0], True, False, 0, False)
B.cov_set(owin.P)   # (B never solves: its resident matrix only has to match the window)


def sequential():
    A.upload(fwin, resident_cov=True)
    A.run_update()
    dx1 = A.download_dx().copy()
    A.cov_commit()
    oo, ores = A._result(owin.n, 1)
    ores.P_out = None
    assert lib.orcvio_msckf_update_object_tracks(A.h, C.byref(ofl), C.byref(ef), owin.N, arr, len(objs), None, C.byref(ores)) == 0
    A.cov_commit()
    return dx1, oo['dx'].copy(), int(oo['accept'][0])


def overlapped():
    A.upload(fwin, resident_cov=True)
    A.run_update()
    A.cov_commit()
    dof = C.c_int32(0)
    assert lib.orcvio_msckf_objects_local_tracks(B.h, C.byref(ofl), C.byref(ef), owin.N, arr, len(objs), None, None, C.byref(dof), None) == 0
    dx1 = A.download_dx().copy()
    d0 = C.c_int32(0)
    assert lib.orcvio_msckf_objects_local_tracks(A.h, C.byref(ofl), C.byref(ef), owin.N, None, 0, None, None, C.byref(d0), None) == 0
    A.n = owin.n
    B.sync()
    A.objects_finish(B.block_ptr()[0], 1, dof.value)
    out = A.objects_download()
    A.cov_commit()
    return dx1, out['dx'].copy(), out['accept']


def timed(fn, reps=100, warm=10):
    for _ in range(warm):
        A.cov_set(fwin.P); A.cov_prefactor(); A.sync(); fn()
    gc.collect(); gc.disable()
    ts = []
    for _ in range(reps):
        A.cov_set(fwin.P); A.cov_prefactor(); A.sync()
        t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
    gc.enable()
    a = np.sort(ts)
    return dict(median=round(float(np.median(a)), 4), p95=round(float(a[int(0.95 * len(a))]), 4))


A.cov_set(fwin.P); A.cov_prefactor(); A.sync()
s = sequential()
A.cov_set(fwin.P); A.cov_prefactor(); A.sync()
o = overlapped()
err = [float(np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-300)) for a, b in zip(s[:2], o[:2])]
print(json.dumps(dict(dx_feature_diff=err[0], dx_object_diff=err[1], accept=(s[2], o[2]), sequential=timed(sequential), overlapped=timed(overlapped))))

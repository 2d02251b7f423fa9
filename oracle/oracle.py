"""ctypes loader for oracle/liborcoracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  See the header of oracle/msckf_oracle.c for the parity status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, 'liborcoracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('msckf_oracle.c', 'msckf_fast.c', 'object_oracle.c', 'object_fast.c', 'Makefile')]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liborcoracle.so'], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_oracle_chi2_quantile.restype = C.c_double
        _LIB.orc_oracle_chi2_quantile.argtypes = [C.c_int, C.c_double]
        _LIB.orc_oracle_gating_gamma.restype = C.c_double
    return _LIB


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def flags_array(f) -> np.ndarray:
    return np.array([f.leg_dim, f.use_larvio, f.use_left_perturbation, f.if_fej, f.estimate_td],
                    dtype=np.int32)


def chi2_quantile(dof: int, p: float = 0.95) -> float:
    return float(lib().orc_oracle_chi2_quantile(int(dof), float(p)))


def chi2_table(p: float = 0.95, nmax: int = 500) -> np.ndarray:
    t = np.zeros(nmax)
    for d in range(1, nmax):
        t[d] = chi2_quantile(d, p)
    return t


def measurement_jacobian(win, i, p_w, z):
    fl = flags_array(win.flags)
    Hx = np.zeros((2, 6)); He = np.zeros((2, 6)); Hf = np.zeros((2, 3)); r = np.zeros(2)
    p_w = np.ascontiguousarray(p_w, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    lib().orc_oracle_measurement_jacobian(
        _i(fl), _d(win.R_b2w[i]), _d(win.t_b_w[i]), _d(win.t_fej[i]), _d(win.R_b2c[i]), _d(win.t_c_b[i]),
        _d(p_w), _d(z), _d(Hx), _d(He), _d(Hf), _d(r))
    return Hx, He, Hf, r


def msckf_update(win, clone_mask=None, want_blocks=True, want_K=True, table=None):
    """Runs the C oracle on a synth.Window.  Returns a dict like mirror.msckf_update
    plus 'seconds' (wall time of the C call)."""
    f = win.flags
    N, F, n = win.N, win.F, win.n
    fl = flags_array(f)
    table = chi2_table(f.chi2_prob) if table is None else np.ascontiguousarray(table, dtype=np.float64)
    nobs = int(win.obs_ptr[-1])
    dx = np.zeros(n); P_out = np.zeros((n, n))
    accept = np.zeros(F, dtype=np.int32); gamma = np.zeros(F)
    H_all = np.zeros((2 * nobs + 1, n)) if want_blocks else None
    r_all = np.zeros(2 * nobs + 1) if want_blocks else None
    block_ptr = np.zeros(F + 1, dtype=np.int32)
    H_thin = np.zeros((n, n)); r_thin = np.zeros(n)
    K = np.zeros((n, n)) if want_K else None
    G = np.zeros((n, n))
    info = np.zeros(2, dtype=np.int32)
    mask = None if clone_mask is None else np.ascontiguousarray(clone_mask, dtype=np.int32)
    zvel = win.obs_zvel if win.obs_zvel is not None else np.zeros((nobs, 2))
    t0 = time.perf_counter()
    rc = lib().orc_oracle_msckf_update(
        C.c_int(N), C.c_int(F), _i(fl), C.c_double(f.noise_feature), C.c_double(f.chi2_prob),
        _d(table), C.c_int(len(table)),
        _d(win.R_b2w), _d(win.t_b_w), _d(win.t_fej), _d(win.R_b2c), _d(win.t_c_b),
        _d(win.p_w), _i(win.obs_ptr), _i(win.obs_clone), _d(win.obs_z), _d(zvel),
        _i(mask), _d(win.P),
        _d(dx), _d(P_out), _i(accept), _d(gamma),
        _d(H_all), _d(r_all), _i(block_ptr),
        _d(H_thin), _d(r_thin), _d(K), _d(G), _i(info))
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError('oracle update failed (S not positive definite)')
    mt = int(info[1])
    out = dict(dx=dx, P_new=P_out, accept=accept, gamma=gamma, G=G, seconds=dt,
               stacked_rows=int(info[0]), thin_rows=mt, block_ptr=block_ptr,
               H_thin=H_thin[:mt], r_thin=r_thin[:mt], updated=mt > 0)
    if want_K:
        out['K'] = K.reshape(-1)[: n * mt].reshape(n, mt)
    if want_blocks:
        out['H_all'] = H_all[: block_ptr[-1]]
        out['r_all'] = r_all[: block_ptr[-1]]
    return out


def msckf_update_fast(win, table=None, threads=None):
    """oracle/msckf_fast.c: the same update with the minimum-work algorithm, tracks parallelised with OpenMP over the host cores
    (bench.py's all-cores CPU baseline; also a third independently written evaluation).  dict(dx, P_new, accept, gamma, seconds,
    threads)."""
    f = win.flags
    N, F, n = win.N, win.F, win.n
    fl = flags_array(f)
    table = chi2_table(f.chi2_prob) if table is None else np.ascontiguousarray(table, dtype=np.float64)
    nobs = int(win.obs_ptr[-1])
    dx = np.zeros(n); P_out = np.zeros((n, n))
    accept = np.zeros(F, dtype=np.int32); gamma = np.zeros(F)
    zvel = win.obs_zvel if win.obs_zvel is not None else np.zeros((nobs, 2))
    used = C.c_int(0)
    if threads is not None:
        os.environ['OMP_NUM_THREADS'] = str(int(threads))
    t0 = time.perf_counter()
    rc = lib().orc_fast_msckf_update(
        C.c_int(N), C.c_int(F), _i(fl), C.c_double(f.noise_feature), C.c_double(f.chi2_prob), _d(table), C.c_int(len(table)),
        _d(win.R_b2w), _d(win.t_b_w), _d(win.t_fej), _d(win.R_b2c), _d(win.t_c_b),
        _d(win.p_w), _i(win.obs_ptr), _i(win.obs_clone), _d(win.obs_z), _d(zvel), _d(win.P),
        _d(dx), _d(P_out), _i(accept), _d(gamma), C.byref(used))
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError('fast CPU update failed (a matrix was not positive definite)')
    return dict(dx=dx, P_new=P_out, accept=accept, gamma=gamma, seconds=dt, threads=int(used.value))


def object_rows_c(obj, R_b2c, t_c_b, obj_left, new_bbox, vio_left, fix_D=False):
    """oracle/object_oracle.c: residual rows of one object track (synth.ObjectTrack-shaped) in window coordinates.
    dict(row_clone, Hx6, Hf, res) or None if no frame is in the window."""
    K = obj.kps.shape[0]
    F = len(obj.frames)
    ncol = 9 + 3 * K
    cap = F * (2 * K + 4)
    wTo = np.ascontiguousarray(obj.wTo, dtype=np.float64)
    shape = np.ascontiguousarray(obj.shape, dtype=np.float64)
    kps = np.ascontiguousarray(obj.kps, dtype=np.float64)
    wTc = np.ascontiguousarray(np.stack([fr['wTc'] for fr in obj.frames]), dtype=np.float64)
    zs = np.ascontiguousarray(np.stack([fr['zs'] for fr in obj.frames]), dtype=np.float64)
    bb = np.ascontiguousarray(np.stack([fr['bbox'] for fr in obj.frames]), dtype=np.float64)
    cl = np.ascontiguousarray([fr['clone'] for fr in obj.frames], dtype=np.int32)
    Rb = np.ascontiguousarray(R_b2c, dtype=np.float64)
    tb = np.ascontiguousarray(t_c_b, dtype=np.float64)
    row_clone = np.zeros(cap, dtype=np.int32)
    Hx6 = np.zeros((cap, 6)); Hf = np.zeros((cap, ncol)); res = np.zeros(cap)
    m = lib().orc_oracle_object_rows(C.c_int(K), C.c_int(F), _d(wTo), _d(shape), _d(kps), _d(wTc), _d(zs), _d(bb), _i(cl),
                                     C.c_int(int(obj_left)), C.c_int(int(new_bbox)), C.c_int(int(vio_left)), C.c_int(int(fix_D)),
                                     _d(Rb), _d(tb), _i(row_clone), _d(Hx6), _d(Hf), _d(res))
    if m == 0:
        return None
    return dict(row_clone=row_clone[:m].copy(), Hx6=Hx6[:m].copy(), Hf=Hf[:m].copy(), res=res[:m].copy())


def _object_block_arrays(blocks):
    ncol = np.array([b['Hf'].shape[1] for b in blocks], dtype=np.int32)
    ncmax = int(ncol.max()) if len(blocks) else 1
    ptr = np.concatenate([[0], np.cumsum([len(b['res']) for b in blocks])]).astype(np.int32)
    rows = int(ptr[-1])
    rc = np.ascontiguousarray(np.concatenate([b['row_clone'] for b in blocks]) if blocks else np.zeros(0), dtype=np.int32)
    hx = np.ascontiguousarray(np.vstack([b['Hx6'] for b in blocks]) if blocks else np.zeros((0, 6)), dtype=np.float64)
    hf = np.zeros((rows, ncmax))
    for b, a0 in zip(blocks, ptr[:-1]):
        hf[a0:a0 + len(b['res']), :b['Hf'].shape[1]] = b['Hf']
    rs = np.ascontiguousarray(np.concatenate([b['res'] for b in blocks]) if blocks else np.zeros(0), dtype=np.float64)
    return ncol, ncmax, ptr, rc, hx, hf, rs


def objects_update_fast(flags, n_clones, blocks, P, threads=None):
    """oracle/object_fast.c: the object update with the minimum-work algorithm (Schur-complement projection from the 7 non-zeros
    per row, objects parallelised with OpenMP; bench.py's all-cores CPU figure for the object update, and a second independently
    written evaluation).  dict(accept, gamma, dof, dx, P_new, seconds, threads)."""
    n = flags.leg_dim + 6 * n_clones
    ncol, ncmax, ptr, rc, hx, hf, rs = _object_block_arrays(blocks)
    Pc = np.ascontiguousarray(P, dtype=np.float64)
    acc = C.c_int(0); gam = C.c_double(0.0); dof = C.c_int(0); used = C.c_int(0)
    dx = np.zeros(n); Pn = np.zeros((n, n))
    if threads is not None:
        lib().orc_fast_objects_set_threads(C.c_int(int(threads)))
    t0 = time.perf_counter()
    rcode = lib().orc_fast_objects_update(C.c_int(n_clones), C.c_int(flags.leg_dim), C.c_int(len(blocks)), _i(ptr), _i(ncol), C.c_int(ncmax),
                                          _i(rc), _d(hx), _d(hf), _d(rs), _d(Pc), C.c_double(flags.noise_feature), C.c_double(flags.chi2_prob),
                                          C.byref(acc), C.byref(gam), C.byref(dof), _d(dx), _d(Pn), C.byref(used))
    dt = time.perf_counter() - t0
    if threads is not None:
        lib().orc_fast_objects_set_threads(C.c_int(0))
    if rcode != 0:
        raise RuntimeError('fast object update failed (a matrix was not positive definite)')
    return dict(accept=int(acc.value), gamma=float(gam.value), dof=int(dof.value), dx=dx, P_new=Pn, seconds=dt, threads=int(used.value))


def objects_update_c(flags, n_clones, blocks, P):
    """oracle/object_oracle.c: the object update (per-object projection, joint gate, measurementUpdate_msckf) from row blocks
    dict(row_clone, Hx6, Hf, res).  dict(accept, gamma, dof, dx, P_new, seconds)."""
    n = flags.leg_dim + 6 * n_clones
    ncol = np.array([b['Hf'].shape[1] for b in blocks], dtype=np.int32)
    ncmax = int(ncol.max()) if len(blocks) else 1
    ptr = np.concatenate([[0], np.cumsum([len(b['res']) for b in blocks])]).astype(np.int32)
    rows = int(ptr[-1])
    rc = np.ascontiguousarray(np.concatenate([b['row_clone'] for b in blocks]) if blocks else np.zeros(0), dtype=np.int32)
    hx = np.ascontiguousarray(np.vstack([b['Hx6'] for b in blocks]) if blocks else np.zeros((0, 6)), dtype=np.float64)
    hf = np.zeros((rows, ncmax))
    for b, a0 in zip(blocks, ptr[:-1]):
        hf[a0:a0 + len(b['res']), :b['Hf'].shape[1]] = b['Hf']
    rs = np.ascontiguousarray(np.concatenate([b['res'] for b in blocks]) if blocks else np.zeros(0), dtype=np.float64)
    Pc = np.ascontiguousarray(P, dtype=np.float64)
    acc = C.c_int(0); gam = C.c_double(0.0); dof = C.c_int(0)
    dx = np.zeros(n); Pn = np.zeros((n, n))
    t0 = time.perf_counter()
    rcode = lib().orc_oracle_objects_update(C.c_int(n_clones), C.c_int(flags.leg_dim), C.c_int(len(blocks)), _i(ptr), _i(ncol), C.c_int(ncmax),
                                            _i(rc), _d(hx), _d(hf), _d(rs), _d(Pc), C.c_double(flags.noise_feature), C.c_double(flags.chi2_prob),
                                            C.byref(acc), C.byref(gam), C.byref(dof), _d(dx), _d(Pn))
    dt = time.perf_counter() - t0
    if rcode != 0:
        raise RuntimeError('object oracle: S not positive definite')
    return dict(accept=int(acc.value), gamma=float(gam.value), dof=int(dof.value), dx=dx, P_new=Pn, seconds=dt)

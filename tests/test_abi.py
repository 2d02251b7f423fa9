"""CPU tests of the C-ABI library: it loads, exports every declared symbol, its host-side
pieces (chi-square quantile, state increment) are right, and it refuses to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import mirror
from helpers import GOLDEN, rel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = []
    for fn in os.listdir(os.path.join(ROOT, 'include')):
        if not fn.endswith('.h'):
            continue
        src = open(os.path.join(ROOT, 'include', fn)).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        names += re.findall(r'\b(orcvio_msckf_[a-z0-9_]+)\s*\(', src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(built):
    lib = capi.load()
    declared = _declared_functions()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/ but not exported'
    assert sorted(capi.EXPORTS) == declared
    assert lib.orcvio_msckf_abi_version() == 1
    # the product library exports the declared C-ABI and NOTHING of the test scaffolding; the diagnostics build does
    import subprocess
    def exported(path):
        out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True, check=True).stdout
        return sorted(set(re.findall(r'\b(orcvio_msckf_[a-z0-9_]+)\b', out)))
    assert exported(capi.LIB_PATH) == declared
    dbg = exported(capi.LIB_DBG_PATH)
    assert set(declared) < set(dbg) and all(n.startswith('orcvio_msckf_debug_') for n in set(dbg) - set(declared))


def test_chi2_quantile_matches_table(built):
    g = np.load(os.path.join(GOLDEN, 'chi2_095.npz'))
    for dof in list(range(1, 500, 7)) + [499]:
        assert abs(capi.chi2_quantile(dof, 0.95) - g['table'][dof]) / g['table'][dof] < 1e-12
    for dof, val in g['big']:
        assert abs(capi.chi2_quantile(int(dof), 0.95) - val) / val < 1e-12
    assert np.isnan(capi.chi2_quantile(0, 0.95))


def _state_struct(st, N):
    s = capi.MsckfState()
    for name in ('R_b2w_imu', 'R_b2c'):
        getattr(s, name)[:] = list(np.asarray(st[name]).ravel())
    for name in ('v', 'p', 'bg', 'ba', 't_c_b'):
        getattr(s, name)[:] = list(st[name])
    s.td = float(st['td'])
    s.n_clones = N
    bufs = dict(R=np.ascontiguousarray(st['R_b2w']).copy(), t=np.ascontiguousarray(st['t_b_w']).copy(),
                Rc=np.zeros((N, 3, 3)), tc=np.zeros((N, 3)))
    dp = C.POINTER(C.c_double)
    s.clone_R_b2w = bufs['R'].ctypes.data_as(dp)
    s.clone_t_b_w = bufs['t'].ctypes.data_as(dp)
    s.clone_R_c2w = bufs['Rc'].ctypes.data_as(dp)
    s.clone_t_c_w = bufs['tc'].ctypes.data_as(dp)
    return s, bufs


@pytest.mark.parametrize('larvio,left', [(1, 0), (0, 0), (0, 1)])
def test_increment_state_matches_mirror(built, larvio, left):
    rng = np.random.default_rng(4)
    N = 5
    w = synth.make_window(N=N, F=2, seed=1, track_len=3)
    f = synth.Flags(use_larvio=larvio, use_left_perturbation=left)
    st = dict(R_b2w_imu=w.R_b2w[-1].copy(), v=rng.standard_normal(3), p=w.t_b_w[-1].copy(), bg=rng.standard_normal(3) * 1e-2,
              ba=rng.standard_normal(3) * 1e-2, R_b2c=w.R_b2c[0].copy(), t_c_b=w.t_c_b[0].copy(), td=np.float64(0.01),
              R_b2w=w.R_b2w.copy(), t_b_w=w.t_b_w.copy())
    dx = rng.standard_normal(22 + 6 * N) * 0.02
    ref, applied = mirror.increment_state(st, dx, f)
    s, bufs = _state_struct(st, N)
    fl = capi.make_flags(f)
    rc = capi.load().orcvio_msckf_increment_state(C.byref(fl), dx.ctypes.data_as(C.POINTER(C.c_double)), C.byref(s))
    assert rc == 1 and applied
    assert rel(np.array(s.R_b2w_imu[:]).reshape(3, 3), ref['R_b2w_imu']) < 1e-12
    assert rel(np.array(s.R_b2c[:]).reshape(3, 3), ref['R_b2c']) < 1e-12
    for a, b in (('v', 'v'), ('p', 'p'), ('bg', 'bg'), ('ba', 'ba'), ('t_c_b', 't_c_b')):
        assert rel(np.array(getattr(s, a)[:]), ref[b]) < 1e-12
    assert abs(s.td - ref['td']) < 1e-15
    assert rel(bufs['R'], ref['R_b2w']) < 1e-12 and rel(bufs['t'], ref['t_b_w']) < 1e-12
    assert rel(bufs['Rc'], ref['R_c2w']) < 1e-12 and rel(bufs['tc'], ref['t_c_w']) < 1e-12


def test_increment_state_discards_large_update(built):
    f = synth.Flags(discard_large_update=1)
    w = synth.make_window(N=2, F=1, seed=1, track_len=2)
    st = dict(R_b2w_imu=np.eye(3), v=np.zeros(3), p=np.zeros(3), bg=np.zeros(3), ba=np.zeros(3), R_b2c=np.eye(3),
              t_c_b=np.zeros(3), td=np.float64(0), R_b2w=w.R_b2w.copy(), t_b_w=w.t_b_w.copy())
    dx = np.zeros(34)
    dx[3] = 1.5   # |dv| > 1  -> reference returns without touching the state (src/orcvio.cpp:4479-4494)
    s, bufs = _state_struct(st, 2)
    fl = capi.make_flags(f)
    rc = capi.load().orcvio_msckf_increment_state(C.byref(fl), dx.ctypes.data_as(C.POINTER(C.c_double)), C.byref(s))
    assert rc == 0
    assert np.array_equal(bufs['t'], w.t_b_w) and s.v[0] == 0.0


def test_no_cpu_fallback(built):
    """Without a GPU the compute path must fail loudly, not route anywhere else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    h = C.c_void_p()
    rc = capi.load().orcvio_msckf_create(0, 8, 8, 64, C.byref(h))
    assert rc == 2   # ORCVIO_ERR_NO_DEVICE
    with pytest.raises(capi.MsckfError):
        capi.MsckfUpdater()


def test_the_product_library_reads_only_the_documented_environment_variables(built):
    """the header documents ten environment variables; nothing else that looks like a switch may be left in the
    product library (the ablation / stamp / experiment switches are compiled out of it: dbg_getenv, msckf_capi.hip), while the
    diagnostics build still knows them."""
    import re
    from orcvio_amd import build as b
    header = open(os.path.join(ROOT, 'include', 'orcvio_msckf.h')).read()
    block = header[header.index(' * Environment.'):header.index('#ifndef ORCVIO_MSCKF_H')]
    documented = set(re.findall(r'^ \*   (ORCVIO_[A-Z_0-9]+) ', block, flags=re.M))
    assert len(documented) == 10, documented
    not_env = ('ORCVIO_OPT_', 'ORCVIO_ERR_', 'ORCVIO_MAX_', 'ORCVIO_OK', 'ORCVIO_CHI2_', 'ORCVIO_POSE_', 'ORCVIO_COUNTERS', 'ORCVIO_COMM_ID_BYTES',
               'ORCVIO_SHARD_', 'ORCVIO_IPC_MAX_', 'ORCVIO_DEBUG_HOOKS', 'ORCVIO_MSCKF_')

    def names(path):
        blob = open(path, 'rb').read()
        return {m.decode() for m in re.findall(rb'ORCVIO_[A-Z][A-Z_0-9]+', blob) if not m.decode().startswith(not_env)}
    prod = names(b.LIB)
    assert prod <= documented, prod - documented
    assert {'ORCVIO_COMM_TRANSPORT', 'ORCVIO_LA_SPIN', 'ORCVIO_FRAME_CHAIN'} <= prod
    dbg = names(b.LIB_DBG)
    assert {'ORCVIO_FUSE_FINISH', 'ORCVIO_BLK2', 'ORCVIO_SPLIT_TRACKS', 'ORCVIO_TIMING'} <= dbg


def test_the_committed_pmc_profile_is_of_these_sources():
    """bench.py quotes roofline.traffic from the newest profiles/r*_pmc_traffic.json only if that profile was taken on the loaded
    library or on another build of the SAME sources (hipcc's output is not reproducible byte for byte): the hash over the library's
    source files is stable, and the committed profile of the round's last build carries the hash of the tree as it stands."""
    import glob
    import json
    import re
    from orcvio_amd import build as b
    s1, s2 = b.source_sha16(), b.source_sha16()
    assert s1 == s2 and len(s1) == 16
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def key(p):
        m = re.match(r'r(\d+)([a-z]*)_', os.path.basename(p))
        return (int(m.group(1)), m.group(2)) if m else (-1, '')
    newest = sorted(glob.glob(os.path.join(root, 'profiles', 'r*_pmc_traffic.json')), key=key)[-1]
    rec = json.load(open(newest)).get('build')
    assert isinstance(rec, dict) and rec.get('source_sha16') == s1, (newest, rec, s1)

"""Diagnostic: 100 MHz time stamps inside k_front (workgroup 1 and the factorisation workgroup)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536, debug_hooks=True)   # diagnostics build: orcvio_msckf_debug_* hooks
upd.upload(synth.config_window(2))
for _ in range(5):
    upd.run_update(); upd.sync()
buf = np.zeros(32, dtype=np.uint64)
lib = upd.lib
lib.orcvio_msckf_debug_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
rc = lib.orcvio_msckf_debug_read(upd.h, 9, buf.ctypes.data_as(C.c_void_p), 256)
st = buf[8:16].astype(np.int64)
t0 = st[0]
names = ['potrf start', 'features done (wg 1)', 'barrier 1 passed', 'grams done', 'barrier 2 passed', 'assembled', 'potrf done']
for i, nm in enumerate(names):
    print(f'{nm:24s} {(st[i] - t0) * 0.01:8.2f} us')
print('barrier 2: stores acknowledged + workgroup barrier at %.2f us' % ((st[7] - t0) * 0.01))
print('counter', buf[0] & 0xffffffff)

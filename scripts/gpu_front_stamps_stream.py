"""Diagnostic: 100 MHz time stamps inside k_front for windows of the stream's shape (20 clones, short tracks, 12 extra states)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=24, max_features=256, max_observations=4096, debug_hooks=True)
upd.set_extra_states(12)
lib = upd.lib
lib.orcvio_msckf_debug_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
names = ['potrf start', 'features done (wg 1)', 'barrier 1 passed', 'grams done', 'barrier 2 passed', 'assembled', 'potrf done']
for F in (20, 60, 100, 200):
    w = synth.with_extra_states(synth.make_window(N=20, F=F, seed=F, track_len=(3, 6), flags=synth.Flags(use_larvio=1), outlier_frac=0.05), 12, seed=1)
    upd.upload(w)
    acc = np.zeros(8)
    for it in range(8):
        upd.run_update(); upd.sync()
        buf = np.zeros(32, dtype=np.uint64)
        lib.orcvio_msckf_debug_read(upd.h, 9, buf.ctypes.data_as(C.c_void_p), 256)
        st = buf[8:16].astype(np.int64)
        if it >= 3: acc += (st - st[0]) * 0.01 / 5
    print(f'F = {F}: ' + ' | '.join(f'{nm} {acc[i]:.1f}' for i, nm in enumerate(names)))

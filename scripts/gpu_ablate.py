import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536, debug_hooks=True)   # diagnostics build: orcvio_msckf_debug_* hooks
upd.upload(synth.config_window(2))
f = upd.lib.orcvio_msckf_debug_potrf_ablate
f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
names = {0: 'full', 1: 'no sweep', 2: 'no trailing', 4: 'no panel mfma', 8: 'no later-diag', 15: 'skeleton only', 14: 'sweep only', 13: 'trailing only', 10: 'no trailing, no later-diag', 31: 'skeleton, wave0 idle', 47: 'skeleton, role2 idle', 127: 'barriers + prologue only'}
for ab, nm in names.items():
    us = C.c_double()
    rc = f(upd.h, ab, 50, C.byref(us))
    print(f'ablate {ab:2d} {nm:28s} rc {rc}  {us.value:8.1f} us')

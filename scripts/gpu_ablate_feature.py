import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536, debug_hooks=True)   # diagnostics build: orcvio_msckf_debug_* hooks
upd.upload(synth.config_window(2))
f = upd.lib.orcvio_msckf_debug_feature_ablate
f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
for ab, nm in {0: 'full', 1: 'no E', 2: 'no QtEQ', 4: 'no Cholesky', 8: 'no outputs', 7: 'Jacobians+QR+Y+outputs', 15: 'Jacobians+QR+Y only', 14: 'E only (+base)', 11: 'Cholesky only (+base)'}.items():
    us = C.c_double()
    rc = f(upd.h, ab, 50, C.byref(us))
    print(f'ablate {ab:2d} {nm:28s} rc {rc}  {us.value:8.1f} us')

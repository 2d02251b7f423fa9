"""Diagnostic: k_feature time against the number of tracks (one or two workgroups per CU)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536, debug_hooks=True)   # diagnostics build: orcvio_msckf_debug_* hooks
f = upd.lib.orcvio_msckf_debug_feature_ablate
f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
for F in (64, 128, 256, 320, 400, 512, 768, 1024):
    upd.upload(synth.make_window(N=30, F=F, seed=0, flags=synth.Flags(use_larvio=1)))
    us = C.c_double()
    rc = f(upd.h, 0, 50, C.byref(us))
    print(f'F {F:5d}  rc {rc}  k_feature {us.value:8.1f} us')

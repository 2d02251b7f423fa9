import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
w = synth.config_window(2)
upd.upload(w)
nb = 13
st = np.zeros((nb, 8, 8), dtype=np.uint64)
upd.lib.orcvio_msckf_debug_potrf_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
rc = upd.lib.orcvio_msckf_debug_potrf_stamps(upd.h, st.ctypes.data_as(C.c_void_p), nb)
print('rc', rc)
t0 = st[0, 0].min()
st = (st - t0).astype(np.int64)
for kb in range(nb):
    s0 = st[kb, 0].max(); s1 = st[kb, 1].max(); s2 = st[kb, 2].max(); s3 = st[kb, 3].max()
    nxt = st[kb + 1, 0].max() if kb + 1 < nb else s3
    print(f'kb {kb:2d}  arrive-A {st[kb,0].min():7d}..{s0:7d}  syncA {s1 - s0:5d}  panel {s2 - s1:6d}  syncB {s3 - s2:5d}  diag+trailing {nxt - s3:6d}')
print('total cycles', st[nb - 1, 3].max())
print('per-wave panel durations (s2-s1) kb=0..3:')
for kb in range(4):
    print(kb, (st[kb, 2] - st[kb, 1]).tolist(), ' trailing+diag (next s0 - s3):', (st[kb + 1, 0] - st[kb, 3]).tolist())

print('panel detail kb=1: per wave [after-li-wait - s1, mid-slots - li, end - mid]')
for w in range(8):
    print(w, int(st[1,4,w]-st[1,1,w]), int(st[1,5,w]-st[1,4,w]), int(st[1,2,w]-st[1,5,w]))
print('owner detail: [update-diag (s6 - s3), factor (s7 - s6)] per kb')
for kb in range(nb-1):
    kn = kb+1; own = (kn*(kn+1)//2+kn) % 8
    print(kb, 'owner', own, int(st[kb,6,own]-st[kb,3,own]), int(st[kb,7,own]-st[kb,6,own]))

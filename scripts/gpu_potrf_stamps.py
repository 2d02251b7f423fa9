"""Diagnostic: where the time goes inside k_potrf_reg (core-clock stamps of wave 0 and wave 1)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536, debug_hooks=True)   # diagnostics build: orcvio_msckf_debug_* hooks
upd.upload(synth.config_window(2))
if os.environ.get('ORCVIO_POTRF_COLD'):   # factor M right behind the k_gemm that writes it (the conditions of the replayed graph)
    upd.run_update(); upd.sync()
f = upd.lib.orcvio_msckf_debug_potrf_stamps
f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
buf = (C.c_uint64 * 256)()
rc = f(upd.h, buf)
st = np.array(list(buf), dtype=np.int64)
w0, w1, w2 = st[0:64], st[64:128], st[128:192]
t0 = w0[0]
wall = (st[193] - st[192]) * 10e-3   # us (100 MHz)
cyc = w0[63] - w0[0]
print(f'rc {rc}  wall {wall:.1f} us  cycles {cyc}  -> {cyc / wall / 1e3:.2f} GHz')
print('wave0: start->loads %d, factor0 %d' % (w0[1] - t0, w0[2] - w0[1]))
print('wave1: start %d, all loads issued %d, all tiles arrived %d, row 0 staged %d; wave 4: diagonal tiles in LDS %d' % (w1[0] - t0, w1[1] - t0, w1[56] - t0, w1[57] - t0, w2[50] - t0))
print(' kb |  w0: waitA  waitB  dupd  factor | w1: toA   A    panel  B   trailing')
for kb in range(13):
    a, b, c, d = w0[3 + 4 * kb], w0[4 + 4 * kb], w0[5 + 4 * kb], w0[6 + 4 * kb]
    prev = w0[2] if kb == 0 else w0[6 + 4 * (kb - 1)]
    x0, x1, x2, x3 = w1[2 + 4 * kb], w1[3 + 4 * kb], w1[4 + 4 * kb], w1[5 + 4 * kb]
    nxt = w1[2 + 4 * (kb + 1)] if kb < 12 else w1[63]
    print(f' {kb:2d} | {a - prev:6d} {b - a:6d} {c - b if c else 0:6d} {d - c if d else 0:6d} | {x0 - t0:7d} {x1 - x0:5d} {x2 - x1:5d} {x3 - x2:5d} {nxt - x3:6d}')
print('end w0 %d  w1 %d' % (w0[63] - t0, w1[63] - t0))
print('sweep cycles per step:', [int(w2[3 * k + 1] - w2[3 * k]) for k in range(13)])
print('staging (step start -> sweep start):', [int(w2[3 * k] - (w0[1] if k == 0 else w0[5 + 4 * (k - 1)])) for k in range(13)])
print('sweep end -> factor end:', [int((w0[2] if k == 0 else w0[6 + 4 * (k - 1)]) - w2[3 * k + 1]) for k in range(13)])


"""Diagnostic: where the host-inclusive time of one update goes (upload / run / sync / download)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
w = synth.config_window(2)
for _ in range(5):
    upd.update_features(w)
T = dict(upload=0.0, run=0.0, sync=0.0, download=0.0, oneshot=0.0)
R = 50
for _ in range(R):
    t0 = time.perf_counter(); upd.upload(w)
    t1 = time.perf_counter(); upd.run_update()
    t2 = time.perf_counter(); upd.sync()
    t3 = time.perf_counter(); upd.download()
    t4 = time.perf_counter(); upd.update_features(w)
    t5 = time.perf_counter()
    T['upload'] += t1 - t0; T['run'] += t2 - t1; T['sync'] += t3 - t2; T['download'] += t4 - t3; T['oneshot'] += t5 - t4
print({k: round(1e6 * v / R, 1) for k, v in T.items()}, 'us')
# isolated phases, each bracketed by a full sync
import ctypes as C
T2 = dict(upload=0.0, upload_again=0.0)
for _ in range(R):
    upd.sync()
    t0 = time.perf_counter(); upd.upload(w); t1 = time.perf_counter(); upd.upload(w); t2 = time.perf_counter()
    T2['upload'] += t1 - t0; T2['upload_again'] += t2 - t1
    upd.run_update(); upd.sync(); upd.download()
print({k: round(1e6 * v / R, 1) for k, v in T2.items()}, 'us')

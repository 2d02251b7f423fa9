"""Diagnostic: where the host-visible time of orcvio_msckf_update_features goes (ORCVIO_TIMING breakdown inside the library:
upload = checks + staging + H2D enqueue, enqueue = graph launch + D2H enqueue, sync, unpack)."""
import sys, os, time
os.environ['ORCVIO_TIMING'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
w = synth.config_window(2)
print('--- P uploaded, P+ downloaded', file=sys.stderr)
call, out1 = upd.make_update_call(w)
for _ in range(260):
    call()
print('--- resident covariance, dx only', file=sys.stderr)
upd.cov_set(w.P)
call, out2 = upd.make_update_call(w, resident_cov=True, want_P=False, commit=False)
for _ in range(260):
    call()
t = []
for _ in range(300):
    t0 = time.perf_counter(); call(); t.append(time.perf_counter() - t0)
print('resident median us', 1e6 * float(np.median(t)))

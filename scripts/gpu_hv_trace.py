"""The host-visible in-place update at config 2, repeated (run under `rocprofv3 --kernel-trace`; scripts/trace_timeline.py shows the
launches and the gaps between them): first with P through the arena and P+ back, then on the resident covariance with the commit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orcvio_amd import capi, synth
win = synth.config_window(2)
upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
call_io, _ = upd.make_io_call(win)
for _ in range(40):
    call_io()
time.sleep(0.002)
upd.cov_set(win.P)
call_res, _ = upd.make_io_call(win, resident_cov=True, want_P=False, commit=True)
for _ in range(40):
    call_res()
    upd.cov_set(win.P)
    time.sleep(0.0005)

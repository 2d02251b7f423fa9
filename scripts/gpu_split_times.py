"""Tracks front end for many tracks: k_feature against k_feature_e + k_feature_gate (ORCVIO_SPLIT_TRACKS), device-resident ms per update
and the launch's own time, over the track count."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=4096, max_observations=131072)
out = {}
for F in (600, 800, 1024, 1500, 2000, 3000):
    w = synth.make_window(N=30, F=F, seed=0, flags=synth.Flags(use_larvio=1))
    upd.upload(w)
    for _ in range(10):
        upd.run_update()
    upd.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        upd.run_update(); upd.sync()
    dt = (time.perf_counter() - t0) / 50 * 1e3
    prof = upd.profile(reps=10)
    out[F] = dict(ms_per_update=round(dt, 4), kernels_us={k: round(v * 1e3, 1) for k, v in prof.items()})
print(os.environ.get('ORCVIO_SPLIT_TRACKS', 'default'), json.dumps(out))

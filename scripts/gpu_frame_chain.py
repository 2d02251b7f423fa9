"""Timing of the config-3 frame in one call, plain and with ORCVIO_FRAME_CHAIN=1 (the object solve chained to the feature update's
prior factor and M, on a stream of its own): host-visible ms per frame, median of 200, each process one mode.
usage: python scripts/gpu_frame_chain.py   (runs itself twice)"""
import os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    from orcvio_amd import capi, synth
    win = synth.config_window(2)
    win = synth.make_window(N=30, F=400, seed=0, flags=synth.Flags(use_larvio=0, use_left_perturbation=0))
    objs = synth.make_objects(win, n_objects=20, seed=2, sigma_kp=0.004)
    upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    upd.cov_set(win.P)
    ts = []
    for it in range(260):
        upd.cov_set(win.P)
        upd.sync()
        t0 = time.perf_counter()
        f, o = upd.update_frame(win, win.flags, objs, win.R_b2c[0], win.t_c_b[0], True, False, 0)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[60:]) * 1e3
    print(json.dumps(dict(mode=os.environ.get('ORCVIO_FRAME_CHAIN', '0'), median_ms=float(np.median(ts)), p95_ms=float(np.percentile(ts, 95)), accept=int(o['accept']),
                          dx=float(np.linalg.norm(o['dx'])))))
else:
    for m in ('0', '1'):
        env = dict(os.environ, ORCVIO_FRAME_CHAIN=m)
        print(subprocess.run([sys.executable, __file__, 'run'], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])

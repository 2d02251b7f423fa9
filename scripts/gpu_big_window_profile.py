import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orcvio_amd import capi, synth
for N in (33, 34, 38, 48, 60):
    big = capi.MsckfUpdater(device=0, max_clones=60, max_features=512, max_observations=16384)
    w = synth.make_window(N=N, F=400, seed=3, flags=synth.Flags(use_larvio=1), track_len=(20, 30))
    big.upload(w); big.run_update(); big.sync()
    p = big.profile(reps=10)
    print(N, w.n, {k: round(v*1e3,1) for k,v in p.items()}, 'sum', round(sum(p.values())*1e3,1))
    big.close()

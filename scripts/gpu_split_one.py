"""One track count through the tracks front end (for rocprofv3 --kernel-trace --stats): F from argv, ORCVIO_SPLIT_TRACKS from the env."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orcvio_amd import synth, capi
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
upd = capi.MsckfUpdater(max_clones=32, max_features=4096, max_observations=131072)
w = synth.make_window(N=30, F=F, seed=0, flags=synth.Flags(use_larvio=1))
upd.upload(w)
for _ in range(60):
    upd.run_update(); upd.sync()

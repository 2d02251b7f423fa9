import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orcvio_amd import synth, capi
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
upd.upload(synth.config_window(2))
upd.run_update(); upd.sync()
print({k: round(v*1e3,1) for k, v in upd.profile(50).items()})

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from orcvio_amd import capi, synth
from oracle import mirror, mirror_hybrid as mh
from test_gpu_hybrid import compact_rows
idp = 3
w0 = synth.make_window(N=9, F=40, seed=37, track_len=(3, 9), flags=synth.Flags(use_larvio=1))
slam = synth.make_slam_features(w0, 6, seed=idp, outlier_frac=0.25)
w = synth.with_extra_states(w0, idp * len(slam), seed=7)
ref = mh.hybrid_update(w, slam, idp)
upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536)
for mode in ('msckf-only', 'with-ekf'):
    upd.set_extra_states(w.n_extra); upd.set_ekf_rows_mode(True)
    upd.upload(w)
    if mode == 'with-ekf':
        He, Ha, Hx, Hf, r = compact_rows(w, slam, idp)
        upd.upload_ekf_rows(idp, [f.anchor for f in slam], [f.state for f in slam], list(range(len(slam))), He, Ha, Hx, Hf, r, z_vel=np.array([f.z_vel for f in slam]))
    upd.run_update(); upd.sync()
    A = capi.debug_read(upd, 'A')
    d = capi.debug_read(upd, 'dims')
    NA = d['NA']
    base = mirror.msckf_update(w)
    blocks = [b for b, a in zip(base['blocks'], base['accept']) if a]
    rs = [b for b, a in zip(base['rs'], base['accept']) if a]
    if mode == 'with-ekf':
        for (H, r_), a in zip(ref['ekf_rows'], ref['ekf_accept']):
            if a: blocks.append(H); rs.append(r_)
        g, acc = upd.download_ekf(); print('ekf gamma', g, ref['ekf_gamma'], acc, ref['ekf_accept'])
    H = np.vstack(blocks); r = np.concatenate(rs)
    Ha_ = H[:, 15:15 + NA]
    Aexp = Ha_.T @ Ha_; bexp = Ha_.T @ r
    print(mode, d, 'A err', np.abs(A[:NA, :NA] - Aexp).max(), np.abs(Aexp).max(), 'b err', np.abs(A[NA, :NA] - bexp).max(), np.isnan(A).sum())
    bad = np.argwhere(np.abs(A[:NA, :NA] - Aexp) > 1e-6 * np.abs(Aexp).max())
    print('bad entries', len(bad), bad[:10])

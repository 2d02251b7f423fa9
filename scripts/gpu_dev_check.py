"""Development check run on the GPU box: stage-by-stage comparison with the oracle."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import synth, capi
from oracle import oracle, mirror


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def check(win, name, upd, detail=False):
    o = oracle.msckf_update(win)
    g = upd.update_features(win, want_G=True, want_thin=True, want_K=True)
    fin = np.isfinite(o['gamma'])
    line = dict(name=name, acc_equal=bool(np.array_equal(o['accept'], g['accept'])), n_acc=int(g['accept'].sum()),
                gamma=rel(g['gamma'][fin], o['gamma'][fin]), dx=rel(g['dx'], o['dx']), P=rel(g['P_new'], o['P_new']),
                dP=rel(g['P_new'] - win.P, o['P_new'] - win.P), G=rel(g['G'], o['G']), stats=g['stats'].tolist())
    print(json.dumps(line), flush=True)
    if detail:
        d = capi.debug_read(upd, 'dims')
        Hs = capi.debug_read(upd, 'Hs')
        NA = d['NA']
        # block Gram invariants per feature
        bp = o['block_ptr']
        worst = 0.0
        for j in range(win.F):
            if not o['accept'][j]:
                continue
            Ho = o['H_all'][bp[j]:bp[j + 1], 15:]
            ro = o['r_all'][bp[j]:bp[j + 1]]
            Hg = Hs[bp[j]:bp[j + 1], :NA]
            rg = Hs[bp[j]:bp[j + 1], NA]
            worst = max(worst, rel(Hg.T @ Hg, Ho.T @ Ho), rel(Hg.T @ rg, Ho.T @ ro))
        print('  block gram worst rel', worst)
        A = capi.debug_read(upd, 'A')
        Hacc = o['H_all'][:, 15:]
        mask = np.concatenate([np.full(bp[j + 1] - bp[j], bool(o['accept'][j])) for j in range(win.F)]) if win.F else np.zeros(0, bool)
        X = np.hstack([Hacc[mask], o['r_all'][mask][:, None]])
        Gref = X.T @ X
        print('  Gram block vs oracle', rel(A[:NA + 1, :NA + 1], Gref))
        n = d['n']
        RP = capi.debug_read(upd, 'RP')[:n, :n]
        if d['reg_path']:
            print('  R_P^T R_P vs P', rel(np.triu(RP).T @ np.triu(RP), win.P), 'lower part max', np.abs(np.tril(RP, -1)).max())
            Lf = np.triu(RP).T
        else:
            Lf = np.tril(RP)
            print('  Lf Lf^T vs P', rel(Lf @ Lf.T, win.P))
        La = Lf[15:, :]
        s2 = win.flags.noise_feature ** 2
        Mref = s2 * np.eye(n) + La.T @ Gref[:NA, :NA] @ La
        M = capi.debug_read(upd, 'M')[:n, :n]
        print('  M (upper) vs ref', rel(np.triu(M), np.triu(Mref)))
        RM = capi.debug_read(upd, 'RM')[:n, :n]
        if d['reg_path']:
            print('  R_M^T R_M vs M', rel(np.triu(RM).T @ np.triu(RM), Mref))
    return line


def main():
    upd = capi.MsckfUpdater(max_clones=32, max_features=2048, max_observations=65536, debug_hooks=True)   # diagnostics build: orcvio_msckf_debug_* hooks
    upd.set_materialize_stack(True)
    ok = True
    cases = []
    for fl in [synth.Flags(use_larvio=1), synth.Flags(use_larvio=0, use_left_perturbation=0),
               synth.Flags(use_larvio=0, use_left_perturbation=1), synth.Flags(use_larvio=1, if_fej=1, estimate_td=1)]:
        cases.append((synth.make_window(N=5, F=8, seed=1, track_len=(3, 5), flags=fl), f'small l{fl.use_larvio} left{fl.use_left_perturbation} fej{fl.if_fej}'))
    cases.append((synth.make_window(N=12, F=40, seed=2, track_len=(2, 12), outlier_frac=0.2), 'ragged+outliers'))
    cases.append((synth.config_window(1), 'config1'))
    for w, nm in cases:
        try:
            check(w, nm, upd, detail=True)
        except Exception as e:  # noqa
            print('FAILED', nm, repr(e), flush=True)
            ok = False
    w2 = synth.config_window(2)
    t0 = time.time()
    line = check(w2, 'config2', upd, detail=True)
    print('config2 total check s', time.time() - t0)
    upd.set_materialize_stack(False)
    upd.upload(w2)
    for _ in range(3):
        upd.run_update(); upd.sync()
    t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        upd.run_update()
    upd.sync()
    dt = (time.perf_counter() - t0) / K
    print('config2 ms/update (device, back-to-back)', dt * 1e3)
    print('profile', json.dumps(upd.profile(20)))


if __name__ == '__main__':
    main()

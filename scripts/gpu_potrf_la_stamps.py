"""Diagnostic: k_potrf_solve (one workgroup holds the trailing matrix) against k_potrf_solve_la (far workgroups, look-ahead LA) on an
SPD matrix of config 2's size (n = 187, 203 right-hand sides): results against numpy, launch time, and the core-clock stamps of the
chain wave, worker 0, the publisher and one far workgroup (layout: potrf_lookahead.hpp)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orcvio_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 187
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 203
upd = capi.MsckfUpdater(max_clones=32, max_features=64, max_observations=1024, debug_hooks=True)
rng = np.random.default_rng(5)
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
X = (Q * np.logspace(0, -5, n)) @ Q.T
X = 0.5 * (X + X.T)
B = rng.standard_normal((n, nrhs))
Lr = np.linalg.cholesky(X)
Zr = np.linalg.solve(Lr, B)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
out = {}
for la in (0, 2, 3):
    r = capi.debug_potrf_solve(upd, X, B, la=la, stamps=(la != 0), reps=200)
    print(f'la {la}: L err {rel(r["L"], Lr):.2e}  Z err {rel(r["Z"], Zr):.2e}  info {r["info"].tolist()}  {r["us"]:.2f} us per launch (with its two clears)')
    out[la] = dict(L=rel(r['L'], Lr), Z=rel(r['Z'], Zr), us=r['us'])
    if la == 0:
        continue
    st = r['stamps']
    nb = (n + 15) // 16
    w0, w1, w4, wf = st[0:64], st[64:128], st[128:192], st[192:256]
    t0 = w0[0]
    print(f'  wall {(st[257] - st[256]) * 10e-3:.1f} us; chain: tile 0 in LDS {w0[1] - t0}, first sweep {w0[2] - w0[1]}; end {w0[63] - t0} cycles')
    print('  kb | A(at)  step  | flag raised (after A) | every worker: products done (cycles after A)')
    for kb in range(nb):
        a = w0[3 + 3 * kb]
        nxt = w0[3 + 3 * (kb + 1)] if kb + 1 < nb else w0[63]
        print(f'  {kb:2d} | {a - t0:7d} {nxt - a:6d} | {w4[2 * kb + 1] - a:6d} | ' + '  '.join(f'{(int(st[320 + 16 * w + kb]) - a):5d}' for w in range(6)))
    print('  worker 5 (second wavefront of SIMD 1), cycles after A: panel published / older panels done / fetch issued / count B passed / done')
    for kb in range(nb):
        a = w0[3 + 3 * kb]
        print(f'  {kb:2d} | ' + ' '.join(f'{int(w1[q + 4 * kb]) - a:6d}' for q in (1, 2, 3, 4)) + f' {int(st[320 + 16 * 5 + kb]) - a:6d}')
    fa = min(la + 4, nb - 2)
    pl = fa - 1 - la
    print(f'  far workgroup of row {fa} (its clock; start {wf[0]}): ' + ', '.join(f'p{p}: seen +{wf[1 + 3 * p] - wf[0]} done +{wf[2 + 3 * p] - wf[0]}' for p in range(pl + 1)) + f', handed over +{wf[3 * (pl + 1) + 1] - wf[0]}')
print(json.dumps(out))
upd.close()

"""Conditioning sweep (VERDICT r1, next #3d): how far is the device from the reference FORMULA, evaluated in 60-digit
arithmetic (oracle/mp_reference.py -- independent of both double restatements and of every factorisation), as the problem
is pushed: sigma from 1 down to 1e-4, the prior scaled from 1e-6 to 1e2, features pushed out to a kilometre (H_f nearly rank
deficient).  The device compresses in Gram form (A = X^T X - T3^T T3) and solves in square-root form; the reference QR-compresses
and solves S K^T = H P with an LDL^T.  Both lose digits as cond(S) ~ |H|^2 |P| / sigma^2 grows; the question is whether the
device leaves the 1e-6 tolerance BEFORE the reference's own double arithmetic does.  The table goes to
gpurun_out/conditioning.json (committed as profiles/r2_conditioning.json, quoted in DESIGN.md)."""
import dataclasses
import json
import os

import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import mirror, oracle, mp_reference
from helpers import rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=8, max_features=64, max_observations=1024)
    yield u
    u.close()


def _case(sigma, pscale, depth, seed):
    flags = synth.Flags(use_larvio=1, noise_feature=sigma)
    w = synth.make_window(N=5, F=12, seed=seed, track_len=(3, 5), flags=flags, depth=depth)
    return dataclasses.replace(w, P=np.ascontiguousarray(w.P * pscale))


def _merge_json(path, key, value):
    d = {}
    if os.path.exists(path):
        try:
            d = json.load(open(path))
            if 'rows' in d:   # (round 2's single-sweep layout)
                d = {}
        except Exception:
            d = {}
    d[key] = value
    with open(path, 'w') as f:
        json.dump(d, f, indent=1)


def test_conditioning_sweep(upd):
    rows = []
    worst_margin = 0.0
    for depth in [(4.0, 12.0), (200.0, 1000.0)]:
        for sigma in [1.0, 8e-3, 1e-3, 1e-4]:
            for pscale in [1e-6, 1e-4, 1e-2, 1.0, 1e2]:
                w = _case(sigma, pscale, depth, seed=int(-np.log10(sigma) * 10 + np.log10(pscale) + 20))
                ref = mp_reference.msckf_update_mp(w)
                if ref['accept'].sum() == 0:
                    continue
                got = upd.update_features(w)
                mir = mirror.msckf_update(w)
                cor = oracle.msckf_update(w, want_blocks=False, want_K=False)
                same_mask = bool(np.array_equal(got['accept'], ref['accept']))
                # cond(S) of the stacked update, from the double restatement (documentation only)
                H = mir.get('H')
                condS = float(np.linalg.cond(H @ w.P @ H.T + sigma ** 2 * np.eye(H.shape[0]))) if H is not None else float('nan')
                dP_ref = ref['P_new'] - w.P
                e = dict(depth=depth, sigma=sigma, pscale=pscale, condS=condS, same_mask=same_mask,
                         dev_dx=rel(got['dx'], ref['dx']), mir_dx=rel(mir['dx'], ref['dx']), c_dx=rel(cor['dx'], ref['dx']),
                         dev_P=rel(got['P_new'], ref['P_new']), mir_P=rel(mir['P_new'], ref['P_new']),
                         dev_dP=rel(got['P_new'] - w.P, dP_ref), mir_dP=rel(mir['P_new'] - w.P, dP_ref))
                rows.append(e)
                # the bar: inside 1e-6 wherever the reference's own double arithmetic is (with a decade of margin), and never
                # more than two decades behind it outside
                ref_err = max(e['mir_dx'], e['c_dx'])
                if same_mask:
                    if ref_err < 1e-7:
                        assert e['dev_dx'] < 1e-6 and e['dev_P'] < 1e-6, e
                    else:
                        assert e['dev_dx'] < 100 * ref_err, e
                    worst_margin = max(worst_margin, e['dev_dx'] / max(ref_err, 1e-16))
    assert len(rows) >= 30
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    _merge_json(os.path.join(out, 'conditioning.json'), 'features',
                dict(what='relative Frobenius distance to the 60-digit evaluation of the reference update FROM THE POSES (per-observation '
                          'blocks restated in mp arithmetic and checked against 60-digit central differences, tests/test_oracle_mp.py); '
                          'dev = device, mir = numpy restatement, c = C restatement (both double)', rows=rows, worst_dev_over_ref=worst_margin))
    # where the device is inside the north-star tolerance
    inside = [r for r in rows if r['dev_dx'] < 1e-6]
    assert len(inside) >= len(rows) // 2



def test_conditioning_sweep_object_update(built):
    """VERDICT r2 'next' 6c: the object update (removeLostObjects on the reference's own one_car frames, cond(H_f) ~ 1e8, and on a
    synthetic car) against the 50-digit evaluation, sigma and the prior's scale swept.  The device projects through a structured
    Householder QR, the double restatement through a full-U SVD."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import objects_update_reference, object_rows_reference
    from test_oracle_mp import _one_car_blocks
    u = capi.MsckfUpdater(device=0, max_clones=8, max_features=64, max_observations=1024)
    rows = []
    try:
        for name in ('one_car', 'synthetic'):
            for sigma in (0.05, 8e-3):
                for pscale in (1e-4, 1.0, 1e2):
                    flags = synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=sigma)
                    win = synth.make_window(N=4, F=2, seed=3, flags=flags, track_len=2)
                    win = dataclasses.replace(win, P=np.ascontiguousarray(win.P * pscale))
                    if name == 'one_car':
                        obj, blocks = _one_car_blocks(win, [0, 12, 24, 36])
                    else:
                        obj = synth.make_objects(win, n_objects=1, seed=2, sigma_kp=sigma)[0]
                        Hx, Hf, r, rc, hx6 = object_rows_reference(win, obj, True, False, 0)
                        blocks = [(Hx, Hf, r)]
                    ref = mp_reference.objects_update_mp(blocks, win.P, sigma)
                    dbl = objects_update_reference(win, [obj], win.P, True, False, 0)
                    got = u.update_object_tracks(flags, win.N, [obj], win.P, win.R_b2c[0], win.t_c_b[0], True, False, 0)
                    refined = u.objects_refined()
                    # three-launch pipeline: only objects above |R|_F |R^-1|_F = 3e6 take the explicit basis (3.6e8 on the real car, 2.4e5 on
                    # the synthetic one); the one-launch compression (k_obj_fused, round 5) projects EVERY object through it
                    assert refined == (1 if (name == 'one_car' or u.counters()['obj_fused']) else 0)
                    u.set_object_refine(0)   # the fast route alone (round 3), for the record
                    try:
                        fast = u.update_object_tracks(flags, win.N, [obj], win.P, win.R_b2c[0], win.t_c_b[0], True, False, 0)
                    finally:
                        u.set_object_refine(1)
                    e = dict(object=name, sigma=sigma, pscale=pscale, cond_Hf=float(np.linalg.cond(blocks[0][1])), gamma=ref['gamma'], dof=ref['dof_ref'],
                             accept=ref['accept_ref'], same_decision=bool(got['accept'] == ref['accept_ref']),
                             dev_gamma=abs(got['gamma'] - ref['gamma']) / abs(ref['gamma']), dbl_gamma=abs(dbl['gamma'] - ref['gamma']) / abs(ref['gamma']))
                    if ref['accept_ref'] and got['accept']:
                        e.update(dev_dx=rel(got['dx'], ref['dx']), dbl_dx=rel(dbl['dx'], ref['dx']), dev_P=rel(got['P_new'], ref['P_new']),
                                 dbl_P=rel(dbl['P_new'], ref['P_new']), refined=refined,
                                 fast_dx=rel(fast['dx'], ref['dx']) if fast['accept'] else None)
                        # Round 4 (VERDICT r3 'next' 1): the bar is north_star's 1e-6 for BOTH objects.  The reference's real car seen in
                        # four frames has cond(H_f) = 3e8; the semi-normal equations Y = R^-T (H_f^T X) lose cond * eps there (1.4e-6 in
                        # delta_x at P x 100, round 3) -- ill-conditioned objects now take Y from an explicit basis (k_obj_refine).
                        # VERDICT r4 'next' 4: assert the bar that is reached.  1e-6 outright wherever the reference's own double arithmetic
                        # (the restatement) is within 1e-8 of the 50-digit result; the 100 x slack only where the restatement itself is worse.
                        lim = 1e-6 if e['dbl_dx'] < 1e-8 else max(1e-6, 100 * e['dbl_dx'])
                        assert e['dev_dx'] < lim and e['dev_P'] < lim, e
                        e['inside_1e-6'] = bool(e['dev_dx'] < 1e-6)
                        assert e['inside_1e-6'], e   # every row of the object sweep (round 4: worst 8.6e-10)
                    assert e['dev_gamma'] < max(1e-6, 100 * e['dbl_gamma']), e
                    rows.append(e)
    finally:
        u.close()
    assert sum(1 for e in rows if 'dev_dx' in e) >= 6
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    _merge_json(os.path.join(out, 'conditioning.json'), 'objects',
                dict(what='object update: relative distance to the 50-digit evaluation (projection onto the left null space of H_f by an mp '
                          'QR, gate, update); dev = device, dbl = the double restatement (full-U SVD projection)', rows=rows))


def test_conditioning_sweep_hybrid_frame(built):
    """... and one hybrid frame (MSCKF tracks + the rows of in-state features, src/orcvio.cpp:1766-1950 with sz_new = 0): the device
    against the 60-digit evaluation, sigma and the prior's scale swept."""
    from oracle import mirror_hybrid as mh
    u = capi.MsckfUpdater(device=0, max_clones=8, max_features=64, max_observations=1024)
    rows = []
    try:
        for idp in (1, 3):
            for sigma in (8e-3, 1e-3):
                for pscale in (1e-4, 1e-2, 1.0, 1e2):
                    fl = synth.Flags(use_larvio=1, estimate_td=1, noise_feature=sigma)
                    w0 = synth.make_window(N=5, F=8, seed=7, track_len=(3, 5), flags=fl)
                    slam = synth.make_slam_features(w0, 4, seed=5)
                    w = synth.with_extra_states(w0, idp * len(slam), seed=4)
                    w = dataclasses.replace(w, P=np.ascontiguousarray(w.P * pscale))
                    ekf = [mh.feature_jacobian_ekf(w, ft, i, idp) for i, ft in enumerate(slam)]
                    ref = mp_reference.hybrid_update_mp(w, ekf)
                    dbl = mh.hybrid_update(w, slam, idp)
                    u.set_extra_states(w.n_extra)
                    u.set_ekf_rows_mode(True)
                    try:
                        u.upload(w)
                        u.upload_slam_features(idp, slam)
                        u.run_update()
                        u.sync()
                        got = u.download()
                        _, ea = u.download_ekf()
                    finally:
                        u.set_ekf_rows_mode(False)
                        u.set_extra_states(0)
                    same = bool(np.array_equal(got['accept'], ref['accept']) and np.array_equal(ea, ref['ekf_accept']))
                    e = dict(idp=idp, sigma=sigma, pscale=pscale, same_masks=same, rows_in=int(ref['accept'].sum()), ekf_in=int(ref['ekf_accept'].sum()))
                    if same and (ref['accept'].sum() + ref['ekf_accept'].sum()) > 0:
                        e.update(dev_dx=rel(got['dx'], ref['dx']), dbl_dx=rel(dbl['dx'], ref['dx']), dev_P=rel(got['P_new'], ref['P_new']),
                                 dbl_P=rel(dbl['P_new'], ref['P_new']))
                        lim = 1e-6 if e['dbl_dx'] < 1e-8 else max(1e-6, 100 * e['dbl_dx'])
                        assert e['dev_dx'] < lim and e['dev_P'] < lim, e
                    rows.append(e)
    finally:
        u.close()
    assert sum(1 for e in rows if 'dev_dx' in e) >= 8
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    _merge_json(os.path.join(out, 'conditioning.json'), 'hybrid',
                dict(what='hybrid frame (MSCKF tracks + rows of in-state features): relative distance to the 60-digit evaluation; dev = device, '
                          'dbl = the double restatement (oracle/mirror_hybrid.py)', rows=rows))

"""Randomised soak of the sharded entry points with SEVERAL RANKS ON ONE GPU (ORCVIO_COMM_TRANSPORT=ipc, csrc/capi_ipc.inc): the parent
starts `world` fresh child processes (before anything touches the GPU); every child runs the same seeded sequence of random windows
-- feature updates (one-shot and queued staged form, ragged / scattered tracks, outliers, three Jacobian conventions, resident prior)
and object updates (random cars dealt round-robin) -- through orcvio_msckf_update_features_sharded / _run_update_sharded /
_update_object_tracks_sharded on ITS share and compares the joint result with the single-call oracle on the full window.
usage: python scripts/gpu_soak_ipc.py [seconds] [first_seed] [world]      (child: ... --rank R --uid HEX)"""
import dataclasses, hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def child(budget, seed0, world, rank, uid):
    import numpy as np
    from orcvio_amd import capi, sharding, synth
    from oracle import oracle
    from helpers import rel, objects_update_reference, scatter_tracks, random_object_case
    u = capi.MsckfUpdater(device=0, max_clones=32, max_features=2048, max_observations=65536)
    u.comm_init(uid, rank, world)
    fails, worst, n_feat, n_obj, n_staged, digests = [], dict(dx=0.0, P=0.0), 0, 0, 0, hashlib.sha256()
    t_end = time.time() + budget
    seed = seed0
    while True:
        # every rank takes the same decision to go on (the slowest clock decides)
        go = u.comm_allreduce_max([1.0 if time.time() < t_end else 0.0])[0]   # (1 while ANY rank has time left: all stop after the same window)
        rng = np.random.default_rng(500000 + seed)
        kind = int(rng.integers(0, 4))
        try:
            if kind < 3:
                N = int(rng.integers(4, 31)); F = int(rng.integers(2, 300)) * world
                fl = synth.Flags(use_larvio=int(rng.integers(0, 2)), use_left_perturbation=int(rng.integers(0, 2)), if_fej=int(rng.integers(0, 2)),
                                 estimate_td=int(rng.integers(0, 2)), noise_feature=float(rng.choice([0.008, 0.02])))
                full = synth.make_window(N=N, F=F, seed=seed, track_len=(2, min(N, 10)), flags=fl, outlier_frac=float(rng.choice([0.0, 0.1, 0.5])))
                if rng.random() < 0.4:
                    full = scatter_tracks(full, rng, 1, 6)
                share, _ = sharding.shard_window(full, rank, world)
                ref = oracle.msckf_update(full, want_blocks=False, want_K=False)
                if kind == 2:   # staged, queued: several sharded updates in flight one behind the other
                    u.upload(share)
                    for _ in range(int(rng.integers(1, 5))):
                        u.run_update_sharded()
                    u.sync()
                    got = u.download()
                    n_staged += 1
                else:
                    resident = bool(rng.integers(0, 2))
                    if resident:
                        u.cov_set(full.P)
                    got = u.update_features_sharded(share, resident_cov=resident)
                ed, eP = rel(got['dx'], ref['dx']) if np.any(ref['dx']) else float(np.linalg.norm(got['dx'])), rel(got['P_new'], ref['P_new'])
                n_feat += 1
            else:
                case = random_object_case(seed)
                win, objs = case['win'], case['objs']
                ref = objects_update_reference(win, objs, win.P, case['obj_left'], case['new_bbox'], case['vio_left'], full_nullspace=True)
                got = u.update_object_tracks_sharded(case['flags'], win.N, objs[rank::world], win.P, win.R_b2c[0], win.t_c_b[0], case['obj_left'],
                                                     case['new_bbox'], case['vio_left'])
                ok = got['accept'] == ref['accept']
                ed = rel(got['dx'], ref['dx']) if ref['accept'] else float(np.linalg.norm(got['dx']))
                eP = rel(got['P_new'], ref['P_new']) if ref['accept'] else rel(got['P_new'], win.P)
                if not ok:
                    ed = 1.0
                n_obj += 1
            worst['dx'] = max(worst['dx'], ed); worst['P'] = max(worst['P'], eP)
            if not (ed < 1e-6 and eP < 1e-6):
                fails.append(dict(seed=seed, kind=kind, e_dx=ed, e_P=eP))
            digests.update(np.ascontiguousarray(got['dx']).tobytes())
        except Exception as e:
            fails.append(dict(seed=seed, kind=kind, error=repr(e)[:300]))
            break   # (after an error the ranks' sequence numbers may differ: stop)
        seed += 1
        if go < 0.5:
            break
    u.comm_barrier()
    u.close()
    print('RESULT ' + json.dumps(dict(rank=rank, feature_windows=n_feat, staged=n_staged, object_windows=n_obj, failures=fails, worst=worst,
                                      digest=digests.hexdigest()[:16], last_seed=seed)), flush=True)


def main():
    if '--rank' in sys.argv:
        i = sys.argv.index('--rank')
        child(float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[i + 1]), bytes.fromhex(sys.argv[sys.argv.index('--uid') + 1]))
        return
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    env = dict(os.environ, ORCVIO_COMM_TRANSPORT='ipc', ORCVIO_COMM_TIMEOUT_S='120', HSA_ENABLE_IPC_MODE_LEGACY='0')
    uid = os.urandom(128).hex()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(budget), str(seed0), str(world), '--rank', str(r), '--uid', uid],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    res = []
    for p in procs:
        so, se = p.communicate(timeout=budget + 600)
        lines = [ln for ln in so.splitlines() if ln.startswith('RESULT ')]
        res.append(json.loads(lines[-1][7:]) if lines else dict(error=(so[-500:] + se[-1500:])))
    same = len({r.get('digest') for r in res}) == 1
    print(json.dumps(dict(world=world, first_seed=seed0, ranks=res, identical_results_on_every_rank=same,
                          failures=sum(len(r.get('failures', [1])) for r in res)), indent=1))


if __name__ == '__main__':
    main()

"""Randomised soak of the STAGED (multi-GPU) form on one device against the oracle: a random window is dealt over 2-8 ranks
(orcvio_amd.sharding, ranks may end up without a track), every rank's share goes through run_local, the compressed blocks are
laid side by side as the all-gather would, run_finish sums and solves; the same for object tracks (objects_local /
objects_finish).  dx, P+ and every rank's accept mask against the one-window oracle.
usage: python scripts/gpu_soak_sharded.py [seconds] [first_seed]"""
import sys, os, json, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from orcvio_amd import capi, synth, sharding
from oracle import oracle
from helpers import rel, scatter_tracks, random_object_case, objects_update_reference, object_rows_reference

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
upd = capi.MsckfUpdater(device=0, max_clones=40, max_features=2048, max_observations=65536)
hip = C.CDLL('libamdhip64.so')
fails, n_feat, n_obj, worst = [], 0, 0, dict(dx=0.0, P=0.0)
t_end = time.time() + budget
seed = seed0


def grab():
    ptr, ne = upd.block_ptr()
    t = torch.empty(ne, dtype=torch.float64, device='cuda:0')
    assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(ne * 8), 3) == 0
    return t


while time.time() < t_end:
    rng = np.random.default_rng(330000 + seed)
    world = int(rng.integers(2, 9))
    par = dict(seed=seed, world=world)
    try:
        if seed % 3 != 2:   # feature tracks
            N = int(rng.integers(2, 41)); F = int(rng.choice([rng.integers(1, 12), rng.integers(12, 300)]))
            variant = int(rng.integers(0, 3))
            flags = synth.Flags(use_larvio=int(variant == 0), use_left_perturbation=int(variant == 1), if_fej=int(rng.integers(0, 2)),
                                estimate_td=int(rng.integers(0, 2)), leg_dim=int(rng.choice([22, 22, 46])))
            lo = int(rng.integers(1, min(N, 6) + 1)); hi = int(rng.integers(lo, min(N, 32) + 1))
            w = synth.make_window(N=N, F=F, seed=seed, track_len=None, flags=flags, outlier_frac=float(rng.choice([0.0, 0.3])), sigma_px=0.008)
            w = scatter_tracks(w, rng, lo, hi)
            par.update(kind='features', N=N, F=F)
            ref = oracle.msckf_update(w, want_blocks=False, want_K=False)
            parts, acc_ok = [], True
            for rank in range(world):
                ws, idx = sharding.shard_window(w, rank, world)
                upd.upload(ws)
                upd.run_local(); upd.sync()
                parts.append(grab())
            gathered = torch.cat(parts); torch.cuda.synchronize()
            upd.run_finish(gathered.data_ptr(), world); upd.sync()
            got = upd.download()
            ed = rel(got['dx'], ref['dx']) if np.linalg.norm(ref['dx']) > 0 else float(np.linalg.norm(got['dx']))
            eP = rel(got['P_new'], ref['P_new'])
            n_feat += 1
        else:               # object tracks
            case = random_object_case(seed)
            win, objs = case['win'], case['objs']
            par.update(kind='objects', **case['par'])
            ref = objects_update_reference(win, objs, win.P, case['obj_left'], case['new_bbox'], case['vio_left'], full_nullspace=True)
            blocks = []
            for ob in objs:
                rows = object_rows_reference(win, ob, case['obj_left'], case['new_bbox'], case['vio_left'])
                if rows is not None:
                    Hx, Hf, r, rc, hx6 = rows
                    blocks.append(dict(row_clone=rc, Hx6=hx6, Hf=Hf, res=r))
            parts, dof = [], 0
            for rank in range(world):
                dof += upd.objects_local(case['flags'], win.N, blocks[rank::world], win.P)
                upd.sync()
                parts.append(grab())
            gathered = torch.cat(parts); torch.cuda.synchronize()
            upd.objects_finish(gathered.data_ptr(), world, dof)
            got = upd.objects_download()
            if got['accept'] != ref['accept']:
                fails.append(dict(par, accept=(got['accept'], ref['accept'])))
            if ref['accept']:
                ed, eP = rel(got['dx'], ref['dx']), rel(got['P_new'], ref['P_new'])
            else:
                ed, eP = float(np.linalg.norm(got['dx'])), rel(got['P_new'], win.P)
            n_obj += 1
        worst['dx'] = max(worst['dx'], ed); worst['P'] = max(worst['P'], eP)
        if not (ed < 1e-6 and eP < 1e-6):
            fails.append(dict(par, e_dx=ed, e_P=eP))
    except Exception as e:
        fails.append(dict(par, error=repr(e)[:300]))
    seed += 1
print(json.dumps(dict(feature_windows=n_feat, object_windows=n_obj, first_seed=seed0, failures=fails, worst=worst), indent=1, default=str))

"""Randomised parity soak of the object update on the GPU box: random windows and object tracks (number of objects, keypoints per
object, frames inside / outside the window, missing keypoints, residual form, perturbation sides, keypoint noise) through
orcvio_msckf_update_object_tracks (the one-launch compression where every track qualifies, the three-launch pipeline otherwise;
every fifth window with bbox-only tracks among its objects) against the numpy mirror (rows of the residual functors + per-object projection onto the whole left
null space, which is the reference's for a full-rank H_f + QR of the stack + update).  usage: python scripts/gpu_soak_objects.py [seconds] [first_seed] [refine]   (refine: 0 / 1 / 2 = ORCVIO_OPT_OBJECT_REFINE, 'mix' = a random mode per window)"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from orcvio_amd import capi, synth
from helpers import rel, objects_update_reference, random_object_case

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
refine = sys.argv[3] if len(sys.argv) > 3 else '1'
n_refined = 0
n_fused = 0
upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=64, max_observations=1024)
fails, n_done, n_acc, n_def, worst = [], 0, 0, 0, dict(dx=0.0, P=0.0, gamma=0.0)
t_end = time.time() + budget
seed = seed0
while time.time() < t_end:
    par = dict(seed=seed)
    try:
        case = random_object_case(seed, bbox_only_frac=0.3 if seed % 5 == 0 else 0.0)
        win, objs, obj_left, new_bbox, vio_left, flags, par = (case[k] for k in ('win', 'objs', 'obj_left', 'new_bbox', 'vio_left', 'flags', 'par'))
        ref = objects_update_reference(win, objs, win.P, obj_left, new_bbox, vio_left, full_nullspace=True)
        n_def += int(ref['rank_deficient'] > 0)
        mode = int(np.random.default_rng(seed).integers(0, 3)) if refine == 'mix' else int(refine)
        upd.set_object_refine(mode)
        par = dict(par, refine_mode=mode)
        resident = case['resident']
        if resident:
            upd.cov_set(win.P)
        got = upd.update_object_tracks(flags, win.N, objs, None if resident else win.P, win.R_b2c[0], win.t_c_b[0], obj_left, new_bbox, vio_left)
        n_refined += upd.objects_refined()
        n_fused += upd.counters()['obj_fused']
        ok = got['accept'] == ref['accept'] and (not ref['accept'] or got['stats'][0] == ref['dof'])
        eg = abs(got['gamma'] - ref['gamma']) / abs(ref['gamma']) if np.isfinite(ref['gamma']) and ref['gamma'] != 0 else 0.0
        if not np.isfinite(ref['gamma']):   # no usable track at all: the call must report 'no update'
            ok = ok and got['accept'] == 0
        if ref['accept']:
            ed, eP = rel(got['dx'], ref['dx']), rel(got['P_new'], ref['P_new'])
            n_acc += 1
        else:
            ed, eP = float(np.linalg.norm(got['dx'])), rel(got['P_new'], win.P)
        worst['dx'] = max(worst['dx'], ed); worst['P'] = max(worst['P'], eP); worst['gamma'] = max(worst['gamma'], eg)
        if not (ok and eg < 1e-6 and ed < 1e-6 and eP < 1e-6):
            fails.append(dict(par, resident=resident, accept=(got['accept'], ref['accept']), dof=(int(got['stats'][0]), ref['dof']), e_gamma=eg, e_dx=ed, e_P=eP))
    except Exception as e:
        fails.append(dict(par, error=repr(e)[:300]))
    n_done += 1
    seed += 1
print(json.dumps(dict(windows=n_done, accepted=n_acc, with_rank_deficient_Hf=n_def, objects_through_the_explicit_basis=n_refined, windows_through_the_one_launch_compression=n_fused, refine=refine, first_seed=seed0, failures=fails, worst=worst), indent=1, default=str))

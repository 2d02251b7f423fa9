import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np
from orcvio_amd import capi, synth
from helpers import rel, objects_update_reference, random_object_case, object_rows_reference
seed = int(sys.argv[1])
case = random_object_case(seed, bbox_only_frac=0.3 if seed % 5 == 0 else 0.0)
win, objs, obj_left, new_bbox, vio_left, flags = (case[k] for k in ('win', 'objs', 'obj_left', 'new_bbox', 'vio_left', 'flags'))
ref = objects_update_reference(win, objs, win.P, obj_left, new_bbox, vio_left, full_nullspace=True)
print('ref gamma', ref['gamma'], 'dof', ref['dof'], 'deficient', ref['rank_deficient'])
for i, ob in enumerate(objs):
    rows = object_rows_reference(win, ob, obj_left, new_bbox, vio_left)
    if rows is None: print(i, 'no rows'); continue
    Hf = rows[1]
    sv = np.linalg.svd(Hf, compute_uv=False)
    print(i, 'K', len(ob.kps), 'Hf', Hf.shape, 'cond %.3e' % (sv[0] / sv[-1] if sv[-1] > 0 else np.inf), 'sv tail', sv[-4:], 'frames', [fr['clone'] for fr in ob.frames])
upd = capi.MsckfUpdater(device=0, max_clones=32, max_features=64, max_observations=1024)
for mode in (0, 1, 2):
    upd.set_object_refine(mode)
    got = upd.update_object_tracks(flags, win.N, objs, win.P, win.R_b2c[0], win.t_c_b[0], obj_left, new_bbox, vio_left)
    print('mode', mode, 'fused', upd.counters()['obj_fused'], 'accept', got['accept'], 'gamma err %.2e' % (abs(got['gamma'] - ref['gamma']) / abs(ref['gamma'])), 'dx err %.2e' % rel(got['dx'], ref['dx']), 'stats', list(got['stats']))
# one object at a time through the fused path
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1
upd.set_object_refine(1)
for i, ob in enumerate(objs):
    r1 = objects_update_reference(win, [ob], win.P, obj_left, new_bbox, vio_left, full_nullspace=True)
    if not r1['blocks'] or (only >= 0 and i != only): continue
    g = upd.update_object_tracks(flags, win.N, [ob], win.P, win.R_b2c[0], win.t_c_b[0], obj_left, new_bbox, vio_left)
    print('object', i, 'fused', upd.counters()['obj_fused'], 'gamma err %.2e' % (abs(g['gamma'] - r1['gamma']) / abs(r1['gamma'])), 'dx err %.2e' % (rel(g['dx'], r1['dx']) if r1['accept'] else -1), 'dropped', g['stats'][7], 'ref deficient', r1['rank_deficient'])

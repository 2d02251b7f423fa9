"""Generates tests/golden/tri_*.npz from the numpy mirror of the reference's triangulation
(oracle/mirror_triangulate.py).  The reference has no test of its own for these functions and cannot be built here
(SURVEY.md 8c): the vectors pin the restatement -- "parity unpinned".  Run from the repo root:
    python scripts/make_golden_tri.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from orcvio_amd import synth  # noqa: E402
from oracle import mirror_triangulate as mt  # noqa: E402


def spoil(w, seed):
    """Makes some tracks fail in each of the reference's ways: a mismatched observation (large cost), a mirrored one
    (solution behind a camera), tracks of two neighbouring frames (too little motion)."""
    rng = np.random.default_rng(seed)
    z = w.obs_z.copy()
    for j in range(w.F):
        lo, hi = int(w.obs_ptr[j]), int(w.obs_ptr[j + 1])
        u = rng.random()
        if u < 0.15:
            z[lo + (hi - lo) // 2] += rng.uniform(0.2, 0.4, 2)
        elif u < 0.3:   # observations of a point BEHIND the cameras: the fit lands at a negative depth
            cl = w.obs_clone[lo:hi]
            Rm, tm = mt.cam_pose(w.R_b2w[cl[0]], w.t_b_w[cl[0]], w.R_b2c[cl[0]], w.t_c_b[cl[0]])
            pw = Rm @ np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), -rng.uniform(4, 9)]) + tm
            for k, c in enumerate(cl):
                Rc, tc = mt.cam_pose(w.R_b2w[c], w.t_b_w[c], w.R_b2c[c], w.t_c_b[c])
                pc = Rc.T @ (pw - tc)
                z[lo + k] = pc[:2] / pc[2] + 1e-3 * rng.standard_normal(2)
    return synth.Window(**{**w.__dict__, 'obs_z': z})


CASES = {
    'short': dict(N=6, F=24, seed=21, track_len=(2, 5)),
    'full': dict(N=12, F=16, seed=22, track_len=None),
    'spoiled': dict(N=10, F=40, seed=23, track_len=(3, 8), spoil=True),
    'prior': dict(N=8, F=20, seed=24, track_len=(3, 6), prior=True),
}


def main():
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    for name, kw in CASES.items():
        kw = dict(kw)
        do_spoil = kw.pop('spoil', False)
        prior = kw.pop('prior', False)
        w = synth.make_window(flags=synth.Flags(), **kw)
        if do_spoil:
            w = spoil(w, kw['seed'])
        ini = None
        if prior:
            ini = (np.arange(w.F) % 2).astype(np.int32)
        r = mt.triangulate_tracks(w, is_initialized=ini)
        np.savez_compressed(os.path.join(out_dir, f'tri_{name}.npz'), R_b2w=w.R_b2w, t_b_w=w.t_b_w, R_b2c=w.R_b2c, t_c_b=w.t_c_b,
                            p_w=w.p_w, obs_ptr=w.obs_ptr, obs_clone=w.obs_clone, obs_z=w.obs_z,
                            is_initialized=np.zeros(0, np.int32) if ini is None else ini,
                            exp_valid=r['valid'], exp_p_w=r['p_w'], exp_solution=r['solution'], exp_flags=r['flags'],
                            exp_cost=r['cost'], exp_motion=r['motion'])
        print(name, 'F', w.F, 'valid', int(r['valid'].sum()), 'flags', np.bincount(r['flags'], minlength=8).tolist())


if __name__ == '__main__':
    main()

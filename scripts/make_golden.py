"""Generates tests/golden/feat_*.npz from the numpy mirror (oracle/mirror.py).

The reference itself cannot run here, so these vectors pin the
restatement, not the reference binary: "parity unpinned" for the feature rows.
Run from the repo root:  python scripts/make_golden.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from orcvio_amd import synth  # noqa: E402
from oracle import mirror  # noqa: E402

CASES = {
    'larvio': dict(flags=synth.Flags(use_larvio=1), N=4, F=6, seed=11, track_len=(3, 4)),
    'larvio_fej_td': dict(flags=synth.Flags(use_larvio=1, if_fej=1, estimate_td=1), N=5, F=8, seed=12, track_len=(3, 5)),
    'orcvio_right': dict(flags=synth.Flags(use_larvio=0, use_left_perturbation=0), N=4, F=6, seed=13, track_len=(3, 4)),
    'orcvio_left': dict(flags=synth.Flags(use_larvio=0, use_left_perturbation=1), N=5, F=8, seed=14, track_len=(2, 5)),
    'orcvio_left_fej': dict(flags=synth.Flags(use_larvio=0, use_left_perturbation=1, if_fej=1), N=3, F=5, seed=15, track_len=3),
    'outliers': dict(flags=synth.Flags(use_larvio=1), N=6, F=12, seed=16, track_len=(3, 6), outlier_frac=0.4),
}


def window_arrays(w):
    f = w.flags
    return dict(R_b2w=w.R_b2w, t_b_w=w.t_b_w, t_fej=w.t_fej, R_b2c=w.R_b2c, t_c_b=w.t_c_b, p_w=w.p_w,
                obs_ptr=w.obs_ptr, obs_clone=w.obs_clone, obs_z=w.obs_z, obs_zvel=w.obs_zvel, P=w.P,
                flags=np.array([f.leg_dim, f.use_larvio, f.use_left_perturbation, f.if_fej, f.estimate_td,
                                f.discard_large_update], dtype=np.int32),
                noise_feature=np.float64(f.noise_feature), chi2_prob=np.float64(f.chi2_prob))


def main():
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)
    for name, kw in CASES.items():
        w = synth.make_window(**kw)
        res = mirror.msckf_update(w)
        Hx, He, Hf, rr = [], [], [], []
        for j in range(w.F):
            for k in range(w.obs_ptr[j], w.obs_ptr[j + 1]):
                a, b, c, d = mirror.measurement_jacobian_msckf(w, int(w.obs_clone[k]), w.p_w[j], w.obs_z[k])
                Hx.append(a); He.append(b); Hf.append(c); rr.append(d)
        arrs = window_arrays(w)
        arrs.update(exp_Hx=np.array(Hx), exp_He=np.array(He), exp_Hf=np.array(Hf), exp_r=np.array(rr),
                    exp_gamma=res['gamma'], exp_accept=res['accept'], exp_dx=res['dx'], exp_P=res['P_new'],
                    exp_G=res['G'])
        # per-feature basis-invariant block data: H'^T H', H'^T r', |r'|^2
        gram = [b.T @ b for b in res['blocks']]
        arrs['exp_block_gram'] = np.array([g for g in gram]) if gram else np.zeros((0,))
        arrs['exp_block_Htr'] = np.array([b.T @ r for b, r in zip(res['blocks'], res['rs'])])
        arrs['exp_block_rr'] = np.array([r @ r for r in res['rs']])
        np.savez_compressed(os.path.join(out_dir, f'feat_{name}.npz'), **arrs)
        print(name, 'F', w.F, 'accepted', int(res['accept'].sum()))
    # chi-square table (boost quantile stand-in: scipy.stats.chi2.ppf)
    np.savez_compressed(os.path.join(out_dir, 'chi2_095.npz'), table=mirror.chi2_table(0.95, 500),
                        big=np.array([[d, mirror.chi2_threshold(d)] for d in (500, 795, 840, 5000, 16800)]))


if __name__ == '__main__':
    main()

"""GPU parity tests: the HIP path, called through the C-ABI, against the oracle.

Tolerance (BASELINE.json north_star): 1e-6 relative Frobenius on delta_x and on the gain.  The
gain itself is basis dependent (SURVEY.md note N1), so it is compared through G = K*H_thin; the
tests also hold the tighter figures the path actually reaches.  Accept masks must be identical."""
import dataclasses

import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import oracle
from helpers import rel, golden_files, window_from_golden, subset_window, scatter_tracks

pytestmark = pytest.mark.gpu

TOL = 1e-6


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=40, max_features=2048, max_observations=65536)   # the product library, default options
    yield u
    u.close()


@pytest.fixture(scope='module')
def upd_dbg(built):
    """Diagnostics build with the projected stack materialised: the golden test inspects the blocks [H' | r'] (debug_read)."""
    u = capi.MsckfUpdater(device=0, max_clones=40, max_features=2048, max_observations=65536, debug_hooks=True)
    u.set_materialize_stack(True)
    yield u
    u.close()


def _compare(got, ref, win, tol=TOL):
    assert np.array_equal(got['accept'], ref['accept'])
    fin = np.isfinite(ref['gamma'])
    assert np.array_equal(np.isfinite(got['gamma']), fin)
    if fin.any():
        assert rel(got['gamma'][fin], ref['gamma'][fin]) < 1e-9
    assert rel(got['dx'], ref['dx']) < tol
    assert rel(got['P_new'], ref['P_new']) < tol
    dP_ref = ref['P_new'] - win.P
    if np.linalg.norm(dP_ref) > 0:
        assert rel(got['P_new'] - win.P, dP_ref) < tol
    if 'G' in got:
        assert rel(got['G'], ref['G']) < tol
    assert np.array_equal(got['P_new'], got['P_new'].T)


@pytest.mark.parametrize('path', golden_files(), ids=lambda p: p.split('feat_')[-1][:-4])
def test_golden_vectors(upd, upd_dbg, path):
    w, g = window_from_golden(path)
    prod = upd.update_features(w, want_G=True)   # the product library, nothing materialised: the same update
    assert np.array_equal(prod['accept'], g['exp_accept']) and rel(prod['dx'], g['exp_dx']) < TOL and rel(prod['P_new'], g['exp_P']) < TOL
    assert rel(prod['G'], g['exp_G']) < TOL
    upd = upd_dbg
    got = upd.update_features(w, want_G=True, want_K=True, want_thin=True)
    assert np.array_equal(got['accept'], g['exp_accept'])
    assert rel(got['gamma'], g['exp_gamma']) < 1e-9
    assert rel(got['dx'], g['exp_dx']) < TOL
    assert rel(got['P_new'], g['exp_P']) < TOL
    assert rel(got['G'], g['exp_G']) < TOL
    # the returned factors are consistent: G = K H_thin, dx = K r_thin
    assert rel(got["K"] @ got["H_thin"], got["G"]) < 1e-5   # K, H_thin go through chol(A) of a singular Gram block (rank decision)
    assert rel(got["K"] @ got["r_thin"], got["dx"]) < 1e-5
    # projected blocks: basis-invariant Gram data of every accepted block
    Hs = capi.debug_read(upd, 'Hs')
    NA = w.n - 15
    row = 0
    for j in range(w.F):
        M = int(w.obs_ptr[j + 1] - w.obs_ptr[j])
        rho = 2 * M - 3 if M >= 2 else 0
        blk = Hs[row:row + rho]
        row += rho
        if not g['exp_accept'][j]:
            assert not blk.any()
            continue
        H, r = blk[:, :NA], blk[:, NA]
        assert rel(H.T @ H, g['exp_block_gram'][j][15:, 15:]) < 1e-9
        assert rel(H.T @ r, g['exp_block_Htr'][j][15:]) < 1e-8
        assert abs(r @ r - g['exp_block_rr'][j]) < 1e-9 * max(1.0, g['exp_block_rr'][j])


@pytest.mark.parametrize('larvio,left,fej,td', [(1, 0, 0, 0), (1, 0, 1, 1), (0, 0, 0, 0), (0, 1, 0, 0), (0, 1, 1, 1), (0, 0, 1, 0)])
@pytest.mark.parametrize('seed', [0, 1])
def test_flag_variants_ragged(upd, larvio, left, fej, td, seed):
    f = synth.Flags(use_larvio=larvio, use_left_perturbation=left, if_fej=fej, estimate_td=td)
    w = synth.make_window(N=9, F=37, seed=100 + seed, track_len=(2, 9), flags=f, outlier_frac=0.25,
                          estimate_extrin=bool(td))
    ref = oracle.msckf_update(w)
    got = upd.update_features(w, want_G=True)
    assert 0 < ref['accept'].sum() < w.F
    _compare(got, ref, w)


def test_config1_euroc_shape(upd):
    w = synth.config_window(1)
    _compare(upd.update_features(w, want_G=True), oracle.msckf_update(w), w)


def test_config2_full_size(upd):
    """30 clones x 400 features x 30 observations: 22 800 stacked rows."""
    w = synth.config_window(2)
    ref = oracle.msckf_update(w)
    got = upd.update_features(w, want_G=True)
    assert got['stats'][0] == 22800 and got['accept'].all()
    _compare(got, ref, w)
    # size-independent properties: information only shrinks P, and P+ stays PSD
    ev = np.linalg.eigvalsh(got['P_new'])
    assert ev.min() > -1e-12
    assert np.linalg.eigvalsh(w.P - got['P_new']).min() > -1e-10


def test_prune_variant(upd):
    """pruneImuStateBuffer (src/orcvio.cpp:2803-2851): the CSR lists only the removed clones."""
    w = synth.make_window(N=12, F=60, seed=9, track_len=(4, 12))
    ws = subset_window(w, [2, 3])
    mask = np.zeros(w.N, dtype=np.int32)
    mask[[2, 3]] = 1
    ref = oracle.msckf_update(w, clone_mask=mask)
    got = upd.update_features(ws, want_G=True)
    assert ref['accept'].sum() > 0
    _compare(got, ref, ws)


def test_rows_fewer_than_columns(upd):
    """No compression in the reference when rows <= cols (:1664, :2533); same update here."""
    w = synth.make_window(N=10, F=3, seed=4, track_len=(3, 5))
    ref = oracle.msckf_update(w)
    assert ref['stacked_rows'] < w.n
    _compare(upd.update_features(w, want_G=True), ref, w)


def test_short_and_empty_tracks(upd):
    """Tracks with < 2 observations are skipped; an update with no usable track leaves P alone."""
    w = synth.make_window(N=6, F=10, seed=2, track_len=(3, 6))
    ptr = w.obs_ptr.copy()
    # cut feature 0 to one observation and feature 1 to none
    keep = np.ones(len(w.obs_clone), dtype=bool)
    keep[ptr[0] + 1:ptr[1]] = False
    keep[ptr[1]:ptr[2]] = False
    newptr = [0]
    for j in range(w.F):
        newptr.append(newptr[-1] + int(keep[ptr[j]:ptr[j + 1]].sum()))
    w2 = dataclasses.replace(w, obs_ptr=np.asarray(newptr, dtype=np.int32), obs_clone=w.obs_clone[keep].copy(),
                             obs_z=w.obs_z[keep].copy(), obs_zvel=w.obs_zvel[keep].copy())
    ref = oracle.msckf_update(w2)
    got = upd.update_features(w2, want_G=True)
    assert got['accept'][0] == 0 and got['accept'][1] == 0 and np.isnan(got['gamma'][0]) and np.isnan(got['gamma'][1])
    _compare(got, ref, w2)
    # nothing usable at all
    w3 = dataclasses.replace(w, obs_ptr=np.zeros(w.F + 1, dtype=np.int32), obs_clone=np.zeros(0, dtype=np.int32),
                             obs_z=np.zeros((0, 2)), obs_zvel=np.zeros((0, 2)))
    got3 = upd.update_features(w3)
    assert not got3['updated'] and not got3['dx'].any()
    assert rel(got3["P_new"], w.P) < 1e-13
    # F = 0
    w4 = dataclasses.replace(w3, p_w=np.zeros((0, 3)), obs_ptr=np.zeros(1, dtype=np.int32))
    got4 = upd.update_features(w4)
    assert not got4["updated"] and rel(got4["P_new"], w.P) < 1e-13


def test_all_rejected(upd):
    w = synth.make_window(N=6, F=8, seed=6, track_len=(4, 6), outlier_frac=1.0)
    ref = oracle.msckf_update(w)
    assert ref['accept'].sum() == 0
    got = upd.update_features(w)
    assert got['accept'].sum() == 0 and not got['updated']
    assert np.abs(got["dx"]).max() < 1e-300 and rel(got["P_new"], w.P) < 1e-13


def test_max_track_length_and_limits(upd):
    w = synth.make_window(N=32, F=12, seed=8, track_len=32)
    _compare(upd.update_features(w, want_G=True), oracle.msckf_update(w), w)
    w_long = synth.make_window(N=34, F=2, seed=8, track_len=34)
    with pytest.raises(capi.MsckfError) as e:
        upd.update_features(w_long)
    assert e.value.code == 4   # ORCVIO_ERR_TRACK_TOO_LONG


def test_leg_dim_46(upd):
    """calib_imu_instrinsic: 24 more legacy columns (src/orcvio.cpp:196-199)."""
    f = synth.Flags(leg_dim=46)
    w = synth.make_window(N=7, F=30, seed=12, track_len=(3, 7), flags=f)
    _compare(upd.update_features(w, want_G=True), oracle.msckf_update(w), w)


def test_kitti_flags_large_noise(upd):
    """config/kitti_raw.yaml: OrcVIO right perturbation, noise_feature 1, discard flag on."""
    f = synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=1.0, discard_large_update=1)
    w = synth.make_window(N=20, F=200, seed=31, track_len=(3, 6), flags=f, sigma_px=0.008)
    ref = oracle.msckf_update(w)
    got = upd.update_features(w, want_G=True)
    _compare(got, ref, w)
    assert got['stats'][4] == 0


def test_staged_equals_one_shot_and_multi_block_finish(upd):
    """run_local + run_finish (the multi-GPU form) on one device: sharding the tracks in two,
    summing the two compressed blocks, equals the single update."""
    import torch
    w = synth.make_window(N=10, F=64, seed=17, track_len=(3, 10), outlier_frac=0.1)
    one = upd.update_features(w, want_G=True)
    # shard features round-robin into two windows
    blocks = []
    accepts = {}
    for rank in range(2):
        idx = np.arange(rank, w.F, 2)
        ptr = [0]
        sel = []
        for j in idx:
            sel += list(range(w.obs_ptr[j], w.obs_ptr[j + 1]))
            ptr.append(len(sel))
        sel = np.asarray(sel, dtype=np.int64)
        ws = dataclasses.replace(w, p_w=w.p_w[idx].copy(), obs_ptr=np.asarray(ptr, dtype=np.int32),
                                 obs_clone=w.obs_clone[sel].copy(), obs_z=w.obs_z[sel].copy(), obs_zvel=w.obs_zvel[sel].copy())
        upd.upload(ws)
        upd.run_local()
        upd.sync()
        ptr_dev, ne = upd.block_ptr()
        t = torch.empty(ne, dtype=torch.float64, device='cuda:0')
        # device-to-device copy of the compressed block (what the all-gather moves)
        import ctypes as C
        hip = C.CDLL('libamdhip64.so')
        assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr_dev), C.c_size_t(ne * 8), 3) == 0
        blocks.append(t)
        accepts[rank] = idx
    gathered = torch.cat(blocks)
    torch.cuda.synchronize()
    upd.run_finish(gathered.data_ptr(), 2)
    upd.sync()
    got = upd.download()
    assert rel(got['dx'], one['dx']) < 1e-9
    assert rel(got['P_new'], one['P_new']) < 1e-10


@pytest.mark.parametrize('cfg', [1, 2, 5])
def test_fused_solve_equals_two_launch_solve(upd, cfg):
    """ORCVIO_OPT_FUSED_SOLVE: the solver workgroups trailing the factorisation inside one launch give the same Z
    (same MFMA sequence per tile) as k_potrf_reg followed by k_trsm_lds; repeated to catch a stale hand-off."""
    w = synth.config_window(cfg)
    upd.set_fused_solve(False)
    ref = upd.update_features(w)
    upd.set_fused_solve(True)
    for _ in range(20):
        got = upd.update_features(w)
        assert np.array_equal(got['accept'], ref['accept'])
        assert rel(got['dx'], ref['dx']) < 1e-12
        assert rel(got['P_new'], ref['P_new']) < 1e-12


@pytest.mark.parametrize('cfg', [1, 2])
def test_fused_front_equals_forked_front(upd, cfg):
    """ORCVIO_OPT_FUSED_FRONT: the tracks, the compression and chol(P) in one launch (k_front: two feature teams per
    workgroup, device-wide counters between the phases) against the forked form (k_potrf_reg on the side stream,
    k_feature, k_gram_pair, k_assemble_A).  The Gram row chunks differ, so the sums agree to rounding, not bit for bit;
    repeated to catch a stale counter or a missed hand-off."""
    w = synth.config_window(cfg)
    upd.set_fused_front(False)
    ref = upd.update_features(w, want_G=True)
    upd.set_fused_front(True)
    for _ in range(20):
        got = upd.update_features(w, want_G=True)
        assert np.array_equal(got['accept'], ref['accept'])
        assert np.allclose(got['gamma'], ref['gamma'], rtol=1e-12, atol=0, equal_nan=True)
        assert rel(got['dx'], ref['dx']) < 1e-11
        assert rel(got['P_new'], ref['P_new']) < 1e-12
        assert rel(got['G'], ref['G']) < 1e-11


@pytest.mark.parametrize('F', [1, 2, 3, 255, 509, 510, 511, 700])
def test_fused_front_track_counts(upd, F):
    """One team idle (odd F), a single workgroup, the co-residency limit (1 + ceil(F/2) workgroups <= 256 CUs: 510
    tracks) and beyond it (falls back to the forked front): all against the oracle."""
    w = synth.make_window(N=12, F=F, seed=300 + F, track_len=(2, 9))
    _compare(upd.update_features(w, want_G=True), oracle.msckf_update(w), w)


@pytest.mark.parametrize('N,leg', [(31, 22), (33, 22), (26, 46), (29, 46)])
def test_fused_front_wide_windows(upd, N, leg):
    """The widest windows of the register path (n up to 224): four 64-column passes in the feature teams, NA > 192 so
    the assembly stays inside k_front (k_gemm_asmA needs NA <= 192), and the 46-dimensional legacy block."""
    w = synth.make_window(N=N, F=90, seed=400 + N, track_len=(3, min(N, 20)), flags=synth.Flags(use_larvio=1, leg_dim=leg))
    assert w.n <= 224
    _compare(upd.update_features(w, want_G=True), oracle.msckf_update(w), w)


@pytest.mark.parametrize('k', [1, 3, 9, 24])
def test_extra_states_behind_the_clones(upd, k):
    """ORCVIO_OPT_EXTRA_STATES: the covariance carries k more states behind the clones (the EKF-SLAM feature states of
    the hybrid filter); the reference's featureJacobian_msckf builds its rows state_cov.cols() wide with zeros there
    (src/orcvio.cpp:1191-1192) and the update moves those states through the cross-covariances.  Oracle: the literal
    numpy restatement on the wider state."""
    from oracle import mirror
    w = synth.with_extra_states(synth.make_window(N=9, F=40, seed=70 + k, track_len=(3, 9)), k, seed=k)
    assert w.P.shape[0] == 22 + 54 + k
    upd.set_extra_states(k)
    try:
        got = upd.update_features(w, want_G=True)
    finally:
        upd.set_extra_states(0)
    ref = mirror.msckf_update(w)
    _compare(got, ref, w)
    assert np.linalg.norm(ref['dx'][-k:]) > 0   # the extra states do move


def test_large_window_takes_the_lds_panel_path(upd):
    """N = 38 clones: n = 250 > 224, beyond the register-resident factorisations: both factorisations and the solve by 2 x 2 blocks out
    of the register kernels (round 5: capi_update.inc blk2; before: the LDS-panel k_potrf and k_trsm_rl)."""
    w = synth.make_window(N=38, F=60, seed=21, track_len=(3, 12))
    assert w.n > 224
    _compare(upd.update_features(w, want_G=True), oracle.msckf_update(w), w)


@pytest.mark.parametrize('N,F', [(34, 80), (38, 60), (47, 50), (60, 40)])
def test_large_windows_block_factorisation_against_the_lds_panel_kernels(built, monkeypatch, N, F):
    """Windows of 15 .. 26 block steps: the 2 x 2 block factorisation (R11 = chol(X11), W = R11^-T X12, S = X22 - W^T W, R22 = chol(S),
    the solve through the same blocks) against the LDS-panel kernels it replaces (ORCVIO_BLK2=0) and the oracle; a semi-definite prior
    (zero-variance extrinsics) in both leading and trailing block positions; the resident factor of the update before it as the prior."""
    w = synth.make_window(N=N, F=F, seed=100 + N, track_len=(3, min(N, 24)))
    ref = oracle.msckf_update(w)
    monkeypatch.setenv('ORCVIO_BLK2', '0')
    old = capi.MsckfUpdater(device=0, max_clones=60, max_features=256, max_observations=8192, debug_hooks=True)   # (the switch exists in the diagnostics build only)
    monkeypatch.delenv('ORCVIO_BLK2')
    new = capi.MsckfUpdater(device=0, max_clones=60, max_features=256, max_observations=8192)
    try:
        a = old.update_features(w, want_G=True)
        b = new.update_features(w, want_G=True)
        _compare(b, ref, w)
        assert np.array_equal(a['accept'], b['accept'])
        assert rel(b['dx'], a['dx']) < 1e-9 and rel(b['P_new'], a['P_new']) < 1e-10 and rel(b['G'], a['G']) < 1e-8
        # two updates in a row on the resident covariance: the second takes the factor the first committed (kf x kf block solve)
        new.cov_set(w.P)
        r1 = new.update_features(w, resident_cov=True, want_P=True)
        new.cov_commit()
        w2 = dataclasses.replace(w, P=r1['P_new'])
        r2 = new.update_features(w2, resident_cov=True, want_P=True)
        ref2 = oracle.msckf_update(w2)
        assert rel(r1['dx'], ref['dx']) < 1e-6 and rel(r2['dx'], ref2['dx']) < 1e-6 and rel(r2['P_new'], ref2['P_new']) < 1e-6
    finally:
        old.close()
        new.close()


def test_maximum_window_of_60_clones(built):
    """ORCVIO_MAX_CLONES: n = 382, six 64-column passes in k_feature, 24 block steps in the LDS-panel factorisation."""
    u = capi.MsckfUpdater(device=0, max_clones=60, max_features=256, max_observations=8192)
    try:
        w = synth.make_window(N=60, F=40, seed=5, track_len=(3, 20))
        _compare(u.update_features(w, want_G=True), oracle.msckf_update(w), w)
    finally:
        u.close()


@pytest.mark.parametrize('seed', range(12))
def test_random_shapes_and_flags(upd, seed):
    """Randomised sweep over window size, track raggedness, Jacobian variant, FEJ, td, leg_dim, noise and outlier
    rate: every combination must agree with the oracle (catches shape-dependent launch and indexing errors)."""
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(2, 40))
    F = int(rng.integers(1, 120))
    lo = int(rng.integers(2, min(N, 6) + 1))
    hi = int(rng.integers(lo, min(N, 32) + 1))
    variant = int(rng.integers(0, 3))
    flags = synth.Flags(use_larvio=int(variant == 0), use_left_perturbation=int(variant == 1), if_fej=int(rng.integers(0, 2)),
                        estimate_td=int(rng.integers(0, 2)), leg_dim=int(rng.choice([22, 22, 46])),
                        noise_feature=float(rng.choice([0.008, 0.05, 1.0])))
    w = synth.make_window(N=N, F=F, seed=seed, track_len=(lo, hi), flags=flags, outlier_frac=float(rng.choice([0.0, 0.3])),
                          sigma_px=0.008)
    ref = oracle.msckf_update(w)
    got = upd.update_features(w, want_G=True)
    _compare(got, ref, w)


@pytest.mark.parametrize('seed', range(24))
def test_random_scattered_tracks_and_prior_modes(upd, seed):
    """As test_random_shapes_and_flags with NON-contiguous clone lists per track (down to one observation) and the prior taken
    from the caller, from the resident covariance, or from the resident covariance factored ahead of the call -- a bounded
    slice of scripts/gpu_soak.py (2 813 random windows in 300 s on the GPU box, no failure, worst dx error 2e-12)."""
    rng = np.random.default_rng(770000 + seed)
    N = int(rng.integers(2, 41))
    F = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 720)], p=[0.4, 0.45, 0.15]))
    variant = int(rng.integers(0, 3))
    flags = synth.Flags(use_larvio=int(variant == 0), use_left_perturbation=int(variant == 1), if_fej=int(rng.integers(0, 2)),
                        estimate_td=int(rng.integers(0, 2)), leg_dim=int(rng.choice([22, 22, 46])),
                        noise_feature=float(rng.choice([0.008, 0.05, 1.0])))
    lo = int(rng.integers(1, min(N, 6) + 1))
    hi = int(rng.integers(lo, min(N, 32) + 1))
    w = synth.make_window(N=N, F=F, seed=seed, track_len=None, flags=flags, outlier_frac=float(rng.choice([0.0, 0.3])), sigma_px=0.008)
    w = scatter_tracks(w, rng, lo, hi)
    ref = oracle.msckf_update(w)
    mode = seed % 3
    if mode == 0:
        got = upd.update_features(w)
    else:
        upd.cov_set(w.P)
        if mode == 2:
            upd.cov_prefactor()
        got = upd.update_features(w, resident_cov=True)
    _compare(got, ref, w)


def test_observation_order_within_a_track_does_not_matter(upd):
    """The reference lists a track's observations in ascending state id; the device forms only the upper triangle of the gate's
    E = J P J^T and limits the columns of P an observation reads by a PREFIX MAXIMUM of the clone indices -- any order of the
    observations gives the same gate and the same update."""
    w = synth.make_window(N=12, F=80, seed=23, track_len=(3, 12), outlier_frac=0.2)
    rng = np.random.default_rng(1)
    oc, oz, ov = w.obs_clone.copy(), w.obs_z.copy(), w.obs_zvel.copy()
    for j in range(w.F):
        lo, hi = int(w.obs_ptr[j]), int(w.obs_ptr[j + 1])
        perm = lo + rng.permutation(hi - lo)
        oc[lo:hi], oz[lo:hi], ov[lo:hi] = w.obs_clone[perm], w.obs_z[perm], w.obs_zvel[perm]
    ws = dataclasses.replace(w, obs_clone=np.ascontiguousarray(oc), obs_z=np.ascontiguousarray(oz), obs_zvel=np.ascontiguousarray(ov))
    a, b = upd.update_features(w), upd.update_features(ws)
    assert np.array_equal(a['accept'], b['accept']) and 0 < a['accept'].sum() < w.F
    assert rel(b['gamma'], a['gamma']) < 1e-9
    assert rel(b['dx'], a['dx']) < 1e-9 and rel(b['P_new'], a['P_new']) < 1e-9
    ref = oracle.msckf_update(w)
    assert rel(b['dx'], ref['dx']) < TOL and rel(b['P_new'], ref['P_new']) < TOL

"""The zero-copy boundary (orcvio_msckf_io_begin / _io_update: the handle's pinned arena written and read in place, inputs pulled
by the first kernel of the graph, results pushed into host-coherent memory by the last, the caller waiting on a flag word) and
the launch-graph cache behind it, against the oracle -- and the sequence ADVICE r2 (high) describes: more than three same-shape
updates on the resident covariance, each followed by a commit, where a single graph keyed without the address of the
double-buffered square-root factor replayed a stale factor."""
import numpy as np
import pytest

from orcvio_amd import capi, synth
from oracle import oracle
from helpers import rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def upd(built):
    u = capi.MsckfUpdater(device=0, max_clones=32, max_features=1024, max_observations=32768)
    yield u
    u.close()


def _ref(win):
    return oracle.msckf_update(win, want_blocks=False, want_K=False)


@pytest.mark.parametrize('shape', [dict(N=8, F=40, track_len=(3, 8)), dict(N=20, F=150, track_len=(3, 6)), dict(N=30, F=400, track_len=None),
                                   dict(N=30, F=700, track_len=None)])
def test_io_update_equals_the_oracle(upd, shape):
    win = synth.make_window(seed=11, outlier_frac=0.1, **shape)
    ref = _ref(win)
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]))
    upd.io_fill(io, win)
    for it in range(5):   # plain launches, then the captured graph, then its replays
        stats = upd.io_update(want_P=True)
        assert np.array_equal(io['accept'], ref['accept']), it
        assert rel(io['dx'], ref['dx']) < 1e-6 and rel(io['P_out'], ref['P_new']) < 1e-6, it
        assert np.allclose(io['gamma'], ref['gamma'], rtol=1e-8, equal_nan=True)
        assert stats[2] == int(ref['accept'].sum()) and stats[3] == 1
        io['dx'][:] = 0.0; io['P_out'][:] = 0.0; io['accept'][:] = -1   # the next publication must rewrite them
    # the copying call runs on the same machinery
    got = upd.update_features(win, want_G=True)
    assert rel(got['dx'], ref['dx']) < 1e-6 and rel(got['P_new'], ref['P_new']) < 1e-6 and rel(got['G'], ref['G']) < 1e-6


def test_new_inputs_in_the_same_arena(upd):
    """One io_begin, several frames of the same sizes with different contents: the arena is re-read by every update."""
    wins = [synth.make_window(N=12, F=60, seed=s, track_len=8, outlier_frac=0.2) for s in (1, 2, 3, 4)]
    io = upd.io_begin(wins[0].flags, 12, 60, int(wins[0].obs_ptr[-1]))
    for w in wins * 2:
        upd.io_fill(io, w)
        upd.io_update(want_P=True)
        ref = _ref(w)
        assert np.array_equal(io['accept'], ref['accept'])
        assert rel(io['dx'], ref['dx']) < 1e-6 and rel(io['P_out'], ref['P_new']) < 1e-6


def test_arena_contents_are_validated(upd):
    win = synth.make_window(N=6, F=10, seed=1, track_len=(3, 6))
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]))
    upd.io_fill(io, win)
    io['obs_clone'][3] = 99
    with pytest.raises(capi.MsckfError) as e:
        upd.io_update()
    assert e.value.code == 1
    io['obs_clone'][3] = win.obs_clone[3]
    io['obs_ptr'][4] = io['obs_ptr'][3] - 1
    with pytest.raises(capi.MsckfError):
        upd.io_update()
    upd.io_fill(io, win)   # repaired: the same arena works
    upd.io_update()
    assert rel(io['dx'], _ref(win)['dx']) < 1e-6
    with pytest.raises(capi.MsckfError):   # more observations than the handle holds
        upd.io_begin(win.flags, win.N, win.F, 10 ** 7)


@pytest.mark.parametrize('prefactor', [False, True])
@pytest.mark.parametrize('path', ['io', 'copying'])
def test_same_shape_updates_on_the_resident_covariance(upd, prefactor, path):
    """update + commit, six times, same shape, the prior resident -- every update must see the factor the
    previous commit left (it alternates between two buffers) and equal the oracle's chain."""
    N, F = 10, 80
    wins = [synth.make_window(N=N, F=F, seed=100 + k, track_len=N, sigma_px=0.008) for k in range(6)]
    P = wins[0].P.copy()
    upd.cov_set(P)
    if prefactor:
        upd.cov_prefactor()
    if path == 'io':
        io = upd.io_begin(wins[0].flags, N, F, int(wins[0].obs_ptr[-1]), with_P=False)
    for k, w in enumerate(wins):
        w.P[:] = P
        ref = _ref(w)
        if path == 'io':
            upd.io_fill(io, w, with_P=False)
            upd.io_update(want_P=False, commit=True)
            dx, acc = io['dx'].copy(), io['accept'].copy()
        else:
            got = upd.update_features(w, resident_cov=True, want_P=False)
            upd.cov_commit()
            dx, acc = got['dx'], got['accept']
        assert np.array_equal(acc, ref['accept']), k
        assert rel(dx, ref['dx']) < 1e-6, (k, rel(dx, ref['dx']))
        P = ref['P_new']
        assert rel(upd.cov_get(), P) < 1e-6, k


def test_refused_update_is_not_committed(upd):
    """A non-finite prior: the device refuses the update (ORCVIO_ERR_NOT_SPD) and the commit inside the same launch refuses itself."""
    win = synth.make_window(N=8, F=30, seed=5, track_len=8)
    upd.cov_set(win.P)
    good = upd.cov_get()
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=True)
    upd.io_fill(io, win)
    io['P'][20, 20] = np.nan
    with pytest.raises(capi.MsckfError) as e:
        upd.io_update(want_P=True, commit=True)
    assert e.value.code == 6
    assert np.array_equal(upd.cov_get(), good)
    with pytest.raises(capi.MsckfError):
        upd.cov_commit()
    upd.io_fill(io, win)
    upd.io_update(want_P=True, commit=True)
    assert rel(upd.cov_get(), _ref(win)['P_new']) < 1e-6


def test_graph_cache_keeps_several_shapes(upd):
    """Shapes that alternate (the feature update and the prune update of a frame) each keep their captured graph."""
    a = synth.make_window(N=10, F=50, seed=1, track_len=10)
    b = synth.make_window(N=10, F=20, seed=2, track_len=(3, 5))
    ra, rb = _ref(a), _ref(b)
    for _ in range(6):
        ga = upd.update_features(a)
        gb = upd.update_features(b)
        assert rel(ga['dx'], ra['dx']) < 1e-6 and rel(gb['dx'], rb['dx']) < 1e-6


def test_submit_and_collect_equal_io_update(upd):
    """orcvio_msckf_io_submit / _io_collect: the launch and the wait as two calls (the caller's thread is free in between);
    same results as io_update, on a host prior with P+ back and on the resident covariance with the commit inside the launch."""
    import time
    win = synth.make_window(N=12, F=80, seed=21, track_len=(3, 12), outlier_frac=0.1)
    ref = _ref(win)
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=True)
    upd.io_fill(io, win)
    upd.io_submit(want_P=True, commit=False)
    with pytest.raises(capi.MsckfError):
        upd.io_submit(want_P=True, commit=False)   # one submission at a time
    time.sleep(0.002)                              # (the caller's own work)
    stats = upd.io_collect()
    assert stats[3] == 1 and np.array_equal(io['accept'], ref['accept'])
    assert rel(io['dx'], ref['dx']) < 1e-6 and rel(io['P_out'], ref['P_new']) < 1e-6
    with pytest.raises(capi.MsckfError):
        upd.io_collect()                           # nothing submitted
    upd.cov_set(win.P)
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=False)
    upd.io_fill(io, win, with_P=False)
    upd.io_submit(want_P=False, commit=True)
    upd.io_collect()
    assert rel(io['dx'], ref['dx']) < 1e-6 and rel(upd.cov_get(), ref['P_new']) < 1e-6


@pytest.mark.parametrize('shape', [dict(N=20, F=1, track_len=2), dict(N=20, F=5, track_len=2), dict(N=20, F=4, track_len=3), dict(N=12, F=8, track_len=(2, 3)),
                                   dict(N=30, F=3, track_len=(2, 4)), dict(N=32, F=6, track_len=2)])
def test_thin_stack_takes_the_direct_form_and_equals_the_oracle(upd, shape):
    """A stack of at most sixteen projected rows -- pruneImuStateBuffer's update: one row per feature seen in both clones that leave
    (reference src/orcvio.cpp:2803-2851) -- is applied in the reference's own direct form (S = H P H^T + s2 I of dimension m,
    measurementUpdate_msckf without its QR, :1664-1753) instead of the square-root form of dimension n (k_thin_gain / k_thin_apply):
    same gate decisions, dx and P+ against the oracle; the commit leaves the covariance resident for the next update."""
    win = synth.make_window(seed=31, outlier_frac=0.0, **shape)
    ref = _ref(win)
    rows = int(sum(max(2 * int(m) - 3, 0) for m in np.diff(win.obs_ptr)))
    assert 0 < rows <= 16
    upd.cov_set(win.P)
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=False)
    upd.io_fill(io, win, with_P=False)
    stats = upd.io_update(want_P=False, commit=True)
    assert np.array_equal(io['accept'], ref['accept'])
    assert np.allclose(io['gamma'], ref['gamma'], rtol=1e-8, equal_nan=True)
    assert rel(io['dx'], ref['dx']) < 1e-6, rel(io['dx'], ref['dx'])
    P1 = upd.cov_get()
    assert rel(P1, ref['P_new']) < 1e-6 and np.array_equal(P1, P1.T)
    assert stats[2] == int(ref['accept'].sum()) and stats[3] == (1 if ref['accept'].any() else 0)
    # the same update through the general (square-root) path: with P+ sent to the host the thin form is not taken
    upd.cov_set(win.P)
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=False)
    upd.io_fill(io, win, with_P=False)
    upd.io_update(want_P=True, commit=False)
    assert rel(io['dx'], ref['dx']) < 1e-6 and rel(io['P_out'], P1) < 1e-9
    # ... and the handle goes on from the covariance the thin update committed (no factor is resident: the next update factors P itself)
    upd.cov_set(P1)
    win2 = synth.make_window(N=win.N, F=30, seed=32, track_len=(3, 6))
    import dataclasses
    win2 = dataclasses.replace(win2, P=P1)
    io = upd.io_begin(win2.flags, win2.N, win2.F, int(win2.obs_ptr[-1]), with_P=False)
    upd.io_fill(io, win2, with_P=False)
    upd.io_update(want_P=False, commit=True)
    ref2 = _ref(win2)
    assert rel(io['dx'], ref2['dx']) < 1e-6 and rel(upd.cov_get(), ref2['P_new']) < 1e-6


def test_thin_update_refuses_a_non_finite_prior(upd):
    """A NaN in the prior where the gate never looks (an IMU row against an extrinsic column: the gate reads the active block only) but
    the gain does (W = P H^T runs over every row of P): dx comes out non-finite, the launch's last workgroup puts the prior back
    into the spare buffer, the call reports ORCVIO_ERR_NOT_SPD and the resident covariance is the prior, bit for bit."""
    win = synth.make_window(N=20, F=4, seed=33, track_len=2)
    bad = win.P.copy()
    bad[3, 15] = bad[15, 3] = np.nan
    upd.cov_set(bad)
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=False)
    upd.io_fill(io, win, with_P=False)
    with pytest.raises(capi.MsckfError) as e:
        upd.io_update(want_P=False, commit=True)
    assert e.value.code == 6
    assert np.array_equal(upd.cov_get(), bad, equal_nan=True)
    # ... and a NaN the gate does see rejects the tracks: no update, the covariance stays what it was
    bad = win.P.copy()
    bad[25, 25] = np.nan
    upd.cov_set(bad)
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=False)
    upd.io_fill(io, win, with_P=False)
    try:
        stats = upd.io_update(want_P=False, commit=True)
        assert stats[3] == 0 or np.isfinite(io['dx']).all()
    except capi.MsckfError as e2:
        assert e2.code == 6
    got = upd.cov_get()
    assert np.array_equal(np.isnan(got), np.isnan(bad))


def test_explicit_commit_behind_a_thin_update(upd):
    """io_update WITHOUT the commit in the launch, then orcvio_msckf_cov_commit: behind the direct form of a thin stack there is no
    square-root factor to keep (no Z): the commit keeps P+ only and the next update factors it -- against the oracle."""
    import dataclasses
    win = synth.make_window(N=20, F=5, seed=41, track_len=2)
    ref = _ref(win)
    upd.cov_set(win.P)
    io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=False)
    upd.io_fill(io, win, with_P=False)
    upd.io_update(want_P=False, commit=False)
    assert rel(io['dx'], ref['dx']) < 1e-6
    upd.cov_commit()
    P1 = upd.cov_get()
    assert rel(P1, ref['P_new']) < 1e-6
    win2 = dataclasses.replace(synth.make_window(N=20, F=40, seed=42, track_len=(3, 6)), P=P1)
    ref2 = _ref(win2)
    io = upd.io_begin(win2.flags, win2.N, win2.F, int(win2.obs_ptr[-1]), with_P=False)
    upd.io_fill(io, win2, with_P=False)
    upd.io_update(want_P=False, commit=True)
    assert rel(io['dx'], ref2['dx']) < 1e-6 and rel(upd.cov_get(), ref2['P_new']) < 1e-6
    # the copying call with the optional outputs asked for takes the square-root path (they are derived from its factors)
    got = upd.update_features(win, want_G=True)
    assert rel(got['dx'], ref['dx']) < 1e-6 and rel(got['G'], ref['G']) < 1e-6

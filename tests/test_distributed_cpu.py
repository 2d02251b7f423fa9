"""World-size-2 gloo test of the multi-GPU form on CPU: shard the tracks, build each
rank's compressed block [A b; b^T c], all-gather, rank-ordered sum, replicated solve == single update.
The per-rank block and the solve are computed with the oracle here (no GPU); the GPU form of the same
algebra is tests/test_gpu_parity.py::test_staged_equals_one_shot_and_multi_block_finish."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _block_of(win):
    from oracle import mirror
    out = mirror.msckf_update(win)
    NA = win.n - 15
    X = [np.hstack([b[:, 15:], r[:, None]]) for b, r, a in zip(out['blocks'], out['rs'], out['accept']) if a]
    X = np.vstack(X) if X else np.zeros((0, NA + 1))
    return X.T @ X, out['accept']


def _finish(A, P, sigma2):
    """delta_x and P+ from the summed block in the square-root form of DESIGN.md section 3."""
    n = P.shape[0]
    NA = n - 15
    w, V = np.linalg.eigh(P)
    Lf = V * np.sqrt(np.clip(w, 0, None))          # any square root of P works
    La = Lf[15:, :]
    M = sigma2 * np.eye(n) + La.T @ A[:NA, :NA] @ La
    dx = Lf @ np.linalg.solve(M, La.T @ A[:NA, NA])
    Pn = sigma2 * Lf @ np.linalg.solve(M, Lf.T)
    return dx, 0.5 * (Pn + Pn.T)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch
    import torch.distributed as dist
    from orcvio_amd import synth, sharding
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    win = synth.make_window(N=8, F=30, seed=11, track_len=(3, 8), outlier_frac=0.2)
    sub, idx = sharding.shard_window(win, rank, world)
    A, acc = _block_of(sub)
    local = torch.from_numpy(np.ascontiguousarray(A))
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    total = sharding.sum_blocks(gathered).numpy()
    dx, Pn = _finish(total, win.P, win.flags.noise_feature ** 2)
    q.put((rank, idx, acc, dx, Pn))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_update_equals_single_update(built):
    import torch.multiprocessing as mp
    from orcvio_amd import synth, sharding
    from oracle import mirror
    from helpers import rel
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    win = synth.make_window(N=8, F=30, seed=11, track_len=(3, 8), outlier_frac=0.2)
    ref = mirror.msckf_update(win)
    # every feature lands on exactly one rank, accept masks agree with the single run
    seen = np.zeros(win.F, dtype=int)
    for rank, idx, acc, dx, Pn in res:
        seen[idx] += 1
        assert np.array_equal(acc, ref['accept'][idx])
    assert np.all(seen == 1)
    # both ranks hold the same (replicated) result, equal to the single update
    (_, _, _, dx0, P0), (_, _, _, dx1, P1) = sorted(res, key=lambda t: t[0])
    assert np.array_equal(dx0, dx1) and np.array_equal(P0, P1)
    assert rel(dx0, ref['dx']) < 1e-8
    assert rel(P0, ref['P_new']) < 1e-9


def test_dealing_is_balanced_and_complete():
    from orcvio_amd import synth, sharding
    win = synth.make_window(N=10, F=101, seed=3, track_len=(2, 10))
    for world in (2, 4, 8):
        parts = sharding.deal_features(win.obs_ptr, world)
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(win.F))
        M = np.diff(win.obs_ptr)
        rho = np.where(M >= 2, 2 * M - 3, 0)
        loads = np.array([rho[p].sum() for p in parts])
        assert loads.max() - loads.min() <= rho.max()


def _object_worker(rank, world, port, q):
    """Objects dealt round-robin over the ranks: each rank projects ITS objects against their own H_f and forms its
    block; all-gather, rank-ordered sum, joint gate with the summed dof, replicated solve."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch
    import torch.distributed as dist
    from orcvio_amd import synth, sharding
    from oracle import mirror, mirror_objects as mo
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    flags = synth.Flags(use_larvio=0)
    win = synth.make_window(N=10, F=4, seed=2, flags=flags, track_len=4)
    objs = synth.make_objects(win, n_objects=5, seed=9, sigma_kp=0.004)
    NA = win.n - 15
    A = np.zeros((NA + 1, NA + 1))
    dof = 0
    for ob in objs[rank::world]:
        res, Hf, Jc, counts = mo.object_rows(ob.wTo, ob.shape, ob.kps, ob.frames, True, False)
        Hx, Hf2, r, rc, hx6 = mo.construct_object_residual_jacobians(Jc, [fr['clone'] for fr in ob.frames], Hf, res, counts,
                                                                      [fr['wTc'] for fr in ob.frames], win.R_b2c[0], win.t_c_b[0], 0,
                                                                      flags.leg_dim, win.N)
        ok, H1, r1 = mirror.nullspace_project_svd(Hf2, Hx, r)
        X = np.hstack([H1[:, 15:], r1[:, None]])
        A += X.T @ X
        dof += H1.shape[0]
    local = torch.from_numpy(np.ascontiguousarray(A))
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    d = torch.tensor([dof])
    dist.all_reduce(d)
    total = sharding.sum_blocks(gathered).numpy()
    dx, Pn = _finish(total, win.P, flags.noise_feature ** 2)
    q.put((rank, int(d.item()), total, dx, Pn))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_object_update_equals_single_update(built):
    import torch.multiprocessing as mp
    from orcvio_amd import synth
    from oracle import mirror, mirror_objects as mo
    from helpers import rel
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 90)
    procs = [ctx.Process(target=_object_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    flags = synth.Flags(use_larvio=0)
    win = synth.make_window(N=10, F=4, seed=2, flags=flags, track_len=4)
    objs = synth.make_objects(win, n_objects=5, seed=9, sigma_kp=0.004)
    Hs, rs = [], []
    for ob in objs:
        r0, Hf, Jc, counts = mo.object_rows(ob.wTo, ob.shape, ob.kps, ob.frames, True, False)
        Hx, Hf2, r, rc, hx6 = mo.construct_object_residual_jacobians(Jc, [fr['clone'] for fr in ob.frames], Hf, r0, counts,
                                                                      [fr['wTc'] for fr in ob.frames], win.R_b2c[0], win.t_c_b[0], 0,
                                                                      flags.leg_dim, win.N)
        ok, H1, r1 = mirror.nullspace_project_svd(Hf2, Hx, r)
        Hs.append(H1); rs.append(r1)
    H = np.vstack(Hs); r = np.concatenate(rs)
    Ht, rt = mirror.qr_compress(H, r)
    dx, K, Pn = mirror.measurement_update(Ht, rt, win.P, flags.noise_feature ** 2)
    (_, d0, A0, dx0, P0), (_, d1, A1, dx1, P1) = res
    assert d0 == d1 == H.shape[0]
    assert np.array_equal(A0, A1) and np.array_equal(dx0, dx1) and np.array_equal(P0, P1)
    assert rel(dx0, dx) < 1e-8 and rel(P0, Pn) < 1e-9


def test_bench_ships_the_communicator_id_between_two_ranks(tmp_path):
    """bench.py's rendezvous for N > 1 (no torch.distributed process group: the 128 bytes of ncclGetUniqueId go through a file
    named after the launcher's pid and the rendezvous port): two rank processes of one launcher end up with the same id, the
    file is rank 0's to remove.  No GPU is touched (ncclGetUniqueId needs none)."""
    import subprocess
    code = (
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "from orcvio_amd import capi\n"
        "rank = int(os.environ['RANK'])\n"
        "uid, path = bench.ship_unique_id(capi, rank, 2)\n"
        "print('UID', rank, bytes(uid).hex(), path or '-')\n")
    env = dict(os.environ, MASTER_PORT='29871', TMPDIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, '-c', code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in (1, 0)]   # (rank 1 first: it has to wait for the file)
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    got = {}
    for out, _ in outs:
        for ln in out.splitlines():
            if ln.startswith('UID'):
                _, rank, hexid, path = ln.split()
                got[int(rank)] = (hexid, path)
    assert set(got) == {0, 1}
    assert got[0][0] == got[1][0] and len(got[0][0]) == 256
    assert got[1][1] == '-' and os.path.exists(got[0][1])   # rank 0 owns the file (bench.py removes it after the first barrier)
    assert os.path.dirname(got[0][1]) == str(tmp_path)

#!/usr/bin/env python3
"""bench.py -- EKF (MSCKF) updates/s on the BASELINE.json configuration, one process per GPU.

A "step" is one complete measurement update (Jacobians -> nullspace -> gate -> stacked H -> Gram compression ->
Kalman solve -> dx, P+) on inputs already resident in HBM (the contract's `value`).
N=1: config 2 (30 clones x 400 features x 30 observations).  N>1 (weak scaling): every rank holds its own shard of
400 features of ONE joint update; per step the ranks all-gather their compressed blocks over RCCL -- through the
communicator the library handle owns (orcvio_msckf_comm_init / orcvio_msckf_run_update_sharded) -- and each performs
the replicated Kalman solve.  There is ONE RCCL communicator per process: the handle's.  The contract's barrier and
max-over-ranks go through it (orcvio_msckf_comm_barrier / _comm_allreduce_max); the 128-byte unique id travels through a
file under the temporary directory -- no torch.distributed process group is created.

`python bench.py --gpus N` without a launcher spawns the N ranks itself (torch.distributed.run, before anything in this
process touches a GPU); under `python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.

Besides the contract line, the N=1 run reports
  latency        per-update latency (median / p95 over >= 200 updates): device-resident, host-visible through
                 the handle's pinned arena written in place (orcvio_msckf_io_update), host-visible through the copying call,
                 and with the covariance resident in HBM;
  objects_update config 3's object update and the north-star frame (400 features, then 20 objects);
  configs        one entry per BASELINE configuration besides the metric's (1, 3 as a frame, one rank's share of 4, 5):
                 device-resident ms, host-visible median / p95, and the CPU restatement beside each (1 thread, all cores);
  stream_config1 the reference's operating point as a filter loop on the resident covariance (euroc.yaml's flags: hybrid
                 filter with in-state features, ragged tracks): frames/s and per-frame p95.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector / matrix peak (not listed in MI355X_MICROARCH.md)
SIMDS = 256 * 4           # 256 CUs x 4 SIMDs
MODEL_T1_US = 97.0        # single-GPU step of config 2 the scaling model of DESIGN.md section 5 is written for


def algorithmic_flops(N, F, M, leg=22):
    """SURVEY.md 8(d) minimum-work count W = W_J + W_N + W_G + W_Q + W_U for full-length tracks."""
    n = leg + 6 * N
    n_a = 6 + 6 * N
    rho = 2 * M - 3
    m = F * rho
    W_J = 350.0 * F * M
    W_N = F * 4.0 * sum((2 * M - k) * (n_a + 1) for k in range(3))
    W_G = F * (2.0 * rho * n_a ** 2 + rho ** 2 * n_a + rho ** 3 / 3.0)
    W_Q = 2.0 * m * n_a ** 2 - (2.0 / 3.0) * n_a ** 3
    W_U = 6.0 * n_a ** 2 * n + n_a ** 3 / 3.0
    return dict(W_J=W_J, W_N=W_N, W_G=W_G, W_Q=W_Q, W_U=W_U, total=W_J + W_N + W_G + W_Q + W_U)


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher: start N ranks under torch.distributed.run as a CHILD process
    (nothing in this process has touched a GPU; no exec of a process that has) and pass its output / exit code on."""
    import torch   # device_count() does not initialise the GPU on this image
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus}: this node shows {have} GPU(s)')
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def comm_fallback(args, rank, world, err, upd):
    """The handle's communicator could not be created over the transport this run asked for: every rank but 0 leaves (exit 0: the
    launcher must not tear rank 0 down), rank 0 closes its handle and runs `bench.py --gpus N` once more as a child process over the
    other transport, then prints that run's line with the failed transport's record beside it.  Exit code: the child's; 4 if there is
    nothing left to try (the ipc transport itself failed, or the fallback is switched off)."""
    first = os.environ.get('ORCVIO_COMM_TRANSPORT', 'rccl') or 'rccl'
    try:
        upd.close()
    except Exception:
        pass
    if rank != 0:
        return 0
    failed = dict(transport=first, world=world, error=err)
    if first == 'ipc' or os.environ.get('ORCVIO_BENCH_NO_FALLBACK') == '1':
        print(json.dumps(dict(metric='EKF updates/sec, 30 clones x 400 feats', value=None, unit='updates/s', n_gpus=world, steps=args.steps,
                              warmup=args.warmup, error='no communicator', comm_failed=failed)), flush=True)
        return 4
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'GROUP_RANK', 'ROLE_RANK',
                                                          'LOCAL_WORLD_SIZE', 'ROLE_WORLD_SIZE', 'TORCHELASTIC_RUN_ID', 'TORCHELASTIC_RESTART_COUNT',
                                                          'TORCHELASTIC_MAX_RESTARTS', 'TORCHELASTIC_USE_AGENT_STORE', 'TORCH_NCCL_ASYNC_ERROR_HANDLING')}
    env.update(ORCVIO_COMM_TRANSPORT='ipc', ORCVIO_IPC_XDEV='1', ORCVIO_BENCH_NO_FALLBACK='1')
    print(f'bench.py: the {first} communicator failed ({err}); running the benchmark again in fresh processes over the ipc transport', file=sys.stderr, flush=True)
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
    sys.stderr.write(p.stderr[-4000:])
    line = None
    for cand in reversed(p.stdout.strip().splitlines()):
        try:
            line = json.loads(cand)
            break
        except ValueError:
            continue
    if line is None:
        line = dict(metric='EKF updates/sec, 30 clones x 400 feats', value=None, unit='updates/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                    error='the ipc run printed no line', tail=p.stdout[-400:])
    line['comm_failed'] = failed
    line['comm_fallback'] = 'ipc transport in fresh processes; across DIFFERENT devices this transport is UNVERIFIED (ORCVIO_IPC_XDEV=1)'
    print(json.dumps(line), flush=True)
    return p.returncode if line.get('value') is not None else (p.returncode or 4)


def ship_unique_id(capi, rank, world):
    """The 128 bytes of ncclGetUniqueId from rank 0 to the other ranks of this node: a file under the temporary directory,
    named after the launcher's process id and the rendezvous port (both the same for every rank of one launch, different
    for the next), written under another name and renamed, removed by rank 0 when everybody has it."""
    if world == 1:
        return capi.comm_unique_id(), None
    tag = '%s_%s' % (os.getppid(), os.environ.get('MASTER_PORT', '0'))
    path = os.path.join(tempfile.gettempdir(), f'orcvio_bench_comm_id_{tag}')
    # A file older than the launcher of THIS run is a leftover of a crashed run with the same tag: rank 0 removes it,
    # the others never accept it.  The file is created exclusively (no following of a planted link) with owner-only permissions.
    try:
        import psutil
        born = psutil.Process(os.getppid()).create_time() - 1.0
    except Exception:
        born = time.time() - 600.0
    if rank == 0:
        uid = capi.comm_unique_id()
        for stale in (path, path + '.part'):
            try:
                os.unlink(stale)
            except FileNotFoundError:
                pass
        fd = os.open(path + '.part', os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, 'wb') as f:
            f.write(uid)
        os.replace(path + '.part', path)
        return uid, path
    t0 = time.time()
    while True:
        try:
            st = os.lstat(path)
            if st.st_mtime >= born and st.st_uid == os.getuid():
                with open(path, 'rb') as f:
                    uid = f.read()
                if len(uid) == capi.COMM_ID_BYTES:
                    return uid, None
        except FileNotFoundError:
            pass
        if time.time() - t0 > 180:
            raise SystemExit(f'bench.py rank {rank}: no communicator id from rank 0 after 180 s ({path})')
        time.sleep(0.01)


def scaling_model(transport, world):
    """DESIGN.md section 5's T(N) for bench.py's weak-scaling step (400 tracks per rank), microseconds: T1 + a + X(N) + R(N) with T1 the
    single-GPU step, a = 5 the assembly of the block inside k_front, X the exchange (RCCL ~ 20: one latency-bound all-gather launch;
    ipc ~ 8 + (N - 1)), R = 3 + 0.5 N the rank-ordered sum; at N = 1 nothing is exchanged (plain step)."""
    T1, a = MODEL_T1_US, 5.0
    if world <= 1:
        return dict(model_us=T1, model='T1 (no exchange at N = 1; the forced world-1 path adds a + X + R)')
    X = 20.0 if transport == 'rccl' else 8.0 + (world - 1)
    R = 3.0 + 0.5 * world
    return dict(model_us=round(T1 + a + X + R, 1), model=f'T1 {T1} + a {a} + X {X} + R {R} (DESIGN.md 5)')


def percentiles(samples_ms):
    import numpy as np
    a = np.sort(np.asarray(samples_ms))
    return dict(median_ms=float(np.median(a)), p95_ms=float(a[min(len(a) - 1, int(np.ceil(0.95 * len(a))) - 1)]),
                mean_ms=float(a.mean()), min_ms=float(a[0]), n=int(len(a)))


def timed_calls(fn, reps, warm=10, after=None):
    """Per-call wall time of fn() in ms (the garbage collector is off: a collection inside a 0.2 ms call is not the
    library's); `after` runs outside the timed part of every iteration (e.g. restoring the resident covariance)."""
    for _ in range(warm):
        fn()
        if after:
            after()
    gc.collect()
    gc.disable()
    out = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        out.append((time.perf_counter() - t) * 1e3)
        if after:
            after()
    gc.enable()
    return out


def device_resident_ms(upd, steps=100, warm=10):
    """ms per update of the uploaded window, graph replay back to back, one synchronisation at the end."""
    for _ in range(warm):
        upd.run_update()
    upd.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        upd.run_update()
    upd.sync()
    return (time.perf_counter() - t0) / steps * 1e3


def fast_cpu_sweep(orc, win, reps=4):
    """oracle/msckf_fast.c (the sparsity the device path exploits: 13 non-zeros per row in E = X P X^T, Gram form X^T X - T3^T T3)
    over OpenMP team sizes: {threads: best ms}, the best team first.  One thread is part of the sweep: the honest single-core
    figure of the minimum-work algorithm beside the literal port's."""
    import ctypes
    lib = orc.lib()
    host = os.cpu_count() or 1
    teams = sorted({t for t in (1, 4, 8, 16, 32, 64, 128, host) if t <= host})
    sweep = {}
    try:
        for t in teams:
            lib.orc_fast_set_threads(ctypes.c_int(t))
            orc.msckf_update_fast(win)   # (team start-up)
            sweep[t] = min(orc.msckf_update_fast(win)['seconds'] for _ in range(reps)) * 1e3
    finally:
        lib.orc_fast_set_threads(ctypes.c_int(0))
    best = min(sweep, key=sweep.get)
    return best, sweep


def cpu_leg(orc, win, budget_s, what):
    """The CPU restatement beside a configuration: one thread (the literal port) and all cores (minimum-work algorithm, OpenMP),
    bounded: a window whose single-threaded update would exceed the budget is timed on its first tracks and scaled by the
    track count (the per-track work dominates and is the same for every track of these windows)."""
    import dataclasses
    import numpy as np
    out = dict(what=what)
    est = 2.7e-3 * win.F * (win.N / 30.0) ** 2 * float(np.mean(np.diff(win.obs_ptr)) / 30.0)   # ~1.1 s at config 2 on the round-2 hosts
    sub, scale = win, 1.0
    if est > budget_s and win.F > 50:
        keep = max(50, int(win.F * budget_s / est))
        nobs = int(win.obs_ptr[keep])
        sub = dataclasses.replace(win, p_w=win.p_w[:keep].copy(), obs_ptr=win.obs_ptr[:keep + 1].copy(), obs_clone=win.obs_clone[:nobs].copy(),
                                  obs_z=win.obs_z[:nobs].copy(), obs_zvel=win.obs_zvel[:nobs].copy())
        scale = win.F / keep
    try:
        t = orc.msckf_update(sub, want_blocks=False, want_K=False)['seconds']
        out['one_thread'] = dict(ms_per_update=t * scale * 1e3, cores=1, kind='port',
                                 sample='the whole window' if scale == 1.0 else f'the first {sub.F} of {win.F} tracks, scaled by the track count')
    except Exception as e:
        out['one_thread'] = dict(error=str(e))
    try:
        best, sweep = fast_cpu_sweep(orc, win, reps=3)
        out['all_cores'] = dict(ms_per_update=sweep[best], cores=best, kind='port (minimum-work algorithm, OpenMP; best team of the sweep)',
                                ms_by_threads={str(t): round(v, 3) for t, v in sweep.items()})
    except Exception as e:
        out['all_cores'] = dict(error=str(e))
    return out


def config_table(upd, capi, synth, orc, reps, cpu_budget_s):
    """One entry per BASELINE configuration besides the metric's: GPU device-resident ms, host-visible median / p95 (the
    arena written in place), CPU figures beside each."""
    from orcvio_amd import sharding
    out = {}
    shard4, _ = sharding.shard_window(synth.config_window(4), 0, 4)   # what one of config 4's four ranks holds: 500 tracks
    cases = [('config1', synth.config_window(1), 'euroc.yaml shape: 20 clones, 120 ragged tracks (3-6 observations), n = 142'),
             ('config4_one_rank_share', shard4, 'one rank\'s share of config 4: 30 clones, 500 full tracks (28 500 rows)'),
             ('config5_one_gpu', synth.config_window(5), 'kitti_raw.yaml flags, 30 clones, 2 000 full tracks (114 000 rows) on ONE GPU: beyond the '
                                                          'co-residency limit of the fused front end')]
    for name, win, what in cases:
        e = dict(what=what, clones=win.N, tracks=win.F, rows=int(sum(max(2 * int(m) - 3, 0) for m in (win.obs_ptr[1:] - win.obs_ptr[:-1]))))
        upd.upload(win)

        def one():
            upd.run_update()
            upd.sync()
        e['device_resident'] = percentiles(timed_calls(one, reps))   # (graph replay + one synchronisation per update)
        call, io = upd.make_io_call(win)
        e['host_visible'] = percentiles(timed_calls(call, reps))
        upd.cov_set(win.P)
        call_r, io = upd.make_io_call(win, resident_cov=True, want_P=False, commit=True)
        e['host_visible_resident_cov'] = percentiles(timed_calls(call_r, reps, after=lambda: upd.cov_set(win.P)))
        out[name] = e
    # the other half of a config-4 rank's share: 25 of the 100 objects (12 keypoints x 30 frames each), object update from tracks
    try:
        import ctypes as C
        import numpy as np
        oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
        owin = synth.make_window(N=30, F=4, seed=0, flags=oflags, track_len=4)
        objs = synth.make_objects(owin, n_objects=25, seed=4, sigma_kp=0.004)
        ofl = capi.make_flags(oflags)
        ef, arr, keep = upd._object_tracks(objs, owin.R_b2c[0], owin.t_c_b[0], True, False, 0, False)
        Pc = np.ascontiguousarray(owin.P)
        o, res = upd._result(owin.n, 1)

        def host():
            assert upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(ofl), C.byref(ef), owin.N, arr, len(objs), capi._d(Pc), C.byref(res)) == 0
        e = dict(what='25 objects x 12 keypoints x 30 frames (a quarter of config 4\'s 100), orcvio_msckf_update_object_tracks', objects=25,
                 host_visible=percentiles(timed_calls(host, reps, warm=5)), accepted=int(o['accept'][0]), dof=int(res.stats[0]))
        upd.cov_set(owin.P); upd.cov_prefactor(); upd.sync()
        o2, res2 = upd._result(owin.n, 1)
        res2.P_out = None

        def resident():
            assert upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(ofl), C.byref(ef), owin.N, arr, len(objs), None, C.byref(res2)) == 0
        e['host_visible_resident_cov'] = percentiles(timed_calls(resident, reps, warm=5))
        out['config4_one_rank_share']['objects'] = e
        # ... and the whole share as ONE frame call: the 500 tracks' update + commit, then the 25 objects' update + commit
        fc = {}

        def renew():
            upd.cov_set(shard4.P)
            io = upd.io_begin(shard4.flags, shard4.N, shard4.F, int(shard4.obs_ptr[-1]), with_P=False)
            upd.io_fill(io, shard4, with_P=False)
            fc['c'], fc['o'] = upd.make_frame_call(shard4, oflags, objs, owin.R_b2c[0], owin.t_c_b[0], True, False, 0, False, True)
        renew()
        out['config4_one_rank_share']['frame_one_call'] = dict(
            percentiles(timed_calls(lambda: fc['c'](), reps, warm=5, after=renew)),
            what='orcvio_msckf_io_update_frame: 500 tracks + 25 objects, both updates committed, host tracks in, dx out twice')
    except Exception as ex:
        out['config4_one_rank_share']['objects'] = dict(error=repr(ex))
    # beyond the register-resident factorisations (n <= 224): 38 clones, n = 250 -- both factorisations and the solve by 2 x 2 blocks out
    # of the register kernels (capi_update.inc blk2; the LDS-panel Cholesky and k_trsm_rl it replaces took 1.0 ms here: "tested, not
    # timed" until round 5)
    try:
        big = capi.MsckfUpdater(device=getattr(upd, 'device', 0), max_clones=40, max_features=512, max_observations=16384)
        wbig = synth.make_window(N=38, F=400, seed=3, flags=synth.Flags(use_larvio=1), track_len=(20, 30))
        big.upload(wbig)

        def one_big():
            big.run_update()
            big.sync()
        out['window_38_clones'] = dict(what='38 clones, 400 tracks of 20-30 observations, n = 250: beyond the register-resident factorisations '
                                            '(2 x 2 block factorisation out of the register kernels, forked front end; ORCVIO_BLK2=0: the LDS-panel kernels, 1.0 ms)', clones=38, tracks=wbig.F, n=int(wbig.n),
                                       device_resident=percentiles(timed_calls(one_big, reps)))
        big.close()
    except Exception as ex:
        out['window_38_clones'] = dict(error=repr(ex))
    # BASELINE config 5 as it is WORDED ("bbox-only OrcVIO-lite"): beside the 2 000 tracks under kitti_raw.yaml's flags, 100 object
    # tracks WITHOUT keypoints -- object state [pose 6 | shape 3], four bbox rows per in-window frame.  A labelled extension: the
    # reference's lite mode sends no residuals at all (SURVEY note N4: config5_one_gpu above is that reading).
    try:
        import ctypes as C
        import numpy as np
        w5 = synth.config_window(5)
        oflags = synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=w5.flags.noise_feature)
        owin = synth.make_window(N=w5.N, F=4, seed=0, flags=oflags, track_len=4)
        objs = synth.make_objects(owin, n_objects=100, seed=6, sigma_kp=0.004, bbox_only=True)
        ofl = capi.make_flags(oflags)
        ef, arr, keep = upd._object_tracks(objs, owin.R_b2c[0], owin.t_c_b[0], True, False, 0, False)
        upd.cov_set(owin.P); upd.cov_prefactor(); upd.sync()
        o2, res2 = upd._result(owin.n, 1)
        res2.P_out = None

        def bbox_resident():
            assert upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(ofl), C.byref(ef), owin.N, arr, len(objs), None, C.byref(res2)) == 0
        lat = percentiles(timed_calls(bbox_resident, reps, warm=5))
        out['config5_bbox_only'] = dict(
            what='EXTENSION, not reference behaviour (SURVEY note N4): 100 bbox-only object tracks (no keypoints: object state 9 columns, 4 rows '
                 'per frame) x 30 frames = 12 000 rows, orcvio_msckf_update_object_tracks on the resident covariance; the 2 000-track feature '
                 'update of the same frame is config5_one_gpu',
            objects=100, rows=int(sum(4 * sum(1 for fr in ob.frames if fr['clone'] >= 0) for ob in objs)), accepted=int(o2['accept'][0]),
            dof=int(res2.stats[0]), one_launch_compression=upd.counters()['obj_fused'], host_visible_resident_cov=lat,
            frame_ms_with_the_feature_update=round(out['config5_one_gpu']['host_visible_resident_cov']['median_ms'] + lat['median_ms'], 5))
    except Exception as ex:
        out['config5_bbox_only'] = dict(error=repr(ex))
    return out, cases


def config_cpu_legs(configs, cases, orc, cpu_budget_s):
    """The CPU legs of the table, run LAST in bench.py: their OpenMP teams (up to one thread per hardware thread) would
    otherwise still be winding down under the GPU latency measurements."""
    for name, win, what in cases:
        configs[name]['cpu_baseline'] = cpu_leg(orc, win, cpu_budget_s, 'oracle/msckf_oracle.c (1 thread), oracle/msckf_fast.c (all cores)')


def stream_cpp(synth, fl, sigma_px, tag, frames=480):
    """The same stream through the C-ABI from C++ (tests/cpp/stream_bench.cpp, built here with g++ and run as a CHILD process: no Python
    in the loop): the separate calls of the round-5 ABI against ONE call per frame (orcvio_msckf_io_step_frame)."""
    import subprocess
    import tempfile
    from orcvio_amd import capi
    d = tempfile.mkdtemp(prefix='orcvio_stream_')
    exe = os.path.join(d, 'stream_bench')
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-DORCVIO_HAVE_STEP_FRAME', '-o', exe, os.path.join(ROOT, 'tests', 'cpp', 'stream_bench.cpp'),
                           '-L', libdir, '-lorcvio_msckf', f'-Wl,-rpath,{libdir}'])
    fr, P0 = synth.make_stream(fl, sigma_px=sigma_px)
    path = os.path.join(d, tag + '.bin')
    synth.write_stream(path, fr, P0, fl, 1)
    out = {}
    for key, extra in (('one_call_per_frame', ['--mode', 'step']), ('separate_calls', ['--mode', 'calls', '--no-prefactor']),
                       ('separate_calls_prefactor', ['--mode', 'calls'])):
        best = None
        for _ in range(2):   # (two runs of 480 frames each, the better one: a fresh process finds the device's clocks down)
            r = subprocess.run([exe, '--stream', path, '--frames', str(frames)] + extra, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                raise RuntimeError('stream_bench failed: ' + (r.stdout + r.stderr)[-400:])
            j = json.loads(r.stdout.strip().splitlines()[-1])
            if best is None or j['frames_per_s'] > best['frames_per_s']:
                best = j
        out[key] = best
    out['same_results'] = (out['one_call_per_frame']['dx_hash'] == out['separate_calls']['dx_hash'] and
                           out['one_call_per_frame']['P_hash'] == out['separate_calls']['P_hash'])
    return out


def stream_hybrid(upd, capi, synth, fl, sigma_px, label, frames=240, seed=0, one_call=False):
    """A filter LOOP on the resident covariance at the reference's shipped operating point (sw_size 20, max_track_len 6,
    max_features_in_one_grid 1 -> the hybrid filter with feature_idp_dim 1 in-state features; config/euroc.yaml:49-109,
    config/kitti_raw.yaml:77-148), 20-200 lost features per frame with 3-6 observations each, flags `fl` -- through ctypes.  Per frame
    (src/orcvio.cpp:567-594): propagate -> augment -> the hybrid update (MSCKF tracks + the rows of the in-state features, evaluated
    on the device) -> commit -> when the window is full: the prune update on the two clones that leave, commit, marginalisation.
    one_call: orcvio_msckf_io_step_frame (everything enqueued at once, one wait); else the separate calls of the round-5 ABI (with
    cov_prefactor between augmentation and update).  The covariance never leaves HBM; tracks and poses go in, dx comes back.  Frames
    WITH the prune update and frames without it are two different amounts of work: reported separately besides the mixed figure."""
    import ctypes as C
    import numpy as np
    n_slam, idp, leg = 12, 1, 22
    cyc, P0 = synth.make_stream(fl, sigma_px=sigma_px, seed=seed, n_slam=n_slam, idp=idp)
    for c in cyc:   # the per-frame C calls with their arguments marshalled once (a C++ caller has its containers at hand)
        c['sl'] = upd.make_slam_call(idp, c['slam'])
        c['poses'] = synth.pack_poses(c['w'])
        if one_call:
            st = capi.FrameStep()
            st.leg_dim = leg
            c['PhiQ'] = (np.ascontiguousarray(c['Phi']), np.ascontiguousarray(c['Q']))
            st.Phi = capi._d(c['PhiQ'][0]); st.Q = capi._d(c['PhiQ'][1]); st.augment = 1
            st.slam_features = C.pointer(c['sl'].hold[-1])
            if c['prune'] is not None:
                pr = c['prune']
                c['pr_arr'] = [np.ascontiguousarray(pr.p_w), np.ascontiguousarray(pr.obs_ptr, dtype=np.int32),
                               np.ascontiguousarray(pr.obs_clone, dtype=np.int32), np.ascontiguousarray(pr.obs_z)]
                c['pr_tr'] = capi.MsckfTracks(int(pr.F), capi._d(c['pr_arr'][0]), capi._i(c['pr_arr'][1]), capi._i(c['pr_arr'][2]), capi._d(c['pr_arr'][3]), None)
                st.prune_tracks = C.pointer(c['pr_tr'])
            c['rm'] = np.ascontiguousarray(c['remove'], dtype=np.int32)
            if len(c['rm']):
                st.remove_clones = capi._i(c['rm'])
            st.n_remove = len(c['rm'])
            c['st'] = st
            c['res'] = capi.FrameResult()
    upd.set_extra_states(idp * n_slam)
    upd.set_ekf_rows_mode(True)
    times, with_prune, which, n_upd, discards = [], [], [], 0, 0
    lib, h = upd.lib, upd.h

    def fill(io, win, poses):
        io['poses'][:] = poses
        io['obs_ptr'][:] = win.obs_ptr
        if win.F:
            io['p_w'][:] = win.p_w
            io['obs_clone'][:] = win.obs_clone
            io['obs_z'][:] = win.obs_z

    def inplace(c, key, slam_call):
        win = c[key]
        io = upd.io_begin(win.flags, win.N, win.F, int(win.obs_ptr[-1]), with_P=False)   # (same sizes -> same addresses: cheap)
        fill(io, win, c['poses'])
        if slam_call is not None:
            slam_call()
        st = upd.io_update(want_P=False, commit=True)
        return io['dx'], st
    try:
        upd.cov_set(P0)
        gc.collect()
        gc.disable()
        for it in range(frames + 16):
            c = cyc[it % len(cyc)]
            t = time.perf_counter()
            if one_call:
                w = c['w']
                io = upd.io_begin(w.flags, w.N, w.F, int(w.obs_ptr[-1]), with_P=2)
                fill(io, w, c['poses'])
                rc = lib.orcvio_msckf_io_step_frame(h, C.byref(c['st']), C.byref(c['res']))
                if rc != 0:
                    raise capi.MsckfError(rc, 'orcvio_msckf_io_step_frame')
                got = io['dx']
                n_upd += 1 + (1 if c['prune'] is not None else 0)
                discards += int(c['res'].stats[4])
            else:
                upd.cov_propagate(c['Phi'], c['Q'])
                upd.cov_augment()
                upd.cov_prefactor()
                got, st = inplace(c, 'w', c['sl'])
                n_upd += 1
                discards += int(st[4]) if st is not None else 0   # (discard_large_update: reported; the caller skips the state increment, P+ stands)
                if c['prune'] is not None:
                    got, st = inplace(c, 'prune', None)   # (no rows of the in-state features in this one)
                    n_upd += 1
                if c['remove']:
                    upd.cov_remove_clones(leg, c['remove'])
                upd.sync()
            if it >= 16:
                times.append((time.perf_counter() - t) * 1e3)
                with_prune.append(c['prune'] is not None)
                which.append(it % len(cyc))
            if not np.all(np.isfinite(got)):
                raise RuntimeError('non-finite dx in the stream')
        upd.sync()
        gc.enable()
    finally:
        gc.enable()
        upd.set_ekf_rows_mode(False)
        upd.set_extra_states(0)
    p = percentiles(times)
    tw = [t for t, w_ in zip(times, with_prune) if w_]
    tn = [t for t, w_ in zip(times, with_prune) if not w_]

    def cls(ts):
        if not ts:
            return None
        q = percentiles(ts)
        q['p95_over_median'] = q['p95_ms'] / q['median_ms']
        return q
    # jitter proper: the SAME frame of the cycle (same tracks, same amount of work) repeated -- p95 / median per distinct frame
    per_frame = {}
    for t, k in zip(times, which):
        per_frame.setdefault(k, []).append(t)
    jitter = {str(k): round(percentiles(v)['p95_ms'] / percentiles(v)['median_ms'], 4) for k, v in sorted(per_frame.items())}
    return dict(p, frames_per_s=1e3 / p['mean_ms'], updates_per_frame=n_upd / (frames + 16), in_state_features=n_slam,
                frames_with_prune_update=cls(tw), frames_without_prune_update=cls(tn), large_update_flags=discards,
                front_fallbacks=upd.counters()['front_fallbacks'],   # (cumulative for the handle: 0 = no fused front end lost its co-residency bet)
                p95_over_median_per_distinct_frame=jitter, worst_p95_over_median_same_frame=max(jitter.values()) if jitter else None,
                tracks_per_frame=[int(c['w'].F) for c in cyc], form='one call per frame (orcvio_msckf_io_step_frame)' if one_call else 'separate calls',
                what=label + ': hybrid filter (12 in-state features, 1 parameter each), 19/20-clone window, 20-200 ragged tracks '
                     'per frame, THROUGH CTYPES (the C++ figure of the same frames: `cpp`); per frame: propagate, augment, hybrid update + commit, '
                     'every second frame the prune update + commit + marginalisation of two clones; covariance resident in HBM; the C calls\' '
                     'arguments are marshalled once per pre-generated frame, the ctypes call overhead is included')


def stream_leg(upd, capi, synth, fl, sigma_px, label, tag):
    """One stream, four ways: C++ one call per frame / C++ separate calls (child process), ctypes one call per frame / ctypes separate calls."""
    out = dict(what=label)
    try:
        out['cpp'] = stream_cpp(synth, fl, sigma_px, tag)
        out['frames_per_s'] = out['cpp']['one_call_per_frame']['frames_per_s']
    except Exception as e:
        out['cpp'] = dict(error=repr(e))
    try:
        out['ctypes_one_call_per_frame'] = stream_hybrid(upd, capi, synth, fl, sigma_px, label, one_call=True)
        out['ctypes_separate_calls'] = stream_hybrid(upd, capi, synth, fl, sigma_px, label, one_call=False)
        out.setdefault('frames_per_s', out['ctypes_one_call_per_frame']['frames_per_s'])
    except Exception as e:
        out['ctypes_error'] = repr(e)
    return out


def stream_config1(upd, capi, synth):
    """config/euroc.yaml's shipped flags: LARVIO Jacobians, sigma 0.008, no discard."""
    return stream_leg(upd, capi, synth, synth.Flags(use_larvio=1), None, 'euroc.yaml flags (LARVIO Jacobians, sigma 0.008)', 'config1')


def stream_config5(upd, capi, synth):
    """config/kitti_raw.yaml's shipped flags (:103, :135-158): OrcVIO right-perturbation Jacobians (use_larvio_flag 0,
    use_left_perturbation_flag 0), noise_feature 1, feature_idp_dim 1, discard_large_update_flag 1 -- BASELINE config 5's real
    operating point on one GPU (sw_size 20, max_track_len 6, max_features_num 200: 20-200 ragged tracks)."""
    fl = synth.Flags(use_larvio=0, use_left_perturbation=0, noise_feature=1.0, discard_large_update=1)
    return stream_leg(upd, capi, synth, fl, 0.008, 'kitti_raw.yaml flags (OrcVIO right-perturbation Jacobians, sigma 1, discard flag on)', 'config5')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the per-configuration table and the config-1 stream')
    ap.add_argument('--clones', type=int, default=30)
    ap.add_argument('--features', type=int, default=400)
    ap.add_argument('--latency-updates', type=int, default=300, help='updates per latency mode (>= 200, SURVEY 8d)')
    ap.add_argument('--timed-blocks', type=int, default=5, help='the K timed steps are run at least this many times (each block bracketed '
                    'by barrier + synchronize on both sides); the MEDIAN of the last `timed-blocks` blocks is the one reported')
    ap.add_argument('--max-blocks', type=int, default=25, help='blocks are timed until the last `timed-blocks` agree within 0.7 %% or this '
                    'many have run (the clocks of a device that was idle ramp up for tens of ms)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args))

    import numpy as np
    import torch
    from orcvio_amd import capi, synth

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and rank == 0:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s)', file=sys.stderr)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU path')
    if os.environ.get('ORCVIO_BENCH_ONE_DEVICE') == '1':   # every rank on device 0 (needs ORCVIO_COMM_TRANSPORT=ipc: RCCL refuses two ranks per
        local_rank = 0                                     # device); exercises the N > 1 path on a one-GPU box -- not a scaling measurement
    torch.cuda.set_device(local_rank)
    # ORCVIO_BENCH_FORCE_DIST=1 drives the multi-GPU code path (communicator, all-gather) with world size 1
    use_dist = world > 1 or os.environ.get('ORCVIO_BENCH_FORCE_DIST') == '1'

    N, F = args.clones, args.features
    # weak scaling: one joint update of F*world tracks, dealt across the ranks
    from orcvio_amd import sharding
    full = synth.make_window(N=N, F=F * world, seed=0, flags=synth.Flags(use_larvio=1))
    win, _ = sharding.shard_window(full, rank, world)
    upd = capi.MsckfUpdater(device=local_rank, max_clones=max(32, N), max_features=max(2048, win.F),
                            max_observations=max(65536, int(win.obs_ptr[-1])))
    id_file = None
    if use_dist:   # the ONE communicator of this process: the handle's
        uid, id_file = ship_unique_id(capi, rank, world)
        try:
            upd.comm_init(uid, rank, world)
            upd.comm_barrier()
        except capi.MsckfError as e:
            # The communicator could not be had (RCCL missing, its bootstrap timed out, a rank never arrived: bounded waits, ORCVIO_COMM_TIMEOUT_S).
            # Day one of a real node must still be informative: rank 0 starts the SAME benchmark again in FRESH processes
            # (a child torchrun: nothing that has touched a GPU is re-executed) over the second transport (HIP IPC + shared memory;
            # across devices it is UNVERIFIED, hence ORCVIO_IPC_XDEV=1 and a marker in the line) and prints ONE line carrying both records.
            if id_file:
                try:
                    os.remove(id_file)
                except OSError:
                    pass
            raise SystemExit(comm_fallback(args, rank, world, repr(e), upd))
        if id_file:
            os.remove(id_file)
    upd.upload(win)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            upd.comm_barrier()   # (bounded: a rank that never arrives is an error, not a hang)
        torch.cuda.synchronize()

    def step():   # kernels AND the collective on the handle's own stream: RCCL is ordered behind the rank's block and in front of the solve
        if not use_dist:
            upd.run_update()
        else:
            upd.run_update_sharded()   # local tracks + compression -> RCCL all-gather -> sum + replicated solve

    gc.collect()
    gc.disable()   # the timed region is a few ms: an interpreter collection in the middle of it would be most of it (and one
                   # between warm-up and timing would let the GPU clock down again)
    # The timed region: EXACTLY K steps between barrier + synchronize on both sides -- run `timed_blocks` times back to back, the
    # median block reported (one 20-step block is 2 ms, of which ~30 us are the first launch reaching the device and
    # the last completion signal reaching the host; a single block reads 1.5-2 % low and scatters by as much from run to run).
    nblk = max(1, args.timed_blocks)
    max_blk = max(nblk, args.max_blocks)

    def run_blocks(step_fn):
        blocks = []
        while True:
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_fn()
            barrier()
            blocks.append(time.perf_counter() - t0)
            if world > 1:   # every block: the slowest rank's time (so that all ranks take the same decision below)
                blocks[-1] = float(upd.comm_allreduce_max([blocks[-1]])[0])
            # a fresh process finds the device idle and its clocks ramp up under load for the first tens of ms: keep timing blocks until the
            # last `timed_blocks` of them agree within 0.7 % (or max_blocks have run); the median of THOSE is reported
            if len(blocks) >= nblk:
                last = blocks[-nblk:]
                if len(blocks) >= max_blk or (max(last) - min(last)) <= 0.007 * min(last):
                    return blocks

    def median_block(blocks):
        used = sorted(blocks[-nblk:])
        return float(used[len(used) // 2])
    # (a) the QUEUED form, a side figure: K device-resident updates enqueued back to back, nothing read between them
    for _ in range(args.warmup):
        step()
    queued_dt = median_block(run_blocks(step))
    # (b) THE metric: host-visible updates, one at a time -- flat inputs in host memory (the handle's pinned
    # arena, written in place by the caller) -> dx, P+, gamma, accept in host memory; update k+1 starts when update k's results are
    # there, as in a filter.  With more than one rank: the sharded update and a stream synchronisation per step.
    if world == 1:
        hv_step, _io_views = upd.make_io_call(win)
    else:
        def hv_step():
            step()
            upd.sync()
    for _ in range(args.warmup):
        hv_step()
    block_dt = run_blocks(hv_step)
    gc.enable()
    dt = median_block(block_dt)
    ms = dt / args.steps * 1e3
    upd.upload(win)   # (the staged form again: what the per-kernel profile and the side measurements below run on)

    # per-update latency of the joint update on every rank count (sync after every update)
    def one_sync():
        step()
        upd.sync()
    lat_dev = timed_calls(one_sync, max(200, args.latency_updates))   # every rank: the sharded step holds a collective
    comm = None
    if use_dist:
        upd.comm_barrier()
        # where the joint update's time goes on this rank (HIP events on the handle's stream between its three parts; collective)
        try:
            parts = upd.profile_sharded(reps=20)
            comm = dict(upd.comm_details(), **{k: round(v, 2) for k, v in parts.items()})
            if not comm.get('ipc_across_devices_unverified'):
                comm.pop('ipc_across_devices_unverified', None)   # (kept only when it says something: ipc between ranks on different devices)
            comm.pop('rank', None)
            comm.update(scaling_model(comm['transport'], world))
        except Exception as e:
            comm = dict(error=repr(e))
        upd.comm_barrier()

    out = None
    if rank == 0:
        M = N
        W = algorithmic_flops(N, F * world, M)
        # roofline of the dominant kernel, timed live with HIP events on the launch stream
        prof = upd.profile(reps=20)
        n = 22 + 6 * N
        NA = n - 15
        # algorithmic FP64 work attributed to each kernel (DESIGN.md "Roofline accounting")
        kflops = {
            'k_feature': (W['W_J'] + W['W_N'] + W['W_G']) / world,   # Jacobians + nullspace + gate
            'k_gram': W['W_Q'] / world,                                # stack compression
            'k_assemble': 0.0,
            'k_potrf(P)': n ** 3 / 3.0,
            'k_gemm(U)': 2.0 * (NA + 1) * NA * n,
            'k_gemm(M)': 2.0 * NA * n * n / 2.0,
            'k_potrf(M)': n ** 3 / 3.0,
            'k_trsm': 1.0 * n * n * (n + 1),
            'k_potrf_solve(M)': n ** 3 / 3.0 + 1.0 * n * n * (n + 1),   # factorisation + trailing solve, one launch
            'k_potrf_solve_la(M)': n ** 3 / 3.0 + 1.0 * n * n * (n + 1),   # the same launch with the trailing update on far workgroups
            'k_finish': 1.0 * n * (n + 1) * (n + 1),
        }
        if 'k_finish' not in prof:   # P+ = s2 Z^T Z and dx by finish workgroups of the factorisation + solve launch (LaFin): its work too
            kflops['k_potrf_solve_la(M)'] += kflops['k_finish']
        # k_front = the tracks, the compression and chol(P) in one launch (the default whenever they are co-resident)
        kflops['k_front'] = kflops['k_feature'] + kflops['k_gram'] + kflops['k_potrf(P)']
        # k_potrf(P) runs on a side stream, overlapped with k_feature/k_gram: not on the critical path
        crit = {k: v for k, v in prof.items() if k != 'k_potrf(P)'}
        longest = max(crit, key=crit.get)
        # The roofline is quoted for the factorisation + solve launch whenever the update runs in its fused form: the step is two
        # dependent 202-pivot Cholesky chains (chol P inside k_front, chol M here: ~80 % of it), and this launch is the one whose
        # SURVEY count (n^3/3 + n^2 (n+1)) is the work it executes.  k_front is a few us LONGER (the max of chol P and the tracks),
        # but the SURVEY count of what it covers is the dense H'PH'^T / QR-of-the-stack count that it does not execute: its
        # dense-equivalent and executed figures are in `longest_kernel`.
        solve = 'k_potrf_solve_la(M)' if 'k_potrf_solve_la(M)' in crit else 'k_potrf_solve(M)'
        dom = solve if solve in crit and 'k_front' in crit else longest
        achieved = kflops[dom] / (prof[dom] * 1e-3) / 1e12
        # HBM traffic and executed matrix-core work of every kernel: PMC counters (FETCH_SIZE + WRITE_SIZE,
        # SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES; separate rocprofv3 passes) of the committed profile of
        # this round, per launch; null if that profile does not list the kernel
        traffic = None
        traffic_source = None
        critical_path = None
        key_of = {'k_feature': 'k_feature<3>', 'k_potrf(M)': 'k_potrf_reg<16>', 'k_potrf_solve(M)': 'k_potrf_solve<>', 'k_potrf_solve_la(M)': 'k_potrf_solve_la<3, false>',
                  'k_trsm': 'k_trsm_lds', 'k_finish': 'k_finish_sqrt', 'k_gram': 'k_gram_pair', 'k_assemble': 'k_assemble_A',
                  'k_front': 'k_front<3, 16>', 'k_gemm(U)': 'k_gemm_asmA', 'k_gemm(M)': 'k_gemm'}
        try:
            import glob
            import hashlib
            import re as _re

            def _round_key(pth):   # r5l < r6a < r10a: by round NUMBER, then suffix (a lexical sort puts r10a in front of r5l: ADVICE r5)
                mm = _re.match(r'r(\d+)([a-z]*)_', os.path.basename(pth))
                return (int(mm.group(1)), mm.group(2)) if mm else (-1, '')
            pm_path = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), key=_round_key)[-1]
            pm_doc = json.load(open(pm_path))
            pm = pm_doc['kernels']
            lib_now = hashlib.sha256(open(capi.LIB_PATH, 'rb').read()).hexdigest()[:16]
            pm_build = pm_doc.get('build')
            pm_sha = pm_build.get('liborcvio_msckf_sha16') if isinstance(pm_build, dict) else pm_build
            # counters of ANOTHER build say nothing about this one: traffic stays null -- unless that build was made from
            # the SAME sources (hipcc's output is not reproducible byte for byte: a rebuild of this tree has another hash, the same kernels)
            from orcvio_amd import build as _build
            src_now = _build.source_sha16()
            pm_src = pm_build.get('source_sha16') if isinstance(pm_build, dict) else None
            same_sources = pm_sha != lib_now and pm_src is not None and pm_src == src_now
            if pm_sha != lib_now and not same_sources:
                traffic_source = dict(file=os.path.relpath(pm_path, ROOT), build=pm_doc.get('build'), measured_in_this_run=False,
                                      refused='taken on another build of the library (this one: %s, sources %s)' % (lib_now, src_now))
                raise LookupError('stale pmc profile')
            # the counters are those of a COMMITTED profile, not of this run: say which file and which build
            traffic_source = dict(file=os.path.relpath(pm_path, ROOT), build=pm_doc.get('build'), measured_in_this_run=False)
            if same_sources: traffic_source['matched_by'] = 'sources (another build of the same sources: %s)' % lib_now
            # (the solve launch's template arguments: look-ahead depth, stamps, finish inside the launch -- two arguments in profiles older than r5i)
            for cand in (('k_potrf_solve_la<3, false, true>',) if 'k_finish' not in prof else ()) + ('k_potrf_solve_la<3, false, false>', 'k_potrf_solve_la<3, false>'):
                if cand in pm:
                    key_of['k_potrf_solve_la(M)'] = cand
                    break
            for cand in ('k_front<3, 16, false>', 'k_front<3, 16>'):   # (a third template argument since the end of round 5)
                if cand in pm:
                    key_of['k_front'] = cand
                    break
            if N == 30 and F == 400:
                def pmc_of(k):   # (the template argument of the factorisation kernels is the block-column capacity: the smallest
                    #                  instantiation that holds this problem's active columns is the one the update launches)
                    if key_of.get(k) in pm:
                        return pm[key_of[k]]
                    want, need = key_of.get(k, '').split('<')[0], (NA + 15) // 16
                    hits = sorted((int(name.split('<')[1].rstrip('>')), name) for name in pm
                                  if want and name.split('<')[0] == want and name.split('<')[1].rstrip('>').isdigit())
                    hits = [name for cap, name in hits if cap >= need]
                    return pm[hits[0]] if hits else None
                if pmc_of(dom):
                    traffic = 1024.0 * (pmc_of(dom)['FETCH_SIZE_KB_median'] + pmc_of(dom)['WRITE_SIZE_KB_median'])
                critical_path = {}
                for k, t_ms in prof.items():
                    e = pmc_of(k)
                    if not e:
                        continue
                    item = dict(ms=round(t_ms, 5))
                    if 'SQ_INSTS_VALU_MFMA_MOPS_F64_median' in e:   # flops the matrix cores actually executed
                        item['executed_mfma_flop'] = 512.0 * e['SQ_INSTS_VALU_MFMA_MOPS_F64_median']
                        item['executed_mfma_tflops'] = item['executed_mfma_flop'] / (t_ms * 1e-3) / 1e12
                    if 'SQ_VALU_MFMA_BUSY_CYCLES_median' in e and 'GRBM_GUI_ACTIVE_median' in e and e['GRBM_GUI_ACTIVE_median'] > 0:
                        # busy cycles summed over SIMDs / (wall cycles x SIMDs of the device)
                        item['mfma_busy_frac'] = e['SQ_VALU_MFMA_BUSY_CYCLES_median'] / (e['GRBM_GUI_ACTIVE_median'] * SIMDS)
                    critical_path[k] = item
                critical_path['chain'] = dict(
                    what='two dependent 202- and 187-pivot Cholesky chains (chol P inside k_front, chol M in k_potrf_solve_la): 13 / 12 block '
                         'steps x 16 pivots each, 180 cycles per pivot + one LDS hand-off and eight dependent MFMAs per block step '
                         '(4.6 k cycles); since round 5 the trailing update is spread over far workgroups (look-ahead 3), the chain '
                         "workgroup's workers apply three panels to the arriving row: 65 k cycles for chol M (first five steps still "
                         'bound by the workers, 6 k each; DESIGN.md 3.3, 6); latency-bound, not MFMA- or HBM-bound',
                    pivots=2 * n, block_steps=2 * ((n + 15) // 16))
        except Exception:
            pass
        executed = None
        if critical_path and dom in critical_path and 'executed_mfma_tflops' in critical_path[dom]:
            executed = dict(mfma_tflops=critical_path[dom]['executed_mfma_tflops'], frac=critical_path[dom]['executed_mfma_tflops'] / FP64_PEAK_TFLOPS,
                            mfma_busy_frac=critical_path[dom].get('mfma_busy_frac'),
                            what='FP64 matrix-core work this kernel EXECUTES (PMC SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flop of the committed profile) / '
                                 'its time in this run: the utilisation figure')
        whole = None
        if critical_path:   # FP64 matrix-core work EXECUTED by all launches of the step / the step time of this run
            fl = sum(v.get('executed_mfma_flop', 0.0) for k, v in critical_path.items() if k != 'chain')
            whole = fl / (queued_dt / args.steps) / 1e12 / FP64_PEAK_TFLOPS if fl else None   # (the device's time per step: the queued form)
        roofline = dict(bound='mfma', kernel=dom, achieved=achieved, peak=FP64_PEAK_TFLOPS, unit='TFLOP/s',
                        frac=achieved / FP64_PEAK_TFLOPS, traffic=traffic, traffic_source=traffic_source if (traffic is not None or (traffic_source or {}).get('refused')) else None,
                        kernel_us=prof[dom] * 1e3, whole_step_executed_frac=whole, executed=executed,
                        per_kernel_frac={k: round(kflops[k] / (prof[k] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5) for k in prof},
                        longest_kernel=dict(
                            kernel=longest, ms=round(prof[longest], 5),
                            dense_equivalent_tflops=round(kflops[longest] / (prof[longest] * 1e-3) / 1e12, 3),
                            executed_mfma_tflops=(critical_path or {}).get(longest, {}).get('executed_mfma_tflops'),
                            what='the longest launch by time.  For k_front (tracks + compression + chol P in one launch; its critical path is '
                                 'the 202-pivot chain of chol P) the SURVEY count of what it covers is the DENSE count of the reference '
                                 "algorithm -- H'PH'^T of every track, the QR of the 22 800-row stack -- which it does not execute (13 "
                                 'non-zeros per un-projected row, Gram form): dense_equivalent_tflops says how fast a dense implementation '
                                 'would have to run to match (it can exceed the peak), executed_mfma_tflops what the matrix cores do'),
                        note='achieved = algorithmic FP64 work of the reference algorithm attributed to the kernel (SURVEY 8d: n^3/3 + '
                             'n^2 (n+1) for the factorisation of M and the two triangular solves, n = 202 -- the count of the full-size '
                             'problem although the launch factors the 187 active columns only' + (
                                 '; + n (n+1)^2 for P+ = s2 Z^T Z and dx, which finish workgroups of the same launch compute' if 'k_finish' not in prof else '')
                             + ') / kernel time by HIP events on the launch stream, median of 20 measurements of (10 x [k_gemm(M); this '
                             'launch] - 10 x [k_gemm(M)]) / 10 -- the launch with one launch gap, as it sits in the update\'s graph (an event '
                             'record on either side of a single launch adds 3-5 us to it; the other stages in kernel_ms are timed that way, '
                             'one pass of the update\'s launches back to back with an event between two stages).  Well under 1 % of the FP64 '
                             'matrix peak: a latency-bound chain '
                             '(critical_path.chain), as is chol P inside k_front; per_kernel_frac lists every launch of the step by the '
                             'same rule.',
                        kernel_ms={k: round(v, 5) for k, v in prof.items()},
                        algorithmic_equiv=dict(
                            what='dense-count flops of the reference algorithm divided by OUR kernel time: how fast a dense '
                                 'implementation would have to run to match; NOT utilisation (the kernels execute far fewer flops)',
                            per_kernel_tflops={k: round(kflops[k] / (prof[k] * 1e-3) / 1e12, 4) for k in prof},
                            whole_update_tflops=W['total'] / (queued_dt / args.steps) / 1e12),
                        critical_path=critical_path)

        latency = dict(device_resident=dict(percentiles(lat_dev), what='graph replay + stream sync per update, inputs and results in HBM'))
        objects = None
        cpu = None
        configs = None
        stream1 = None
        stream5 = None
        orc = None
        if world == 1:
            reps = max(200, args.latency_updates)
            # host-visible (SURVEY 8d's metric): tracks + poses + P in host memory -> dx, P+, gamma, accept in host memory.  The
            # caller has written its flat inputs into the handle's pinned arena (orcvio_msckf_io_begin) and reads the results
            # where they land: ONE graph launch, the calling thread waits on a flag word in host-coherent memory
            call_io, io = upd.make_io_call(win)
            latency['host_visible'] = dict(percentiles(timed_calls(call_io, reps)),
                                           what='orcvio_msckf_io_update: tracks + poses + P in the handle\'s pinned arena (written in place by '
                                                'the caller) -> dx, P+, gamma, accept in pinned host memory')
            # ... the copying call: the caller's own arrays are copied into the arena, the results out of it
            call_host, _ = upd.make_update_call(win)   # argument structs marshalled once: the C call is what is timed
            latency['host_visible_copying_call'] = dict(percentiles(timed_calls(call_host, reps)),
                                                        what='orcvio_msckf_update_features: the same through caller-owned buffers (two more copies '
                                                             'of P, 326 KB each)')
            # ... with the covariance resident in HBM: only tracks + poses go in, dx / gamma / accept come back, P+ and its
            # square-root factor are committed on the device inside the same launch; the prior is restored outside the timed part
            upd.cov_set(win.P)
            call_res, io = upd.make_io_call(win, resident_cov=True, want_P=False, commit=True)
            latency['host_visible_resident_cov'] = dict(percentiles(timed_calls(call_res, reps, after=lambda: upd.cov_set(win.P))),
                                                        what='resident prior, commit inside the launch: tracks + poses in, dx out')
            # ... and with the Cholesky of the prior started when the covariance was last touched (orcvio_msckf_cov_prefactor behind
            # propagate / augment, i.e. while the front end still tracks the image): the update finds the factor resident

            def restore_and_prefactor():
                upd.cov_set(win.P)
                upd.cov_prefactor()
                upd.sync()
            restore_and_prefactor()
            latency['host_visible_resident_prefactored'] = dict(
                percentiles(timed_calls(call_res, reps, after=restore_and_prefactor)),
                what='as host_visible_resident_cov, the prior factored ahead of the call (orcvio_msckf_cov_prefactor, outside the timed '
                     'part: it runs while the front end tracks the image)')
            upd.upload(win)
            if not args.no_cpu_baseline:
                from oracle import oracle as orc   # checker used as the reported CPU baseline ("port")
            # config 3 adds 20 objects x 12 keypoints to the same window: the object update (a second EKF update per
            # frame in the reference, src/orcvio.cpp:2154-2193) from object tracks, host buffers in and out
            if N == 30:
                try:
                    objects = objects_section(upd, capi, synth, orc, np, win)
                except Exception as e:   # never let the side measurement break the metric line
                    objects = dict(error=repr(e))
            cases = None
            if not args.no_configs and N == 30 and F == 400:
                try:
                    configs, cases = config_table(upd, capi, synth, orc, 100, 0.8)
                except Exception as e:
                    configs = dict(error=repr(e))
                try:
                    stream1 = stream_config1(upd, capi, synth)
                except Exception as e:
                    stream1 = dict(error=repr(e))
                try:
                    stream5 = stream_config5(upd, capi, synth)
                except Exception as e:
                    stream5 = dict(error=repr(e))
                upd.upload(win)
            # every GPU figure is taken: now the CPU legs (their OpenMP teams spin down for a while after each call)
            if orc is not None:
                reps_c = 2
                t = []
                for _ in range(reps_c):
                    t.append(orc.msckf_update(win, want_blocks=False, want_K=False)['seconds'])
                cpu = dict(value=1.0 / min(t), unit='updates/s', cores=1, kind='port',
                           sample=f'{reps_c} full updates of the same workload, best of {reps_c} ({min(t):.2f} s each), '
                                  'single-threaded plain-C restatement of the reference algorithm (full-U nullspace, dense gate, '
                                  'QR of the stack, LDLT-style solve); the reference itself (Eigen / SPQR) cannot be built in this '
                                  'image, so this is NOT the Eigen denominator of the >= 50x target in BASELINE.json',
                           host_cores=os.cpu_count())
                try:   # best-effort all-cores CPU variant (minimum-work algorithm, tracks parallelised)
                    best, sweep = fast_cpu_sweep(orc, win, reps=5)
                    cpu['all_cores'] = dict(value=1e3 / sweep[best], unit='updates/s', cores=best, ms_per_update=sweep[best],
                                            ms_by_threads={str(t): round(v, 3) for t, v in sweep.items()},
                                            what='same results with the minimum-work algorithm on the CPU -- the sparsity the device '
                                                 'path exploits (E = X P X^T from 13 non-zeros per row, three reflectors, Gram form '
                                                 'X^T X - T3^T T3, square-root solve), tracks parallelised with OpenMP; the best team '
                                                 'size of the sweep.  This, not the literal port, is the CPU figure a tuned host '
                                                 'implementation would be near')
                except Exception as e:
                    cpu['all_cores'] = dict(error=str(e))
            if cases is not None:
                try:
                    if orc is not None:
                        config_cpu_legs(configs, cases, orc, 0.8)
                    if objects and 'frame_config3' in objects:
                        configs['config3_frame'] = dict(what='400-feature update, then the 20-object update on the P+ it left (SURVEY note N7), '
                                                             'covariance resident in between; host_visible: ONE call '
                                                             '(orcvio_msckf_io_update_frame), two_calls: orcvio_msckf_io_update + '
                                                             'orcvio_msckf_update_object_tracks + cov_commit; see objects_update',
                                                        host_visible=objects.get('frame_config3_one_call', objects['frame_config3']),
                                                        two_calls=objects['frame_config3'], cpu_baseline=dict(
                                                            features=cpu and dict(ms_per_update=1e3 / cpu['value'], cores=1, kind='port'),
                                                            objects=objects.get('cpu_baseline')))
                except Exception as e:
                    configs['cpu_legs_error'] = repr(e)
        # Weak scaling: every rank keeps one 400-feature shard, a step is ONE joint update of 400 x world features
        # (rank-local tracks + compression, one RCCL all-gather, replicated solve).  `value` is the whole-job
        # aggregate in the metric's own unit -- 400-feature update shards processed per second by all ranks =
        # world x joint updates/s -- so that value(N) / (N value(1)) is the usual weak-scaling efficiency T(1)/T(N);
        # the joint-update rate is reported beside it.
        try:
            import hashlib
            lib_sha16 = hashlib.sha256(open(capi.LIB_PATH, 'rb').read()).hexdigest()[:16]
        except Exception:
            lib_sha16 = None
        out = dict(lib_sha16=lib_sha16, metric='EKF updates/sec, 30 clones x 400 feats', value=world * args.steps / dt, unit='updates/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms, higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
                   config=dict(workload='config2: synthetic 30-clone window, 400 point features x 30 observations '
                                        'per GPU (22 800 stacked rows x 202 columns), LARVIO Jacobians',
                               clones=N, features_per_gpu=F, observations_per_feature=N,
                               unit='one update of 30 clones x 400 features; at N GPUs one step is a joint update of '
                                    '400 N features = N units',
                               joint_updates_per_s=args.steps / dt, features_per_joint_update=F * world,
                               parallelism=f'features sharded over {world} GPU(s), all-gather of compressed blocks through '
                                           'the handle\'s RCCL communicator (the only communicator of the process)',
                               value_is='HOST-VISIBLE updates, one at a time (SURVEY 8d): tracks + poses + P in the pinned arena -> dx, P+, gamma, '
                                        'accept in host memory (orcvio_msckf_io_update); the queued device-resident throughput is queued_updates_per_s'),
                   queued_updates_per_s=world * args.steps / queued_dt, queued_ms_per_step=queued_dt / args.steps * 1e3,
                   timed_blocks=len(block_dt), blocks_reported=nblk, block_ms_per_step=[round(v / args.steps * 1e3, 5) for v in block_dt],
                   sequential_updates_per_s=(1000.0 / latency['host_visible']['median_ms']) if 'host_visible' in latency else None,
                   sequential_is='1 / median host-visible latency over >= 200 updates of the same call `value` times in blocks of K',
                   comm=comm, roofline=roofline, cpu_baseline=cpu, latency=latency, objects_update=objects, configs=configs, stream_config1=stream1, stream_config5=stream5)
    if use_dist:
        upd.comm_barrier()
    if out is not None:
        try:   # cumulative counters of the handle (front_fallbacks: fused front ends re-run because another tenant held compute units)
            out['counters'] = upd.counters()
        except Exception:
            pass
    upd.close()
    bad_comm = bool(use_dist and out is not None and isinstance(out.get('comm'), dict) and out['comm'].get('ranks_seen') != world)
    if bad_comm:   # a block was missing from the joint update: the figure above is not a measurement of N ranks
        out['comm']['error'] = 'ranks_seen %s != world %d: the line is NOT a valid N-rank measurement' % (out['comm'].get('ranks_seen'), world)
    if rank == 0:
        sys.stdout.flush()
        try:   # RCCL writes its version banner through C stdio: flush that buffer first so that the JSON line comes last
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        emit(out, block_dt)
    if bad_comm:
        raise SystemExit(3)


def _r(v, nd=6):
    return None if v is None else round(float(v), nd)


def compact_line(out, block_dt=None, detail_file=None):
    """The contract line: <= 4 KB, the LAST line on stdout (the round driver keeps only the tail of stdout, and a
    20 KB line with every side measurement in it could not be parsed).  Everything else goes to `bench_detail.json`."""
    rf = out.get('roofline') or {}
    cpu = out.get('cpu_baseline')
    lat = out.get('latency') or {}
    obj = out.get('objects_update') or {}
    cfg = out.get('config') or {}
    line = {k: out.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                    'vs_baseline', 'dtype', 'data')}
    line['config'] = {k: cfg.get(k) for k in ('workload', 'clones', 'features_per_gpu', 'observations_per_feature', 'features_per_joint_update')}
    line['config']['workload'] = (cfg.get('workload') or '')[:160]
    line['config']['value_is'] = 'host-visible updates one at a time (SURVEY 8d: pinned arena in -> dx, P+ in host memory); queued device-resident: queued_updates_per_s'
    line['queued_updates_per_s'] = _r(out.get('queued_updates_per_s'), 1)
    line['queued_ms_per_step'] = _r(out.get('queued_ms_per_step'), 5)
    line['timed_blocks'] = out.get('timed_blocks')
    line['blocks_reported'] = out.get('blocks_reported')
    if block_dt:
        line['timed_region_s'] = _r(sum(block_dt), 5)           # every timed block of K steps, barriers included
        line['reported_block_s'] = _r(out['ms_per_step'] * out['steps'] * 1e-3, 6)   # the median block: K steps
    ex = rf.get('executed') or {}
    line['roofline'] = dict(bound=rf.get('bound'), kernel=rf.get('kernel'), achieved=_r(rf.get('achieved'), 5), peak=rf.get('peak'),
                            unit=rf.get('unit'), frac=_r(rf.get('frac'), 6), traffic=rf.get('traffic'),
                            traffic_source=rf.get('traffic_source'), kernel_us=_r(rf.get('kernel_us'), 3),
                            executed_mfma_frac=_r(ex.get('frac'), 6),
                            whole_step_executed_frac=_r(rf.get('whole_step_executed_frac'), 6),
                            kernel_ms=rf.get('kernel_ms'))
    if cpu:
        ac = cpu.get('all_cores') or {}
        line['cpu_baseline'] = dict(value=_r(cpu.get('value'), 4), unit=cpu.get('unit'), cores=cpu.get('cores'), kind=cpu.get('kind'),
                                    sample=(cpu.get('sample') or '')[:150], host_cores=cpu.get('host_cores'),
                                    all_cores=dict(value=_r(ac.get('value'), 2), cores=ac.get('cores'), kind='port (minimum-work, OpenMP)')
                                    if 'value' in ac else None)
        if cpu.get('value'):
            line['gpu_over_cpu_1core'] = _r(out['value'] / cpu['value'], 1)
        if ac.get('value'):
            line['gpu_over_cpu_all_cores'] = _r(out['value'] / ac['value'], 2)
    else:
        line['cpu_baseline'] = None
    line['sequential_updates_per_s'] = _r(out.get('sequential_updates_per_s'), 1)
    hv = lat.get('host_visible') or {}
    line['host_visible_ms'] = _r(hv.get('median_ms'), 5)
    line['host_visible_p95_ms'] = _r(hv.get('p95_ms'), 5)
    line['device_resident_sync_ms'] = _r((lat.get('device_resident') or {}).get('median_ms'), 5)
    line['host_visible_resident_cov_ms'] = _r((lat.get('host_visible_resident_cov') or {}).get('median_ms'), 5)
    f1 = obj.get('frame_config3_one_call') or {}
    line['config3_frame_ms'] = _r(f1.get('median_ms'), 5)
    line['config3_frame_objects_staged_ahead_ms'] = _r((obj.get('frame_config3_one_call_objects_staged_ahead') or {}).get('median_ms'), 5)   # (orcvio_msckf_io_stage_object_tracks before the call)
    line['config3_frame_unchained_ms'] = _r((obj.get('frame_config3_one_call_unchained') or {}).get('median_ms'), 5)   # (ORCVIO_FRAME_CHAIN=0: the object solve behind the feature half, bit-identical to the two calls)
    line['config3_object_update_ms'] = _r((obj.get('resident') or {}).get('median_ms'), 5)
    oc = (obj.get('cpu_baseline') or {}).get('all_cores') or {}
    ac = (cpu or {}).get('all_cores') or {}
    if f1.get('median_ms') and oc.get('ms_per_update') and ac.get('value'):
        cpu_frame = 1e3 / ac['value'] + oc['ms_per_update'] + (oc.get('rows_ms') or 0.0)
        line['config3_frame_cpu_all_cores_ms'] = _r(cpu_frame, 4)
        line['config3_frame_gpu_over_cpu_all_cores'] = _r(cpu_frame / f1['median_ms'], 2)
    cf = out.get('configs') or {}
    line['configs_device_resident_ms'] = {k: _r((v.get('device_resident') or {}).get('median_ms'), 5) for k, v in cf.items()
                                          if isinstance(v, dict) and 'device_resident' in v} or None
    for k in ('stream_config1', 'stream_config5'):   # C++ harness (tests/cpp/stream_bench.cpp, a child process), one call per frame; beside it: the separate calls from C++, and both through ctypes
        st = out.get(k) or {}
        cpp = st.get('cpp') or {}
        line[k + '_frames_per_s'] = _r(st.get('frames_per_s'), 1)
        line[k + '_median_ms'] = _r((cpp.get('one_call_per_frame') or {}).get('median_ms'), 5)
        line[k + '_separate_calls_frames_per_s'] = _r((cpp.get('separate_calls') or {}).get('frames_per_s'), 1)
        line[k + '_ctypes_frames_per_s'] = _r((st.get('ctypes_one_call_per_frame') or {}).get('frames_per_s'), 1)
    if out.get('comm') is not None:
        line['comm'] = out['comm']
    line['detail'] = detail_file
    line['lib_sha16'] = out.get('lib_sha16')   # the library this run loaded (compare roofline.traffic_source.build)
    s = json.dumps(line)
    if len(s) > 4000:   # never again: drop the optional parts rather than outgrow the parser
        for k in ('configs_device_resident_ms', 'stream_config1_ctypes_frames_per_s', 'stream_config5_ctypes_frames_per_s', 'stream_config5_separate_calls_frames_per_s',
                  'stream_config5_median_ms'):
            line.pop(k, None)
        line['roofline'].pop('kernel_ms', None)
        s = json.dumps(line)
    return s


def emit(out, block_dt):
    """Detail to bench_detail.json (beside bench.py, and under gpurun_out/ when that exists), the compact contract line last on stdout."""
    detail_file = None
    for d in (os.path.join(ROOT, 'gpurun_out'), ROOT):
        if os.path.isdir(d) and os.access(d, os.W_OK):
            try:
                path = os.path.join(d, os.environ.get('ORCVIO_BENCH_DETAIL', 'bench_detail.json'))
                with open(path, 'w') as f:
                    json.dump(out, f, indent=1)
                detail_file = detail_file or os.path.relpath(path, ROOT)
            except OSError:
                pass
    print(f'bench.py: every side measurement (latency modes, configs, streams, objects, per-kernel roofline) is in {detail_file}', file=sys.stderr, flush=True)
    print(compact_line(out, block_dt, detail_file), flush=True)   # the one JSON line, last thing on stdout


def objects_section(upd, capi, synth, orc, np, win):
    """Config 3's object update (20 cars x 12 keypoints x 30 frames) and the north-star frame."""
    import ctypes as C
    N = win.N
    oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
    owin = synth.make_window(N=N, F=4, seed=0, flags=oflags, track_len=4)
    objs = synth.make_objects(owin, n_objects=20, seed=1, sigma_kp=0.004)
    ofl = capi.make_flags(oflags)
    ef, arr, keep = upd._object_tracks(objs, owin.R_b2c[0], owin.t_c_b[0], True, False, 0, False)   # marshalled once
    Pc = np.ascontiguousarray(owin.P)
    last = {}

    def call():
        o, res = upd._result(owin.n, 1)
        rc = upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(ofl), C.byref(ef), owin.N, arr, len(objs),
                                                       capi._d(Pc), C.byref(res))
        assert rc == 0
        last['g'] = (int(o['accept'][0]), int(res.stats[0]))
    lat_obj = timed_calls(call, 100, warm=5)
    objects = dict(percentiles(lat_obj), objects=20, accepted=last['g'][0], dof=last['g'][1],
                   what='orcvio_msckf_update_object_tracks: 20 cars x 12 keypoints x 30 frames, rows evaluated on the '
                        'device, host buffers (tracks + P) in, dx and P+ out')
    # per-stage device times of the object update (HIP events between the stages, median of 20 runs)
    upd.set_stage_profile(True)
    runs = []
    for _ in range(20):
        call()
        runs.append(upd.profile_stages())
    upd.set_stage_profile(False)
    objects['stage_ms'] = {name: round(float(np.median([r[i][1] for r in runs])), 5) for i, (name, _) in enumerate(runs[0])}
    # the north-star frame (config 3): the 400-feature update, then the 20-object update on the P+ it left
    # (SURVEY note N7), covariance and its square-root factor resident in HBM in between: tracks + poses in,
    # dx out, twice; the prior is restored outside the timed part
    fwin = synth.config_window(3)
    upd.cov_set(fwin.P)
    call_f, io = upd.make_io_call(fwin, resident_cov=True, want_P=False, commit=True)
    oo, ores = upd._result(owin.n, 1)
    ores.P_out = None

    def frame():
        call_f()
        rc = upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(ofl), C.byref(ef), owin.N, arr, len(objs), None, C.byref(ores))
        assert rc == 0
        rc = upd.lib.orcvio_msckf_cov_commit(upd.h)
        assert rc == 0

    def renew():   # the object update reuses the handle's arena: lay it out for the feature update again (outside the timed part)
        nonlocal call_f
        call_f, _ = upd.make_io_call(fwin, resident_cov=True, want_P=False, commit=True)

    def restore():
        upd.cov_set(fwin.P)
        renew()
    restore()
    lat_frame = timed_calls(frame, 100, warm=5, after=restore)
    objects['frame_config3'] = dict(percentiles(lat_frame), object_update_accepted=int(oo['accept'][0]),
                                    what='400-feature update (arena written in place) with its commit + 20-object update + commit, '
                                         'covariance and its factor resident in HBM: host tracks / poses in, dx out (twice)')
    # ... and in ONE call (orcvio_msckf_io_update_frame): the same two updates, the object tracks' compression -- which depends on
    # neither the prior nor the feature update -- running on its own stream beside the feature update's solve
    frame_call = {}

    def renew_frame(prefactor=False):
        upd.cov_set(fwin.P)
        if prefactor:
            upd.cov_prefactor()
            upd.sync()
        io = upd.io_begin(fwin.flags, fwin.N, fwin.F, int(fwin.obs_ptr[-1]), with_P=False)
        upd.io_fill(io, fwin, with_P=False)
        frame_call['c'], frame_call['o'] = upd.make_frame_call(fwin, oflags, objs, owin.R_b2c[0], owin.t_c_b[0], True, False, 0, False, True)
    renew_frame()
    def run_frame():
        frame_call['c']()
        frame_call['last'] = frame_call['o']   # (the result buffers of the call that ran: renew_frame makes new ones)
    lat_frame1 = timed_calls(run_frame, 100, warm=5, after=renew_frame)
    objects['frame_config3_one_call'] = dict(
        percentiles(lat_frame1), object_update_accepted=int(frame_call['last']()[1]['accept']),
        what='orcvio_msckf_io_update_frame: the same frame in one call -- feature update + commit, object update + commit, the object '
             'tracks\' compression (rows, structured QR, A\') on its own stream beside the feature update\'s solve, the object solve chained to '
             'the feature update\'s prior factor and M (M12 = M1 + L_a^T A\' L_a) on that stream too; equal to the two calls to rounding '
             '(tests/test_gpu_frame.py; ORCVIO_FRAME_CHAIN=0: bit for bit)')

    # ... with the object tracks staged AHEAD of the call (orcvio_msckf_io_stage_object_tracks, outside the timed part like the feature
    # tracks' io_fill: the object mapper's results are at hand before the frame's feature update starts)
    def renew_frame_staged():
        renew_frame()
        frame_call['c'].stage()
    renew_frame_staged()
    lat_frame_staged = timed_calls(run_frame, 100, warm=5, after=renew_frame_staged)
    objects['frame_config3_one_call_objects_staged_ahead'] = dict(
        percentiles(lat_frame_staged), object_update_accepted=int(frame_call['last']()[1]['accept']),
        prestaged_frames=int(upd.counters().get('prestaged_frames', 0)),
        what='as frame_config3_one_call, the object tracks scanned and packed into the pinned staging arena before the call '
             '(orcvio_msckf_io_stage_object_tracks): ~14 us of host time that otherwise stands between the tracks\' launch and the '
             'compression\'s; bit-identical results (tests/test_gpu_frame.py)')

    def renew_frame_pre():
        renew_frame(True)
    renew_frame_pre()
    objects['frame_config3_one_call_prefactored'] = dict(
        percentiles(timed_calls(run_frame, 100, warm=5, after=renew_frame_pre)),
        what='as frame_config3_one_call, the Cholesky of the frame\'s prior started ahead of the call')
    # ... and with ORCVIO_FRAME_CHAIN=0 (read at create): the object solve BEHIND the feature half, on the covariance it leaves -- the form that is
    # bit-identical to the two calls (the default chains the object solve to the feature update's prior factor and M: DESIGN.md 3.6)
    try:
        os.environ['ORCVIO_FRAME_CHAIN'] = '0'
        upc = capi.MsckfUpdater(device=upd.device if hasattr(upd, 'device') else 0, max_clones=32, max_features=2048, max_observations=65536)
        os.environ.pop('ORCVIO_FRAME_CHAIN', None)
        chained = {}

        def renew_chained():
            upc.cov_set(fwin.P)
            io2 = upc.io_begin(fwin.flags, fwin.N, fwin.F, int(fwin.obs_ptr[-1]), with_P=False)
            upc.io_fill(io2, fwin, with_P=False)
            chained['c'], chained['o'] = upc.make_frame_call(fwin, oflags, objs, owin.R_b2c[0], owin.t_c_b[0], True, False, 0, False, True)
        renew_chained()

        def run_chained():
            chained['c']()
            chained['last'] = chained['o']
        lat_c = timed_calls(run_chained, 100, warm=5, after=renew_chained)
        ref_o = frame_call['last']()[1]
        got_o = chained['last']()[1]
        objects['frame_config3_one_call_unchained'] = dict(
            percentiles(lat_c), object_update_accepted=int(got_o['accept']),
            dx_rel_diff_to_the_default_one_call_form=float(np.linalg.norm(got_o['dx'] - ref_o['dx']) / max(np.linalg.norm(ref_o['dx']), 1e-300)),
            what='orcvio_msckf_io_update_frame with ORCVIO_FRAME_CHAIN=0: the object solve behind the feature half\'s commit, on the factor it leaves')
        upc.close()
    except Exception as e:   # (a side measurement: never in the way of the contract line)
        os.environ.pop('ORCVIO_FRAME_CHAIN', None)
        objects['frame_config3_one_call_unchained'] = dict(error=repr(e))
    restore()
    # the same frame with the prior factored ahead (orcvio_msckf_cov_prefactor when the image arrives)

    def restore_frame_prior():
        upd.cov_set(fwin.P)
        upd.cov_prefactor()
        upd.sync()
        renew()
    restore_frame_prior()
    lat_frame_pre = timed_calls(frame, 100, warm=5, after=restore_frame_prior)
    objects['frame_config3_prefactored'] = dict(
        percentiles(lat_frame_pre), what='as frame_config3, the Cholesky of the frame\'s prior started ahead of the first '
                                         'update (outside the timed part: it runs while the front end tracks the image)')
    # the object update alone in that mode (prior and its factor resident)
    upd.cov_set(fwin.P)
    renew()
    call_f()

    def obj_res():
        rc = upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(ofl), C.byref(ef), owin.N, arr, len(objs), None, C.byref(ores))
        assert rc == 0
    lat_or = timed_calls(obj_res, 100, warm=5)
    objects['resident'] = dict(percentiles(lat_or), what='the object update with the prior and its square-root factor '
                                                         'resident (no P upload, no Cholesky of P), P+ left in HBM')
    upd.set_stage_profile(True)
    runs = []
    for _ in range(20):
        obj_res()
        runs.append(upd.profile_stages())
    upd.set_stage_profile(False)
    objects['resident']['stage_ms'] = {name: round(float(np.median([r[i][1] for r in runs])), 5) for i, (name, _) in enumerate(runs[0])}
    if orc is not None:   # the same object update on one host core (oracle/object_oracle.c)
        t_rows = time.perf_counter()
        blocks_c = [orc.object_rows_c(o, owin.R_b2c[0], owin.t_c_b[0], True, False, 0) for o in objs]
        t_rows = time.perf_counter() - t_rows
        cu = orc.objects_update_c(oflags, owin.N, blocks_c, owin.P)
        objects['cpu_baseline'] = dict(ms_per_update=(t_rows + cu['seconds']) * 1e3, cores=1, kind='port', accepted=cu['accept'],
                                       what='rows (C restatement of the CameraLM / ObjectLM functors) + per-object '
                                            'Householder projection on dense rows x n blocks + QR of the stack + update')
        try:   # the minimum-work port of the update (oracle/object_fast.c: Schur-complement projection from the 7 non-zeros per row,
               # objects in parallel, square-root solve); the rows' evaluation (single-threaded C) is timed beside it
            host = os.cpu_count() or 1
            sweep = {}
            for t in sorted({t for t in (1, 4, 8, 16, 20, 32) if t <= host}):
                orc.objects_update_fast(oflags, owin.N, blocks_c, owin.P, threads=t)
                sweep[t] = min(orc.objects_update_fast(oflags, owin.N, blocks_c, owin.P, threads=t)['seconds'] for _ in range(3)) * 1e3
            best = min(sweep, key=sweep.get)
            objects['cpu_baseline']['all_cores'] = dict(
                ms_per_update=sweep[best], rows_ms=t_rows * 1e3, cores=best, ms_by_threads={str(t): round(v, 3) for t, v in sweep.items()},
                kind='port (minimum-work algorithm, OpenMP; best team of the sweep)',
                what='the update alone from the evaluated rows: H\'^T H\' = X^T X - Y^T Y with Y = R^-T (H_f^T X) per object (QR of H_f, 7 '
                     'non-zeros per row), one object per thread, square-root solve, joint gate; same decision and update as the '
                     'literal port (tests/test_oracle_objects.py)')
        except Exception as e:
            objects['cpu_baseline']['all_cores'] = dict(error=repr(e))
    upd.upload(win)   # the feature tracks again for what follows
    return objects


if __name__ == '__main__':
    main()

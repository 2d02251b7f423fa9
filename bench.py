#!/usr/bin/env python3
"""bench.py -- EKF (MSCKF) updates/s on the BASELINE.json configuration, one process per GPU.

A "step" is one complete measurement update (Jacobians -> nullspace -> gate -> stacked H ->
Gram compression -> Kalman solve -> dx, P+) on inputs already resident in HBM.
N=1: config 2 (30 clones x 400 features x 30 observations).  N>1 (weak scaling): every rank
holds its own shard of 400 features of one joint update; per step the ranks all-gather their
compressed blocks over RCCL and each performs the (replicated) Kalman solve.
"""
import gc
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector / matrix peak (not listed in MI355X_MICROARCH.md)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--clones', type=int, default=30)
    ap.add_argument('--features', type=int, default=400)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from orcvio_amd import capi, synth

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU path')
    torch.cuda.set_device(local_rank)
    # ORCVIO_BENCH_FORCE_DIST=1 drives the multi-GPU code path (RCCL all-gather included) with world size 1
    use_dist = world > 1 or os.environ.get('ORCVIO_BENCH_FORCE_DIST') == '1'
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    N, F = args.clones, args.features
    # weak scaling: one joint update of F*world tracks, dealt across the ranks
    from orcvio_amd import sharding
    full = synth.make_window(N=N, F=F * world, seed=0, flags=synth.Flags(use_larvio=1))
    win, _ = sharding.shard_window(full, rank, world)
    upd = capi.MsckfUpdater(device=local_rank, max_clones=max(32, N), max_features=max(2048, win.F),
                            max_observations=max(65536, int(win.obs_ptr[-1])))
    upd.upload(win)
    # one explicit (non-default) stream carries the kernels AND the collective, so RCCL is ordered after the
    # rank's block is written and before the solve reads the gathered blocks
    tstream = torch.cuda.Stream()
    stream = tstream.cuda_stream
    gathered = None
    if use_dist:
        _, ne = upd.block_ptr()
        gathered = torch.empty(world * ne, dtype=torch.float64, device='cuda')
        local = torch.empty(ne, dtype=torch.float64, device='cuda')

    def step():
        with torch.cuda.stream(tstream):
            if not use_dist:
                upd.run_update(stream)
            else:
                upd.run_local_to(local.data_ptr(), stream)          # this rank's compressed block -> send buffer
                dist.all_gather_into_tensor(gathered, local)        # the one data-path collective (RCCL over xGMI)
                upd.run_finish(gathered.data_ptr(), world, stream)  # rank-ordered sum + replicated Kalman solve

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()   # the timed region is tens of ms: an interpreter collection in the middle of it would be most of it
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3

    out = None
    if rank == 0:
        M = N
        W = algorithmic_flops(N, F * world, M)
        # roofline of the dominant kernel, timed live with HIP events on the launch stream
        prof = upd.profile(reps=20, stream=stream)
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
            'k_finish': 1.0 * n * (n + 1) * (n + 1),
        }
        # k_front = the tracks, the compression and chol(P) in one launch (the default whenever they are co-resident)
        kflops['k_front'] = kflops['k_feature'] + kflops['k_gram'] + kflops['k_potrf(P)']
        # k_potrf(P) runs on a side stream, overlapped with k_feature/k_gram: not on the critical path
        crit = {k: v for k, v in prof.items() if k != 'k_potrf(P)'}
        dom = max(crit, key=crit.get)
        achieved = kflops[dom] / (prof[dom] * 1e-3) / 1e12
        # HBM traffic of the dominant kernel: PMC counters (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes) of the
        # committed profile of this round, bytes per launch; null if that profile does not list the kernel
        traffic = None
        executed = None
        try:
            import glob
            pm = json.load(open(sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))[-1]))['kernels']
            key = {'k_feature': 'k_feature<3>', 'k_potrf(M)': 'k_potrf_reg<16>', 'k_potrf_solve(M)': 'k_potrf_solve<16>',
                   'k_trsm': 'k_trsm_lds', 'k_finish': 'k_finish_sqrt', 'k_gram': 'k_gram_pair', 'k_assemble': 'k_assemble_A',
                   'k_front': 'k_front<3, 16>'}.get(dom)
            if key in pm and N == 30 and F == 400:
                traffic = 1024.0 * (pm[key]['FETCH_SIZE_KB_median'] + pm[key]['WRITE_SIZE_KB_median'])
                if 'SQ_INSTS_VALU_MFMA_MOPS_F64_median' in pm[key]:   # flops the matrix cores actually executed
                    executed = 512.0 * pm[key]['SQ_INSTS_VALU_MFMA_MOPS_F64_median'] / (prof[dom] * 1e-3) / 1e12
        except Exception:
            traffic = None
        roofline = dict(bound='mfma', kernel=dom, achieved=achieved, peak=FP64_PEAK_TFLOPS, unit='TFLOP/s',
                        frac=achieved / FP64_PEAK_TFLOPS, traffic=traffic,
                        executed_mfma_tflops=executed,
                        note='achieved = algorithmic FP64 work of the reference algorithm (SURVEY 8d dense minimum) / kernel '
                             'time; the kernel reaches the same result with far fewer executed flops (structured gate, fused '
                             'compression), see executed_mfma_tflops and DESIGN.md 6: the path is latency-bound, not MFMA-bound',
                        kernel_ms={k: round(v, 5) for k, v in prof.items()},
                        kernel_tflops={k: round(kflops[k] / (prof[k] * 1e-3) / 1e12, 4) for k in prof},
                        whole_update_tflops=W['total'] / (ms * 1e-3) / 1e12)
        # the same update with host buffers in and (dx, P+, gamma, accept) out through the one-shot C-ABI call: reported
        # beside `value`, never as `value` (PCIe + four synchronous copies per update)
        host_inclusive = None
        if world == 1:
            for _ in range(5):
                upd.update_features(win)
            gc.collect()
            gc.disable()   # (a collection of the interpreter in the middle of a call costs tens of ms: not the library's)
            th = time.perf_counter()
            reps_h = 50
            for _ in range(reps_h):
                upd.update_features(win)
            th = (time.perf_counter() - th) / reps_h
            gc.enable()
            host_inclusive = dict(updates_per_s=1.0 / th, ms_per_update=th * 1e3,
                                  what='orcvio_msckf_update_features: host tracks + P in, dx, P+, gamma, accept out')
        # config 3 adds 20 objects x 12 keypoints to the same window: the object update (a second EKF update per frame in
        # the reference, src/orcvio.cpp:2154-2193) from object tracks, host buffers in and out; reported beside the metric
        objects = None
        if world == 1 and N == 30:
            try:
                oflags = synth.Flags(use_larvio=0, use_left_perturbation=0)
                owin = synth.make_window(N=N, F=4, seed=0, flags=oflags, track_len=4)
                objs = synth.make_objects(owin, n_objects=20, seed=1, sigma_kp=0.004)
                import ctypes as C
                ofl = capi.make_flags(oflags)
                ef, arr, keep = upd._object_tracks(objs, owin.R_b2c[0], owin.t_c_b[0], True, False, 0, False)   # marshalled once
                Pc = np.ascontiguousarray(owin.P)

                def call():
                    out, res = upd._result(owin.n, 1)
                    rc = upd.lib.orcvio_msckf_update_object_tracks(upd.h, C.byref(ofl), C.byref(ef), owin.N, arr, len(objs),
                                                                   capi._d(Pc), C.byref(res))
                    assert rc == 0
                    return int(out['accept'][0]), int(res.stats[0])
                for _ in range(5):
                    g = call()
                gc.collect()
                gc.disable()
                to = time.perf_counter()
                for _ in range(20):
                    g = call()
                to = (time.perf_counter() - to) / 20
                gc.enable()
                objects = dict(ms_per_update=to * 1e3, objects=20, accepted=g[0], dof=g[1],
                               what='orcvio_msckf_update_object_tracks: 20 cars x 12 keypoints x 30 frames, rows evaluated on the '
                                    'device, host buffers in, dx and P+ out')
                upd.upload(win)   # the feature tracks again for what follows
            except Exception as e:   # never let the side measurement break the metric line
                objects = dict(error=str(e))
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as orc   # checker used as the reported CPU baseline ("port")
            reps = 2
            t = []
            for _ in range(reps):
                t.append(orc.msckf_update(win, want_blocks=False, want_K=False)['seconds'])
            cpu = dict(value=1.0 / min(t), unit='updates/s', cores=1, kind='port',
                       sample=f'{reps} full updates of the same workload, best of {reps} ({min(t):.2f} s each), '
                              'single-threaded plain-C restatement of the reference algorithm')
        # Weak scaling: every rank keeps one 400-feature shard, a step is ONE joint update of 400 x world features
        # (rank-local tracks + compression, one RCCL all-gather, replicated solve).  `value` is the whole-job
        # aggregate in the metric's own unit -- 400-feature update shards processed per second by all ranks =
        # world x joint updates/s -- so that value(N) / (N value(1)) is the usual weak-scaling efficiency T(1)/T(N);
        # the joint-update rate is reported beside it.
        out = dict(metric='EKF updates/sec, 30 clones x 400 feats', value=world * args.steps / dt, unit='updates/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms, higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
                   config=dict(workload='config2: synthetic 30-clone window, 400 point features x 30 observations '
                                        'per GPU (22 800 stacked rows x 202 columns), LARVIO Jacobians',
                               clones=N, features_per_gpu=F, observations_per_feature=N,
                               unit='one update of 30 clones x 400 features; at N GPUs one step is a joint update of '
                                    '400 N features = N units',
                               joint_updates_per_s=args.steps / dt, features_per_joint_update=F * world,
                               parallelism=f'features sharded over {world} GPU(s), all-gather of compressed blocks'),
                   roofline=roofline, cpu_baseline=cpu, host_inclusive=host_inclusive, objects_update=objects)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    upd.close()
    if rank == 0:
        sys.stdout.flush()
        try:   # RCCL writes its version banner through C stdio: flush that buffer first so that the JSON line comes last
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)   # the one JSON line, last thing on stdout


if __name__ == '__main__':
    main()

"""VERDICT r2 "next" 2(b): a child process creating a one-rank RCCL communicator on a GPU while its parent holds one sat for ever
now and then (tests/test_host_shim.py, round 2).  This reproduces the situation in a loop with NCCL_DEBUG=INFO and a bounded
wait in the child (ORCVIO_COMM_TIMEOUT_S), in three settings: (A) the parent (torch + the handle's communicator) holds a
communicator, (B) the parent holds the GPU (a handle) but no communicator, (C) the parent never touches the GPU.
usage: python scripts/gpu_comm_hang_repro.py [launches_A] [launches_B] [launches_C] [child_timeout_s]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nA = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nB = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nC = int(sys.argv[3]) if len(sys.argv) > 3 else 20
child_to = float(sys.argv[4]) if len(sys.argv) > 4 else 25.0
LIB = os.path.join(ROOT, 'orcvio_amd', 'lib')
out_dir = os.path.join(ROOT, 'gpurun_out', 'comm_hang')
os.makedirs(out_dir, exist_ok=True)
exe = os.path.join(out_dir, 'comm_probe')
subprocess.check_call(['g++', '-std=c++17', '-O1', '-o', exe, os.path.join(ROOT, 'tests', 'cpp', 'comm_probe.cpp'), '-L', LIB, '-lorcvio_msckf',
                       f'-Wl,-rpath,{LIB}', '-lpthread'])


def run_children(tag, n, extra_env=None):
    res = []
    for i in range(n):
        log = os.path.join(out_dir, f'{tag}_{i}.nccl.log')
        env = dict(os.environ, NCCL_DEBUG='INFO', NCCL_DEBUG_FILE=log, ORCVIO_COMM_TIMEOUT_S=str(child_to), HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.update(extra_env or {})
        t = time.time()
        p = subprocess.Popen([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        try:
            so, _ = p.communicate(timeout=child_to * 3 + 30)
            rc = p.returncode
        except subprocess.TimeoutExpired:
            p.kill()
            so, _ = p.communicate()
            rc = -9
        dt = time.time() - t
        rec = dict(i=i, rc=rc, seconds=round(dt, 2))
        if rc != 0:
            rec['stdout'] = so[-3000:]
            try:
                rec['nccl_tail'] = open(log).read()[-3000:]
            except Exception:
                pass
        else:
            try:
                os.remove(log)
            except Exception:
                pass
        res.append(rec)
        print(tag, rec['i'], rc, rec['seconds'], flush=True)
    return res


summary = {}
# (C) first: nothing in this process has touched the GPU
summary['C_parent_without_gpu'] = run_children('C', nC)
import torch  # noqa: F401  (the parent of the test-suite has torch loaded: its librccl, its libamdhip64)
from orcvio_amd import capi
upd = capi.MsckfUpdater(device=0, max_clones=8, max_features=64, max_observations=1024)
summary['B_parent_holds_gpu'] = run_children('B', nB)
upd.comm_init(capi.comm_unique_id(), 0, 1)
upd.comm_barrier()
summary['A_parent_holds_communicator'] = run_children('A', nA)
# (A2) the same with the parent's proxy / bootstrap threads busy: a collective every now and then
summary['A2_parent_runs_collectives'] = []
for k in range(max(1, nA // 6)):
    for _ in range(20):
        upd.comm_barrier()
    summary['A2_parent_runs_collectives'] += run_children('A2_%d' % k, 1)
upd.close()
# (D) what the test-suite's parent actually looked like when round 2 saw the hang: it had CREATED AND DESTROYED communicators
# (tests/test_gpu_comm.py's module fixture is closed before tests/test_host_shim.py runs) and run many updates
for _ in range(3):
    u2 = capi.MsckfUpdater(device=0, max_clones=8, max_features=64, max_observations=1024)
    u2.comm_init(capi.comm_unique_id(), 0, 1)
    u2.comm_barrier()
    u2.close()
summary['D_parent_destroyed_communicators'] = run_children('D', nA)
# (E) the executable of the test itself (updates through k_front first, then the communicator), bounded the same way
exe_t = os.path.join(out_dir, 'test_host_gpu')
orc = os.path.join(ROOT, 'oracle')
subprocess.check_call(['g++', '-std=c++17', '-O1', '-o', exe_t, os.path.join(ROOT, 'tests', 'cpp', 'test_host_gpu.cpp'), '-L', LIB, '-lorcvio_msckf',
                       f'-Wl,-rpath,{LIB}', '-L', orc, '-lorcoracle', f'-Wl,-rpath,{orc}', '-lm', '-lpthread'])
exe = exe_t
summary['E_test_host_gpu_executable'] = run_children('E', nA)
stat = {k: dict(launches=len(v), failed=sum(1 for r in v if r['rc'] != 0), slowest=max((r['seconds'] for r in v), default=0),
                median=sorted(r['seconds'] for r in v)[len(v) // 2] if v else 0) for k, v in summary.items()}
json.dump(dict(stat=stat, runs=summary), open(os.path.join(ROOT, 'gpurun_out', 'r3_comm_hang.json'), 'w'), indent=1)
print(json.dumps(stat, indent=1))

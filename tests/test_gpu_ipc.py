"""The sharded entry points with TWO RANKS on the one GPU of the test box: the handle's second transport
(ORCVIO_COMM_TRANSPORT=ipc, csrc/capi_ipc.inc) -- every rank stores its compressed block straight into its peers' gather buffers
(HIP IPC), flags in a shared-memory segment -- lets two processes share a device, which RCCL refuses.  Two fresh child processes
(started before they touch the GPU; nothing is exec'ed afterwards) run tests/ipc_rank_worker.py: 20 sharded feature updates of a
2 x 200-track window against the single-call oracle, six queued staged updates, an all-rejected share, a refused share
(ORCVIO_ERR_INVALID on its rank, ORCVIO_ERR_PEER on the other), sharded object updates, barrier and max.  Both ranks must pass and
return bit-identical results (rank-ordered sum, replicated solve)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(world, extra_env=None, timeout=900, mode=None):
    env = dict(os.environ, ORCVIO_COMM_TRANSPORT='ipc', ORCVIO_COMM_TIMEOUT_S='120', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.update(extra_env or {})
    uid = os.urandom(128).hex()   # (what orcvio_msckf_comm_unique_id returns under this transport: 128 random bytes)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'ipc_rank_worker.py'), str(r), str(world), uid] + ([mode] if mode else []),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, so, se))
    return outs


def test_two_ranks_on_one_device(built):
    outs = _run_ranks(2)
    res = []
    for rc, so, se in outs:
        lines = [ln for ln in so.splitlines() if ln.startswith('RESULT ')]
        assert lines, (rc, so[-2000:], se[-3000:])
        res.append(json.loads(lines[-1][7:]))
    for r, (rc, so, se) in zip(res, outs):
        failed = [c for c in r['checks'] if not c['ok']]
        assert not failed and rc == 0 and r['passed'], (failed, se[-2000:])
    assert len(res[0]['checks']) >= 30
    # the replicated solve: every rank holds the same bits
    assert res[0]['feature_digests'] == res[1]['feature_digests']
    assert res[0]['staged_digest'] == res[1]['staged_digest']
    assert res[0]['object_digest'] == res[1]['object_digest']


def test_a_slipped_update_counter_is_a_loud_error_on_every_rank(built):
    """the ipc transport's sequence numbers used to be free-running host counters nothing checked; a rank that took
    a rank-local early return stayed one count behind and summed its peers' PREVIOUS blocks.  Now every block carries its update's
    number and the counters advance at the top of each call: a rank made to slip (debug hook) gets ORCVIO_ERR_TIMEOUT, its peer
    ORCVIO_ERR_PEER, and nobody has an update to commit."""
    outs = _run_ranks(2, extra_env=dict(ORCVIO_IPC_WAIT_S='2', ORCVIO_COMM_TIMEOUT_S='30'), timeout=300, mode='skew')
    for rc, so, se in outs:
        lines = [ln for ln in so.splitlines() if ln.startswith('RESULT ')]
        assert lines, (rc, so[-2000:], se[-3000:])
        r = json.loads(lines[-1][7:])
        assert r['passed'] and rc == 0, (r, se[-2000:])


def test_unique_id_of_the_ipc_transport(built):
    """orcvio_msckf_comm_unique_id under ORCVIO_COMM_TRANSPORT=ipc: 128 random bytes (the name of the shared segment is derived from
    them), no call into RCCL -- ORCVIO_RCCL_LIB points at nothing and the call still succeeds."""
    code = ("import os, sys\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "from orcvio_amd import capi\n"
            "a, b = capi.comm_unique_id(), capi.comm_unique_id()\n"
            "assert len(a) == 128 and a != b\n"
            "print('RESULT ok')\n")
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ, ORCVIO_COMM_TRANSPORT='ipc', ORCVIO_RCCL_LIB='/nonexistent/librccl.so'), timeout=120)
    assert 'RESULT ok' in p.stdout, p.stdout[-1000:] + p.stderr[-2000:]


def test_one_rank_over_the_ipc_transport(built):
    """The world-size-1 tests of tests/test_gpu_comm.py (sharded feature / object updates equal to the one-shot calls, barrier and
    max, a refused share) with ORCVIO_COMM_TRANSPORT=ipc: the same entry points over the second transport, in a child pytest."""
    env = dict(os.environ, ORCVIO_COMM_TRANSPORT='ipc', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_comm.py'), '-x', '-q', '-m', 'gpu', '-k',
                        'world_1 or barrier or refused_share or need_a_communicator', '-p', 'no:cacheprovider'],
                       capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    tail = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ''
    assert p.returncode == 0 and ' passed' in tail and 'failed' not in tail, p.stdout[-2000:] + p.stderr[-2000:]


def test_three_ranks_on_one_device_random_windows(built):
    """A bounded slice of scripts/gpu_soak_ipc.py (1 888 random windows through two / three processes on one GPU this round, no failure):
    THREE ranks (an odd deal of the tracks), twelve seconds of random one-shot feature updates, queued staged updates and object
    updates on the ranks' shares against the single-call oracle; identical bits on every rank."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'gpu_soak_ipc.py'), '12', '777', '3'], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    txt = p.stdout[p.stdout.index('{'):] if '{' in p.stdout else ''
    assert txt, p.stdout[-1500:] + p.stderr[-1500:]
    d = json.loads(txt)
    assert d['world'] == 3 and d['failures'] == 0 and d['identical_results_on_every_rank'], d
    assert all(r['feature_windows'] + r['object_windows'] >= 10 for r in d['ranks']), d

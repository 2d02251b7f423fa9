"""The ipc transport's protocol code on the CPU: tests/cpp/test_ipc_protocol.cpp runs the functions of
orcvio_amd/csrc/host/ipc_protocol.hpp -- the ones capi_ipc.inc and k_gram_reduce are compiled from -- with 4 .. 8 threads standing in
for ranks: the rank-ordered sum, the two-generation slots under ranks of different speeds, a slipped counter, a missing rank."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ipc_protocol_with_simulated_ranks(tmp_path):
    exe = str(tmp_path / 'test_ipc_protocol')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-pthread', '-Wall', '-o', exe, os.path.join(ROOT, 'tests', 'cpp', 'test_ipc_protocol.cpp')])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'test_ipc_protocol: ok' in r.stdout, r.stdout + r.stderr


def test_the_library_is_compiled_from_the_same_protocol_header():
    ipc = open(os.path.join(ROOT, 'orcvio_amd', 'csrc', 'capi_ipc.inc')).read()
    kern = open(os.path.join(ROOT, 'orcvio_amd', 'csrc', 'msckf_kernels.hpp')).read()
    assert '#include "host/ipc_protocol.hpp"' in ipc and 'ipc_slots_allreduce_max(' in ipc and 'ipc_slots_sum_dofs(' in ipc and 'ipc_slot_offset(' in ipc
    assert '#include "host/ipc_protocol.hpp"' in kern and 'rank_ordered_sum(parts, nparts, part_stride' in kern

"""Builds and runs the C++ tests of the host mirror (orcvio_amd/csrc/host/orcvio_msckf_host.hpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'orcvio_amd', 'lib')


def _build(src, exe, extra=()):
    cmd = ['g++', '-std=c++17', '-O1', '-Wall', '-o', exe, os.path.join(ROOT, 'tests', 'cpp', src), '-L', LIB,
           '-lorcvio_msckf', f'-Wl,-rpath,{LIB}'] + list(extra)
    subprocess.check_call(cmd)


def test_host_shim_flatten_and_object_row_layout(built, tmp_path):
    """No device call: container flattening and the row re-indexing the reference pins in
    src/tests/test_state_update.cpp:16-103."""
    exe = str(tmp_path / 'test_host_shim')
    _build('test_host_shim.cpp', exe)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'host shim ok' in out.stdout


@pytest.mark.gpu
def test_host_backend_end_to_end(built, tmp_path):
    """std::map containers -> MsckfBackend::msckfUpdate on the GPU vs the C oracle (both call sites)."""
    exe = str(tmp_path / 'test_host_gpu')
    orc = os.path.join(ROOT, 'oracle')
    _build('test_host_gpu.cpp', exe, ['-L', orc, '-lorcoracle', f'-Wl,-rpath,{orc}', '-lm'])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'host gpu ok' in out.stdout

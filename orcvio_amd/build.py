"""In-tree build of the HIP library (hipcc --offload-arch=gfx950) -- no JIT cache."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_DIR = os.path.join(_HERE, 'lib')
LIB = os.path.join(LIB_DIR, 'liborcvio_msckf.so')
SOURCES = ['msckf_capi.hip']
DEPS = ['msckf_capi.hip', 'msckf_kernels.hpp', 'msckf_math.hpp', 'object_rows.hpp', 'triangulate.hpp', 'cov_ops.hpp', 'ekf_rows.hpp',
        os.path.join('..', '..', 'include', 'orcvio_msckf.h')]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compiles every HIP source for gfx950 into orcvio_amd/lib/liborcvio_msckf.so."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared', '-o', LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB

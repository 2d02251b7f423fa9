"""In-tree build of the HIP library (hipcc --offload-arch=gfx950) -- no JIT cache."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_DIR = os.path.join(_HERE, 'lib')
LIB = os.path.join(LIB_DIR, 'liborcvio_msckf.so')
LIB_DBG = os.path.join(LIB_DIR, 'liborcvio_msckf_dbg.so')   # the same sources + the orcvio_msckf_debug_* test hooks
SOURCES = ['msckf_capi.hip']
# every file the one translation unit is made of: the C-ABI sections (capi_*.inc), the kernel headers, the public header
DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.hpp', '.inc'))) + [os.path.join('..', '..', 'include', 'orcvio_msckf.h')]


def _stale(lib) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def source_sha16() -> str:
    """sha256 (first 16 hex digits) over the names and contents of every file the library is built from (DEPS).  hipcc's output is
    not reproducible byte for byte (two builds of the same sources differ), so a profile records this beside the binary's hash:
    bench.py accepts counters taken on another BUILD of the same SOURCES."""
    import hashlib
    hsh = hashlib.sha256()
    for d in DEPS:
        hsh.update(os.path.basename(d).encode() + b'\0')
        with open(os.path.join(CSRC, d), 'rb') as f:
            hsh.update(f.read())
    return hsh.hexdigest()[:16]


def build_library(force: bool = False, verbose: bool = False, debug_hooks: bool = False) -> str:
    """Compiles every HIP source for gfx950 into orcvio_amd/lib/liborcvio_msckf.so (the product: the C-ABI of
    include/orcvio_msckf.h and nothing else), or with debug_hooks the diagnostics build liborcvio_msckf_dbg.so, which adds the
    orcvio_msckf_debug_* test hooks (-DORCVIO_DEBUG_HOOKS)."""
    lib = LIB_DBG if debug_hooks else LIB
    if not force and not _stale(lib):
        return lib
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-shared', '-o', lib] + \
          (['-DORCVIO_DEBUG_HOOKS'] if debug_hooks else []) + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return lib

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("GFN_ASAN_HOST_LIB"):
        _use_sanitizer_build_of_the_host_code(os.environ["GFN_ASAN_HOST_LIB"])


def _use_sanitizer_build_of_the_host_code(path):
    """tests/test_sanitizers.py: this pytest process runs under LD_PRELOAD=libasan, and the product's HOST C++ (octree build, ProcOctree)
    is to be the AddressSanitizer + UBSan build of oracle/Makefile's `asan` target (octree_build.cpp + octree_proc.cpp + a stub of the
    runtime's error plumbing; no HIP objects).  gfnerf_amd._lib.lib() then hands out a proxy that resolves the host entry points in
    that library and refuses everything else -- test plumbing only: the product never loads it."""
    import ctypes as C

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import _lib

    handle = C.CDLL(path)

    class _HostOnly:
        def __getattr__(self, name):
            if name not in _lib.SIGNATURES or not hasattr(handle, name):
                raise _lib.GfnError(f"{name} is not part of the sanitizer build of the host code ({path})")
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = _lib.SIGNATURES[name]
            return fn

    _lib._lib = _HostOnly()


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
    return dict(np.load(path))


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as _o  # oracle/oracle.py -- the CPU checker (test infrastructure)

    _o.lib()
    return _o

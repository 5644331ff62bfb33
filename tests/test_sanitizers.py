"""the host C++ of the product (csrc/octree_build.cpp, csrc/octree_proc.cpp: 843 lines) and the C oracle under
AddressSanitizer + UndefinedBehaviorSanitizer.  `make -C oracle asan` builds both with -fsanitize=address,undefined (the host C++ linked
with a stub of the runtime's error plumbing: oracle/asan_host_stub.cpp); a child pytest under LD_PRELOAD=libasan then runs the oracle-golden
tests and the octree build / ProcOctree tests against those builds (tests/conftest.py swaps the libraries in when GFN_ASAN_HOST_LIB /
GFN_ORACLE_LIB are set).  CPU box only; skipped where gcc has no libasan.  (GPU code: sanitizers are not available on the pool.)"""
import os
import shutil
import subprocess
import sys

import pytest

from helpers import ROOT


def _libasan():
    gcc = shutil.which("gcc")
    if not gcc or not shutil.which("make"):
        return None
    p = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(p) if p and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc has no libasan here")
def test_host_code_and_oracle_are_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    d = os.path.join(ROOT, "oracle", "_asan")
    env = dict(os.environ, LD_PRELOAD=_libasan(), GFN_ASAN_HOST_LIB=os.path.join(d, "libgfn_host_asan.so"), GFN_ORACLE_LIB=os.path.join(d, "libgfn_oracle_asan.so"),
               GFN_NO_REBUILD="1", OMP_NUM_THREADS="4",
               # python itself leaks by design and is not instrumented: no leak report; everything else aborts the child on the first finding
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    tests = [os.path.join(ROOT, "tests", t) for t in ("test_octree_pins.py", "test_octree_proc.py", "test_oracle_golden.py")]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + tests, env=env, capture_output=True, text=True,
                       timeout=1500, cwd=ROOT)
    tail = (p.stdout + p.stderr)[-4000:]
    assert p.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in p.stdout and "failed" not in p.stdout, tail

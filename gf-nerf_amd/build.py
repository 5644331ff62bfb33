"""Builds csrc/ into libgfnerf_amd.so (in-tree) with hipcc for gfx950.

    python gf-nerf_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the tree.
-ffp-contract=off is required: the sampler and the hash kernels are bit-compared
against the CPU oracle, which is built the same way.
"""
import concurrent.futures as cf
import contextlib
import fcntl
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libgfnerf_amd.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIP_SOURCES = ["runtime.hip", "hash3d.hip", "lm_lookup.hip", "sampler.hip", "mlp.hip", "field.hip", "field_x3.hip", "fused_fwd.hip", "wgrad_x3.hip", "render.hip", "batch.hip", "grad_part.hip", "octree_proc.hip"]
CXX_SOURCES = ["octree_build.cpp", "octree_proc.cpp"]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall",
               "-Wno-unused-function", "-I", INCLUDE]
CXX_FLAGS = ["-O2", "-fPIC", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-Wall", "-I", INCLUDE]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 extension cannot be built")


def _key(deps, flags):
    """Content key of an object: its source, the headers, the flags.  (File times are not used: a tree copied to another machine
    keeps its contents, not necessarily its mtimes, and must not trigger a rebuild there.)"""
    h = hashlib.sha256(" ".join("<include>" if f == INCLUDE else f for f in flags).encode())  # not the tree's location
    for d in deps:
        with open(d, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


def _stale(out, key):
    kf = out + ".key"
    return not (os.path.exists(out) and os.path.exists(kf) and open(kf).read() == key)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return r.stderr


@contextlib.contextmanager
def _build_lock():
    """One builder at a time (N ranks starting together would otherwise compile into the same objects and link over each other)."""
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def build_extension(force=False, verbose=False):
    """Incremental: compiles the sources whose object is older than the source or a header, links when an object or the source
    LIST changed; the new library is linked beside the old one and moved into place atomically."""
    with _build_lock():
        return _build_locked(force, verbose)


def _build_locked(force, verbose):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("gfn_common.h", "x3_device.h", "hash_device.h")]
    headers += [os.path.join(INCLUDE, "gfnerf_amd.h")]
    hipcc = _hipcc()
    jobs, objs, keys = [], [], []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", "_hip.o"))
        k = _key([s] + headers, HIPCC_FLAGS)
        objs.append(o); keys.append(k)
        if force or _stale(o, k):
            jobs.append((o, k, [hipcc] + HIPCC_FLAGS + ["-c", s, "-o", o]))
    for src in CXX_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".cpp", ".o"))
        k = _key([s] + headers, CXX_FLAGS)
        objs.append(o); keys.append(k)
        if force or _stale(o, k):
            jobs.append((o, k, ["g++"] + CXX_FLAGS + ["-c", s, "-o", o]))
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for (o, k, _), warn in zip(jobs, ex.map(_run, [j[2] for j in jobs])):
                with open(o + ".key", "w") as f:
                    f.write(k)
                if verbose and warn:
                    print(warn, file=sys.stderr)
    manifest = os.path.join(OBJ, "linked.txt")
    listed = "\n".join(f"{os.path.basename(o)} {k}" for o, k in zip(objs, keys))
    same_list = os.path.exists(manifest) and open(manifest).read() == listed
    if jobs or force or not same_list or not os.path.exists(LIB):
        tmp = f"{LIB}.{os.getpid()}.tmp"
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-lgomp"])
        os.replace(tmp, LIB)
        with open(manifest, "w") as f:
            f.write(listed)
    return LIB


if __name__ == "__main__":
    path = build_extension(force="--force" in sys.argv, verbose=True)
    print(path)

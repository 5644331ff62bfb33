"""Builds csrc/ into libgfnerf_amd.so (in-tree) with hipcc for gfx950.

    python gf-nerf_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the tree.
-ffp-contract=off is required: the sampler and the hash kernels are bit-compared
against the CPU oracle, which is built the same way.
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libgfnerf_amd.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIP_SOURCES = ["runtime.hip", "hash3d.hip", "part_lookup.hip", "lm_lookup.hip", "sampler.hip", "mlp.hip", "field.hip", "field_x3.hip", "fused_fwd.hip", "wgrad_x3.hip", "render.hip", "batch.hip", "grad_part.hip", "octree_proc.hip"]
CXX_SOURCES = ["octree_build.cpp", "octree_proc.cpp"]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall",
               "-Wno-unused-function", "-I", INCLUDE]
CXX_FLAGS = ["-O2", "-fPIC", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-Wall", "-I", INCLUDE]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 extension cannot be built")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return r.stderr


def build_extension(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("gfn_common.h", "x3_device.h", "hash_device.h")]
    headers += [os.path.join(INCLUDE, "gfnerf_amd.h"), os.path.abspath(__file__)]
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", "_hip.o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + HIPCC_FLAGS + ["-c", s, "-o", o])
    for src in CXX_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".cpp", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append(["g++"] + CXX_FLAGS + ["-c", s, "-o", o])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(_run, jobs):
                if verbose and warn:
                    print(warn, file=sys.stderr)
    if jobs or force or _stale(LIB, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lgomp"])
    return LIB


if __name__ == "__main__":
    path = build_extension(force="--force" in sys.argv, verbose=True)
    print(path)

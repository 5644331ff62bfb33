"""Round 5: the level-major lookup the product runs against the UPPER BOUND of "a level pair per XCD"
(tools/xcd_lookup_experiment.hip: same gathers, nothing parked, features wrong -- timing only) on a bench-sized batch
(1,600-camera scene, 4096 rays x 192 = 787 k samples, T = 2^21).  Build the experiment first:

    cd tools && hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -I ../include xcd_lookup_experiment.hip \
        -o libxcd_lookup_experiment.so -L../gf-nerf_amd -l:libgfnerf_amd.so -Wl,-rpath,'$ORIGIN/../gf-nerf_amd'

FLUSH=1 streams 1 GiB between the launches (the table is then in neither L2 nor the memory-side cache, as inside a training step)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import gfnerf_amd  # noqa: F401
from gfnerf_amd import hotpath, ops, scene

sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
cfg = hotpath.HotPathConfig(log2_T=21)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=0)
t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
o, d, cam = scene.make_rays(sc["c2w"], 4096, seed=0)
s = hp.sample(t(o), t(d), t(scene.make_noise(4096, True, 7.59, seed=1)))
n = int(s["pts_offsets"][-1].item())
pts, anc = s["warped_pts"][:n].contiguous(), s["anchors"][:n].contiguous()
V, T = hp.n_volumes, cfg.log2_T
out = torch.empty((16, n, 2), dtype=torch.float32, device="cuda")
out2 = torch.empty_like(out)
lib = C.CDLL(os.path.join(ROOT, "tools", "libxcd_lookup_experiment.so"))
lib.xcd_lookup.restype = C.c_int
lib.xcd_lookup.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda") if os.environ.get("FLUSH") else None


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if flush is not None:
            flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def xcd(nt):
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.xcd_lookup(hp.table16.data_ptr(), hp.prim.data_ptr(), pts.data_ptr(), anc.data_ptr(), anc.stride(0), n, None, V, T, out2.data_ptr(), nt, st)
    assert rc == 0


print("samples", n, "flush", flush is not None)
print("product lm_lookup_kernel (halves from level 6)      : %.4f ms" % timed(lambda: ops.hash_fwd_lm(hp.table16, hp.prim, None, pts, anc, V, T, planar=True, out=out)))
print("product lm_lookup_kernel (every level split, hf = 0): %.4f ms" % timed(lambda: ops.hash_fwd_lm(hp.table16, hp.prim, None, pts, anc, V, T, halves_from=0, planar=True, out=out)))
print("level pair per XCD, upper bound (nothing parked)    : %.4f ms" % timed(lambda: xcd(0)))
print("... points read non-temporally                      : %.4f ms" % timed(lambda: xcd(1)))

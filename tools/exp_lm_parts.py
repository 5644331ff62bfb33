"""Scratch: level-major lookup on a bench-sized batch (1,600-camera scene, 4096 rays x 192): halves / quarters / eighths of a level
per pass (split_log2) x the first split level (HF=4,5,6,...).  The variants this script also timed in round 2 were debug builds of
csrc/lm_lookup.hip (every level reading the same 2 MB: 0.325 ms, the all-warm bound; a prefetch of the next cold part; 512-thread
workgroups; the barrier after the issue) -- numbers in tools/README.md."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd
from gfnerf_amd import hotpath, scene, ops

sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
cfg = hotpath.HotPathConfig(log2_T=21)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
hp.blocks[0].uniform_(-1e-3, 1e-3); hp.refresh_shadows()
t = lambda a: torch.from_numpy(a).cuda()
o, d, cam = scene.make_rays(sc["c2w"], 4096, seed=0)
noise = t(scene.make_noise(4096, True, 7.59, seed=1))
s = hp.sample(t(o), t(d), noise)
n = int(s["pts_offsets"][-1].item())
print("samples", n)
pts = s["warped_pts"][:n].contiguous(); anc = s["anchors"][:n].contiguous()
V, T = hp.n_volumes, cfg.log2_T
out = torch.empty((16, n, 2), dtype=torch.float32, device="cuda")
flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda") if os.environ.get("FLUSH") else None   # 1 GiB: what a step streams between lookups
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if flush is not None:
            flush.add_(1.0)   # L2 and the memory-side cache forget the table
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for sl in (1, 2, 3):
    for hf in [int(x) for x in os.environ.get("HF", "6").split(",")]:
        t1 = timed(lambda: ops.hash_fwd_lm(hp.table16, hp.prim, None, pts, anc, V, T, halves_from=hf, split_log2=sl, planar=True, out=out))
        t2 = timed(lambda: ops.hash_fwd_lm(hp.table16, hp.prim, None, pts, anc, V, T, table2_f16=hp.blocks16[0], prim_pool2=hp.block_prims[0],
                                           halves_from=hf, split_log2=sl, planar=True, out=out))
        print("split_log2=%d halves_from=%d flush=%s: one table %.4f ms, two tables %.4f ms" % (sl, hf, bool(os.environ.get("FLUSH")), t1, t2), flush=True)
ref = ops.hash_fwd_f16(hp.table16, hp.prim, None, pts, anc, V, T)
got = ops.hash_fwd_lm(hp.table16, hp.prim, None, pts, anc, V, T, halves_from=6, planar=True, out=out).permute(1, 0, 2).reshape(n, 32)
print("bit-exact vs direct:", bool(torch.equal(ref, got)))

"""Scratch: level-major lookup (gfn_hash_fwd_lm) vs the direct f16 lookup: bit-exactness and time."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd
from gfnerf_amd import hotpath, scene, ops

sc = scene.build_scene(n_side=20, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
cfg = hotpath.HotPathConfig(log2_T=21)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
hp.blocks[0].uniform_(-1e-3, 1e-3); hp.refresh_shadows()
t = lambda a: torch.from_numpy(a).cuda()
o, d, cam = scene.make_rays(sc["c2w"], 4096, seed=0)
noise = t(scene.make_noise(4096, True, 6.04, seed=1))
s = hp.sample(t(o), t(d), noise)
n = int(s["pts_offsets"][-1].item())
print("samples", n)
pts = s["warped_pts"][:n].contiguous(); anc = s["anchors"][:n].contiguous()
V, T = hp.n_volumes, cfg.log2_T
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), r
td, ref = timed(lambda: ops.hash_fwd_f16(hp.table16, hp.prim, None, pts, anc, V, T))
td2, ref2 = timed(lambda: ops.hash_fwd_f16(hp.table16, hp.prim, None, pts, anc, V, T, table2_f16=hp.blocks16[0], prim_pool2=hp.block_prims[0]))
print("direct: one table %.3f ms, two tables %.3f ms" % (td, td2), flush=True)
for hf in (16, 6, 8, 10, 12):
    for planar in (False, True):
        t1, got = timed(lambda: ops.hash_fwd_lm(hp.table16, hp.prim, None, pts, anc, V, T, halves_from=hf, planar=planar))
        t2, got2 = timed(lambda: ops.hash_fwd_lm(hp.table16, hp.prim, None, pts, anc, V, T, table2_f16=hp.blocks16[0], prim_pool2=hp.block_prims[0], halves_from=hf, planar=planar))
        if planar:
            got, got2 = got.permute(1, 0, 2).reshape(n, 32), got2.permute(1, 0, 2).reshape(n, 32)
        print("level-major halves_from=%2d planar=%d: one table %.3f ms (bit-exact %s), two tables %.3f ms (bit-exact %s)" %
              (hf, planar, t1, bool(torch.equal(ref, got)), t2, bool(torch.equal(ref2, got2))), flush=True)

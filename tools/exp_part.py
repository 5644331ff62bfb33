"""Scratch: radix-partitioned lookup (gfn_hash_fwd_part) vs the direct f16 lookup: bit-exactness and time."""
import os, sys, json
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
# exact-size views so that the partition grid matches the live count
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
tp, (got, ovf) = timed(lambda: ops.hash_fwd_part(hp.table16, hp.prim, None, pts, anc, V, T))
print("direct single table %.3f ms   partitioned %.3f ms   overflow %d" % (td, tp, int(ovf.item())))
print("bit-exact:", bool(torch.equal(ref, got)), "mismatch frac", float((ref != got).float().mean()))
td2, ref2 = timed(lambda: ops.hash_fwd_f16(hp.table16, hp.prim, None, pts, anc, V, T, table2_f16=hp.blocks16[0], prim_pool2=hp.block_prims[0]))
def two():
    out, _ = ops.hash_fwd_part(hp.table16, hp.prim, None, pts, anc, V, T)
    out, ov = ops.hash_fwd_part(hp.blocks16[0], hp.block_prims[0], None, pts, anc, V, T, out=out, accumulate=True)
    return out
tp2, got2 = timed(two)
print("direct dual %.3f ms   partitioned dual %.3f ms   bit-exact %s" % (td2, tp2, bool(torch.equal(ref2, got2))))

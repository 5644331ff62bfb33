"""Scratch experiment: what bounds hash_fwd?  cold (MALL flushed) vs warm (back-to-back) timings."""
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
hp.blocks[0].uniform_(-1e-3, 1e-3)
hp.refresh_shadows()
o, d, cam = scene.make_rays(sc["c2w"], 4096, seed=0)
t = lambda a: torch.from_numpy(a).cuda()
fin = float(os.environ.get("FINENESS", "3.3"))
noise = t(scene.make_noise(4096, True, fin, seed=1))
s = hp.sample(t(o), t(d), noise)
n = s["pts_offsets"][-1].item()
print("samples", n, n / 4096)
flush = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")

def timeit(fn, cold, reps=10):
    ts = []
    for _ in range(reps):
        if cold:
            flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

V, T = hp.n_volumes, cfg.log2_T
dual = lambda: ops.hash_fwd_f16(hp.table16, hp.prim, None, s["warped_pts"], s["anchors"], V, T, table2_f16=hp.blocks16[0], prim_pool2=hp.block_prims[0], n_dev=s["n_dev"])
single = lambda: ops.hash_fwd_f16(hp.table16, hp.prim, None, s["warped_pts"], s["anchors"], V, T, n_dev=s["n_dev"])
res = {}
for name, fn in (("dual", dual), ("single", single)):
    fn(); torch.cuda.synchronize()
    res[name] = dict(cold=timeit(fn, True), warm=timeit(fn, False))
print(json.dumps(res))

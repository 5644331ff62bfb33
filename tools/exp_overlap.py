"""Scratch: does the MFMA-bound decoder overlap with the fabric-bound hash lookup when they run on two streams?"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd
from gfnerf_amd import hotpath, scene, ops, _lib

sc = scene.build_scene(n_side=20, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
cfg = hotpath.HotPathConfig(log2_T=21)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
hp.blocks[0].uniform_(-1e-3, 1e-3); hp.refresh_shadows()
t = lambda a: torch.from_numpy(a).cuda()
def batch(seed, R=2048):
    o, d, cam = scene.make_rays(sc["c2w"], R, seed=seed)
    noise = t(scene.make_noise(R, True, 6.0, seed=1))
    s = hp.sample(t(o), t(d), noise)
    return s, t(cam)
sA, camA = batch(0); sB, camB = batch(1)
print("samples", sA["pts_offsets"][-1].item(), sB["pts_offsets"][-1].item())
V, T = hp.n_volumes, cfg.log2_T
def hashf(s):
    return ops.hash_fwd_f16(hp.table16, hp.prim, None, s["warped_pts"], s["anchors"], V, T, table2_f16=hp.blocks16[0], prim_pool2=hp.block_prims[0], n_dev=s["n_dev"])
featA = hashf(sA)
shA = ops.sh4(sA["dirs_unit"])
img = hp._x3()[0]
CUS = [0]
def field(s, feat, cam, sh):
    return ops.field_fwd(feat, s["ray_id"], sh, hp.emb, cam, hp.layers, n_dev=s["n_dev"], save=True, image=img, opts=ops.launch_opts(decoder_cus=CUS[0]))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
print("hash(B) alone", timed(lambda: hashf(sB)))
for cus in (0, 128, 64, 48, 32):
    CUS[0] = cus
    ta = timed(lambda: field(sA, featA, camA, shA))
    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            field(sA, featA, camA, shA)
        with torch.cuda.stream(s1):
            hashf(sB)
        cur.wait_stream(s1); cur.wait_stream(s2)
    tb = timed(both)
    print(f"decoder on {cus or 256} CUs: field alone {ta:.3f} ms; field(A) || hash(B) on two streams {tb:.3f} ms")

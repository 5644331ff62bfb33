"""Scratch: how full are the record streams of the LDS-accumulated backward (global stage, after some training)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd
from gfnerf_amd import hotpath, scene, ops
sc = scene.build_scene(n_side=20, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
stage = sys.argv[1] if len(sys.argv) > 1 else "global"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = hotpath.HotPathConfig(log2_T=21)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
hp.blocks[0].uniform_(-1e-3, 1e-3); hp.refresh_shadows()
t = lambda a: torch.from_numpy(a).cuda()
S = 128 if stage == "global" else 192
fin = {128: 9.16, 192: 6.04}[S]
B = []
for k in range(4):
    o, d, cam = scene.make_rays(sc["c2w"], 4096, seed=k)
    B.append((t(o), t(d), t(cam), t(np.random.default_rng(k).random((4096, 3)).astype(np.float32)), t(scene.make_noise(4096, True, fin, seed=k + 1))))
s3 = ops.s3im_index(4096, 10, device="cuda")
hp.record_counts_seen = torch.zeros_like(hp._record_buffers()[1])
for i in range(steps):
    o, d, cam, gt, noise = B[i % 4]
    hp.train_step(o, d, cam, gt, noise, stage=stage, block=0 if stage == "focal" else None, s3im_index=s3, update_oct=stage == "global")
    if i in (0, 19, 49, steps - 1):
        torch.cuda.synchronize()
        cnt = hp.record_counts_seen.cpu().numpy()
        cap = hp._records[0].numel() // cnt.size
        print(f"step {i}: records {int(cnt.sum())}, stream capacity {cap}, streams at capacity {int((cnt >= cap).sum())} of {cnt.size}, "
              f"mean {cnt.mean():.1f}, p99 {np.percentile(cnt, 99):.0f}, max {cnt.max()}", flush=True)

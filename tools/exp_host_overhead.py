"""Scratch: host time to enqueue one training step vs its GPU time (is the loop launch-bound?)."""
import os, sys, time
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
gt = t(np.random.default_rng(0).random((4096, 3)).astype(np.float32))
noise = t(scene.make_noise(4096, True, 6.04, seed=1))
s3 = ops.s3im_index(4096, 10, device="cuda")
args = (t(o), t(d), t(cam), gt, noise)
for _ in range(5):
    hp.train_step(*args, stage="focal", block=0, s3im_index=s3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    hp.train_step(*args, stage="focal", block=0, s3im_index=s3)
t_host = (time.perf_counter() - t0) / 50
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 50
print(f"host enqueue {t_host*1e3:.3f} ms/step, wall {t_all*1e3:.3f} ms/step")

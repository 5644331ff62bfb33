"""Scratch: run-length statistics of the bench batch per hash level -- how many of a wave's 64 consecutive samples open a
new (volume, cell) run, i.e. what a run-sharing lookup could save."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd
from gfnerf_amd import hotpath, scene
cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene")
sc = scene.build_scene(n_side=20, max_depth=16, seed=1234, cache_dir=cache)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=21), seed=0, n_blocks=1)
o, d, cam = scene.make_rays(sc["c2w"], 4096, seed=0)
noise = torch.from_numpy(scene.make_noise(4096, True, 6.0448, seed=1)).cuda()
s = hp.sample(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), noise)
n = int(s["pts_offsets"][-1].item())
pts = s["warped_pts"][:n].cpu().numpy().astype(np.float64)
vol = s["anchors"][:n, 0].cpu().numpy()
print("n", n, "pts range", pts.min(0), pts.max(0))
for level in range(16):
    scale = 2.0 ** (7.0 * level / 15.0 + 3.0)
    cell = np.floor(pts * scale).astype(np.int64)
    new = np.ones(n, bool)
    new[1:] = (vol[1:] != vol[:-1]) | (cell[1:] != cell[:-1]).any(1)
    lane0 = (np.arange(n) % 64) == 0
    heads = new | lane0
    per_wave = np.add.reduceat(heads.astype(np.int64), np.arange(0, n, 64))
    print(f"level {level:2d} res {scale:7.1f}: heads/sample {heads.mean():.3f}, waves with <=40 heads {np.mean(per_wave <= 40):.3f}, gathers kept with sharing {np.where(np.repeat(per_wave <= 40, 64)[:n], heads, True).mean():.3f}")

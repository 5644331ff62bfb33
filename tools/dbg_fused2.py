import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gfnerf_amd
from gfnerf_amd import hotpath, scene, ops
from helpers import small_scene, rays_for
sc = small_scene()
cfg = hotpath.HotPathConfig(log2_T=14, early_T=0.0, grad_f16=False, keep_features=True)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
rng = np.random.default_rng(11)
hp.blocks[0].copy_(torch.from_numpy(rng.uniform(-1e-2, 1e-2, tuple(hp.blocks[0].shape)).astype(np.float32)))
hp.table.mul_(100.0); hp.layers[1][0].mul_(4.0); hp.refresh_shadows()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
R = 384
o, d, cam = rays_for(sc, R, seed=3)
noise = t(scene.make_noise(R, True, 5.0, seed=2))
s = hp.sample(t(o), t(d), noise)
N = int(s["pts_offsets"][-1])
for block in (None, 0):
    f, comp = hp._forward_fused(s, t(cam), block, 0.0, save=True)
    sep = hp._field_forward(s, t(cam), block=block, fused=True, save=True)
    a, b = f["feat"].cpu().numpy()[:N], sep["feat"].cpu().numpy()[:N]
    bad = a != b
    print(block, "mismatch frac", bad.mean(), "nan", np.isnan(a).sum(), np.isnan(b).sum(), "rows", np.unique(np.nonzero(bad)[0])[:10], "cols", np.unique(np.nonzero(bad)[1])[:32])
    if bad.any():
        i, j = np.argwhere(bad)[0]; print(a[i, j], b[i, j], i, j, "ray", int(s["ray_id"][i]))

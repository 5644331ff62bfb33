import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
log = open(os.path.join(ROOT, "gpurun_out", "dbg_fused.log"), "w")
def P(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)
import numpy as np, torch
import gfnerf_amd
from gfnerf_amd import hotpath, scene, ops
from helpers import small_scene, rays_for
sc = small_scene()
cfg = hotpath.HotPathConfig(log2_T=14, early_T=0.0, grad_f16=False, keep_features=True)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
hp.table.mul_(100.0); hp.refresh_shadows()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for R in (4, 64, 384):
    o, d, cam = rays_for(sc, R, seed=3)
    noise = t(scene.make_noise(R, True, 5.0, seed=2))
    s = hp.sample(t(o), t(d), noise)
    torch.cuda.synchronize(); P("sampled", R, int(s["pts_offsets"][-1]))
    for block in (None, 0):
        for eT in (0.0, 1e-4):
            t0 = time.time()
            f, comp = hp._forward_fused(s, t(cam), block, eT, save=True)
            torch.cuda.synchronize()
            P("fused ok", R, block, eT, round(time.time() - t0, 3), float(comp["rgb"].sum()), float((f["sigma"][:int(s["pts_offsets"][-1])] == 0).float().mean()))
P("done")

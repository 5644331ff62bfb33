"""Scratch: gradient pool of the LDS-accumulated backward against the atomic one (and the atomic one against itself)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gfnerf_amd
from gfnerf_amd import hotpath, scene, ops
from helpers import small_scene, rays_for
sc = small_scene()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
o, d, cam = rays_for(sc, 512, seed=3)
o, d, cam = t(o), t(d), t(cam)
gt = t(np.random.default_rng(0).random((512, 3)).astype(np.float32))
noise = t(scene.make_noise(512, True, 6.0, seed=1))
s3 = ops.s3im_index(512, 10, device="cuda")
def run(lds, cap=1 << 22):
    cfg = hotpath.HotPathConfig(log2_T=14, lds_accumulate=lds, record_capacity=cap)
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=1, n_blocks=1)
    hp.blocks[0].uniform_(-1e-3, 1e-3); hp.refresh_shadows()
    if lds:
        hp.record_counts_seen = torch.zeros_like(hp._record_buffers()[1])
    hp.forward_backward(o, d, cam, gt, noise, stage="focal", block=0, s3im_index=s3)
    torch.cuda.synchronize()
    extra = hp.record_counts_seen.cpu().numpy() if lds else None
    return hp.table_grad.float().cpu().numpy().copy(), extra
a1, _ = run(False); a2, _ = run(False); l1, cnt = run(True); l2, cnt2 = run(True, 512 * 40)
scale = np.abs(a1).max()
def cmp(x, y, name):
    dlt = np.abs(x - y)
    print(f"{name}: max|diff|/scale {dlt.max() / scale:.3e}, sum x {x.sum():.6e} sum y {y.sum():.6e}, nnz {int((x != 0).sum())} {int((y != 0).sum())}, "
          f"violations(4e-3 scale + 3e-2 rel) {int((dlt > scale * 4e-3 + 3e-2 * np.abs(y)).sum())}")
    i = np.unravel_index(np.argmax(dlt), dlt.shape)
    print("   worst at", i, x[i], y[i])
print("scale", scale, "records", int(cnt.sum()), "max per region", int(cnt.max()), "| small capacity:", int(cnt2.sum()))
cmp(a2, a1, "atomic vs atomic")
cmp(l1, a1, "lds vs atomic")
cmp(l2, a1, "lds (overflowing) vs atomic")

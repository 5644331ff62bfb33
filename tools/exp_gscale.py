#!/usr/bin/env python3
"""Where do the per-sample gradient magnitudes sit relative to the batch maximum the global-stage x3 backward scales by?"""
import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa
from gfnerf_amd import hotpath, ops, procedural, scene

sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
dev = "cuda"
cfg = hotpath.HotPathConfig(log2_T=21)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=1600, cfg=cfg, seed=0, n_blocks=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
pool = []
for k in range(16):
    o, d, cam = scene.make_rays(sc["c2w"], 4096, seed=k)
    o, d, cam = t(o), t(d), t(cam)
    pool.append((o, d, cam, procedural.gt_colors(o, d)))
fin = 11.4
gen = torch.Generator(device=dev).manual_seed(11)
for i in range(1501):
    o, d, cam, gt = pool[i % 16]
    noise = ops.sampler_noise(4096, fin, train=True, generator=gen, device=dev)
    s3 = ops.s3im_index(4096, 10, generator=gen, device=dev)
    if i in (60, 300, 1500):
        comp, ctx = hp.forward_train(o, d, cam, noise, stage="global")
        s, f = ctx["samples"], ctx["field"]
        n = int(s["pts_offsets"][-1].item())
        loss, g = ops.charbonnier(comp["rgb"], gt)
        ops.s3im(gt, comp["rgb"], s3, patch_h=32, mult=1.0, loss=loss, g_pred=g)
        d_sigma, d_rgb = ops.composite_bwd(s["pts_offsets"], f["sigma"], s["dists"], f["rgb"], g, None, early_T=cfg.early_T)
        a = (d_sigma[:n] * torch.exp(torch.clamp(f["h0"][:n] + 1, -15, 15))).abs()
        b = d_rgb[:n].abs().max(1).values
        m = torch.maximum(a, b)
        gmax = float(m.max())
        live = m > 0
        r = (m[live] / gmax).double()
        q = torch.quantile(r, torch.tensor([0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99], dtype=torch.float64, device=dev)).tolist()
        # share of the total |gradient| mass carried by samples below 2^-k of the maximum
        tot = float(m.sum())
        shares = {k: float(m[m < gmax * 2.0 ** -k].sum()) / tot for k in (8, 12, 16, 20, 24)}
        print(f"step {i}: loss {float(loss):.4f} gmax {gmax:.3e} scale 2^{math.log2(ops.grad_scale(gmax)):.0f}; live {float(live.float().mean()):.3f}; m/gmax quantiles", ["%.1e" % v for v in q],
              "mass below 2^-k of max:", {k: round(v, 4) for k, v in shares.items()}, "which term is the max: dsigma" if float(a.max()) >= float(b.max()) else "drgb", flush=True)
    hp.train_step(o, d, cam, gt, noise, stage="global", s3im_index=s3, update_oct=True)

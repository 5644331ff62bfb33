#!/usr/bin/env python3
"""One global-stage training step from the same state through the x3 engine path and through the exact-f32 per-layer path: where do the updates differ?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa
from gfnerf_amd import hotpath, ops, procedural, scene

sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
dev = "cuda"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
pool = []
for k in range(8):
    o, d, cam = scene.make_rays(sc["c2w"], 4096, seed=k)
    o, d, cam = t(o), t(d), t(cam)
    pool.append((o, d, cam, procedural.gt_colors(o, d)))
fin = 11.4
WARM = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def engine(prec):
    cfg = hotpath.HotPathConfig(log2_T=21, mlp_precision=prec)
    return hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=1600, cfg=cfg, seed=0, n_blocks=0)


a = engine("x3")
gen = torch.Generator(device=dev).manual_seed(11)
for i in range(WARM):  # a common, partly trained state
    o, d, cam, gt = pool[i % 8]
    a.train_step(o, d, cam, gt, ops.sampler_noise(4096, fin, train=True, generator=gen, device=dev), stage="global",
                 s3im_index=ops.s3im_index(4096, 10, generator=gen, device=dev), update_oct=False)
b = engine("f32")
for name in ("table", "mlp_flat", "emb"):
    getattr(b, name).copy_(getattr(a, name))
b.refresh_shadows(); a.refresh_shadows()
for k in ("table", "mlp", "emb"):
    b._adam[k] = tuple(x.clone() for x in a._adam[k]); b._adam_t[k] = a._adam_t[k]
b.step = a.step
o, d, cam, gt = pool[3]
noise = ops.sampler_noise(4096, fin, train=True, generator=gen, device=dev)
s3 = ops.s3im_index(4096, 10, generator=gen, device=dev)
before = {n: getattr(a, n).clone() for n in ("table", "mlp_flat", "emb")}
res = {}
for tag, hp in (("x3", a), ("f32", b)):
    loss, ctx = hp.forward_backward(o, d, cam, gt, noise, stage="global", s3im_index=s3)
    n = int(ctx["samples"]["pts_offsets"][-1].item())
    tg = hp.table_grad.float().clone() / (128.0 if hp.table_grad.dtype == torch.float16 else 1.0)
    res[tag] = dict(loss=float(loss), rgb=ctx["comp"]["rgb"].clone(), tg=tg, mg=hp.mlp_grad.clone(), eg=hp.emb_grad.clone(), path=hp.last_forward, n=n)
    hp.optimizer_step("global")
    res[tag]["d"] = {nm: (getattr(hp, nm) - before[nm]) for nm in before}
x, f = res["x3"], res["f32"]
print("paths", x["path"], f["path"], "loss", x["loss"], f["loss"], "max |rgb diff|", float((x["rgb"] - f["rgb"]).abs().max()))
for k in ("tg", "mg", "eg"):
    dd = (x[k] - f[k]).abs()
    print(k, "max |g|", float(f[k].abs().max()), "max diff", float(dd.max()), "rel L1", float(dd.sum() / f[k].abs().sum()), "cos", float((x[k] * f[k]).sum() / (x[k].norm() * f[k].norm())))
for nm in before:
    dx, df = x["d"][nm], f["d"][nm]
    print("update", nm, "L1 x3", float(dx.abs().sum()), "L1 f32", float(df.abs().sum()), "L1 diff", float((dx - df).abs().sum()), "cos", float((dx * df).sum() / (dx.norm() * df.norm())))
# layer by layer
off = 0
for li, (W, B) in enumerate(a.layers):
    for nm, p in (("W", W), ("b", B)):
        sl = slice(off, off + p.numel()); off += p.numel()
        gx, gf = x["mg"][sl], f["mg"][sl]
        print(f"layer {li} {nm}: |g| max {float(gf.abs().max()):.3e} max diff {float((gx - gf).abs().max()):.3e} rel L1 {float((gx - gf).abs().sum() / gf.abs().sum()):.3e}")

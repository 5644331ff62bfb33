#!/usr/bin/env python3
"""x3 engine path and exact-f32 per-layer path trained in lockstep from one init on the same batches: where do the trajectories part?"""
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
NPOOL = int(sys.argv[3]) if len(sys.argv) > 3 else 16
for k in range(NPOOL):
    o, d, cam = scene.make_rays(sc["c2w"], 4096, seed=k)
    o, d, cam = t(o), t(d), t(cam)
    pool.append((o, d, cam, procedural.gt_colors(o, d)))
fin = 11.4
kw = dict(a=dict(mlp_precision="x3"), b=dict(mlp_precision="f32"))
if len(sys.argv) > 1:
    kw["a"].update(eval(sys.argv[1]))
E = {k: hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=1600, cfg=hotpath.HotPathConfig(log2_T=21, **v), seed=0, n_blocks=0) for k, v in kw.items()}
gen = torch.Generator(device=dev).manual_seed(11)
eo, ed, ecam = scene.make_rays(sc["c2w"], 16384, seed=7_000_000)
eo, ed, ecam = t(eo), t(ed), t(ecam)
egt = procedural.gt_colors(eo, ed)


def evaluate(hp):
    out = [hp.render(eo[a:a + 4096].contiguous(), ed[a:a + 4096].contiguous(), ecam[a:a + 4096].contiguous(), ops.sampler_noise(4096, fin, train=False, device=dev))["rgb"] for a in range(0, 16384, 4096)]
    return procedural.psnr(torch.cat(out), egt)


N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
run = {k: 0.0 for k in E}
for i in range(N):
    o, d, cam, gt = pool[i % NPOOL]
    noise = ops.sampler_noise(4096, fin, train=True, generator=gen, device=dev)
    s3 = ops.s3im_index(4096, 10, generator=gen, device=dev)
    L = {}
    for k, hp in E.items():
        hp.cfg.lr = 1e-2 * float(np.exp(np.log(1e-2) * i / 2000))
        loss, ctx = hp.train_step(o, d, cam, gt, noise, stage="global", s3im_index=s3, update_oct=False)
        L[k] = float(loss)
        run[k] = 0.98 * run[k] + 0.02 * L[k]
    if (i + 1) % 200 == 0:
        print(f"step {i + 1}: psnr x3 {evaluate(E['a']):.3f} f32 {evaluate(E['b']):.3f}  running loss x3 {run['a']:.5f} f32 {run['b']:.5f}", flush=True)
    if False:
        a, b = E["a"], E["b"]
        dt = (a.table - b.table).abs()
        print(f"step {i:4d} {a.last_forward:6s} loss x3 {L['a']:.5f} f32 {L['b']:.5f} diff {L['a'] - L['b']:+.2e} | table: mean |d| {float(dt.mean()):.2e} max {float(dt.max()):.2e} | mlp max |d| {float((a.mlp_flat - b.mlp_flat).abs().max()):.2e} emb {float((a.emb - b.emb).abs().max()):.2e}", flush=True)

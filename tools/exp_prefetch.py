"""Scratch: sampler of the next batch on a side stream beside the fused forward / backward capped to fewer CUs."""
import os, sys, time
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
B = []
for k in range(4):
    o, d, cam = scene.make_rays(sc["c2w"], 4096, seed=k)
    B.append((t(o), t(d), t(cam), t(np.random.default_rng(k).random((4096, 3)).astype(np.float32)), t(scene.make_noise(4096, True, 6.04, seed=k + 1))))
s3 = ops.s3im_index(4096, 10, device="cuda")
side = torch.cuda.Stream()
def run(cus, prefetch, steps=30):
    _lib.lib().gfn_set_decoder_cus(cus)
    pre = None
    def step(i):
        nonlocal pre
        o, d, cam, gt, noise = B[i % 4]
        main = torch.cuda.current_stream()
        hp.zero_grads("focal", 0)
        if pre is not None:
            main.wait_stream(side); s = pre; pre = None
            for v in list(s.values()) + list(s["workspace"].values()):
                if isinstance(v, torch.Tensor): v.record_stream(main)
        else:
            s = hp.sample(o, d, noise)
        f, comp = hp._forward_fused(s, cam, 0, cfg.early_T, save=True)
        if prefetch:  # next batch's sampler beside this step's backward + Adam
            no, nd, _, _, nn = B[(i + 1) % 4]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pre = hp.sample(no, nd, nn)
        loss, g = ops.charbonnier(comp["rgb"], gt)
        ops.s3im(gt, comp["rgb"], s3, loss=loss, g_pred=g)
        ctx = dict(samples=s, field=f, comp=comp, cam_idx=cam, stage="focal", fused=True, block=0)
        hp.backward(ctx, g)
        hp.optimizer_step("focal", 0)
    for i in range(4): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(4 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for cus in (0, 240, 224, 192):
    print(f"fused kernels on {cus or 256} CUs: no prefetch {run(cus, False):.3f} ms/step, sampler prefetched beside the backward {run(cus, True):.3f} ms/step")

#!/usr/bin/env python3
"""Round-3 probes on the bench workload (1,600-camera scene, T = 2^21, 4096 x 192):
  1. lm_lookup_kernel time against the number of samples it is given (fixed cost per launch vs cost per sample),
  2. where rays terminate (index of the last alive sample / samples of the ray): what a two-phase forward could skip,
  3. the step through the plugin surface (pipeline.get_train_loss_dict -> backward -> Optimizers) beside the native step.
Usage: python tools/exp_r03_probe.py [lm] [term] [plugin]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import hotpath, ops, scene  # noqa: E402

what = set(sys.argv[1:]) or {"lm", "term", "plugin"}
N_RAYS = 4096
out = {}


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
if what & {"lm", "term"}:
    cfg = hotpath.HotPathConfig(log2_T=21)
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
    g = torch.Generator(device="cuda").manual_seed(100)
    hp.blocks[0].uniform_(-1e-3, 1e-3, generator=g)
    hp.refresh_shadows()
    o, d, cam = scene.make_rays(sc["c2w"], N_RAYS, seed=0)
    o, d, cam = (torch.from_numpy(a).cuda() for a in (o, d, cam))
    lo, hi = 0.5, 64.0
    for _ in range(14):
        fin = (lo * hi) ** 0.5
        s = hp.sample(o, d, torch.from_numpy(scene.make_noise(N_RAYS, True, fin, seed=1)).cuda())
        mean = s["pts_offsets"][-1].item() / N_RAYS
        if abs(mean - 192) / 192 < 0.005:
            break
        lo, hi = (fin, hi) if mean > 192 else (lo, fin)
    n_live = int(s["pts_offsets"][-1].item())
    out["fineness"], out["n_live"] = fin, n_live

if "lm" in what:
    T, V = cfg.log2_T, hp.n_volumes
    rows = []
    for frac in (1.0, 0.8, 0.62, 0.5, 0.35, 0.25, 0.1, 0.03):
        n = int(n_live * frac)
        nd = torch.tensor([n], dtype=torch.int32, device="cuda")
        feat = torch.empty((16, s["ts"].shape[0], 2), dtype=torch.float32, device="cuda")
        ms = timed(lambda: ops.hash_fwd_lm(hp.table16, hp.prim, None, s["warped_pts"], s["anchors"], V, T, n_dev=nd, planar=True, out=feat))
        # the direct (sample-major) two-table lookup on the same prefix
        ms_d = timed(lambda: ops.hash_fwd_f16(hp.table16, hp.prim, None, s["warped_pts"], s["anchors"], V, T, table2_f16=hp.blocks16[0],
                                              prim_pool2=hp.block_prims[0], n_dev=nd), iters=5, warm=1)
        rows.append(dict(frac=frac, n=n, lm_ms_per_table=round(ms, 4), direct_two_tables_ms=round(ms_d, 4)))
        print(rows[-1], flush=True)
    out["lm_scaling"] = rows

if "term" in what:
    gt = torch.rand((N_RAYS, 3), device="cuda")
    noise = torch.from_numpy(scene.make_noise(N_RAYS, True, fin, seed=1)).cuda()
    comp, ctx = hp.forward_train(o, d, cam, noise, stage="focal", block=0)
    off = s["pts_offsets"].long().cpu().numpy()
    w = comp["weights"][:n_live].cpu().numpy()
    cnt = np.diff(off)
    last = np.zeros(N_RAYS)
    for r in range(N_RAYS):
        ww = w[off[r]:off[r + 1]]
        nz = np.nonzero(ww > 0)[0]
        last[r] = (nz[-1] + 1) if nz.size else 0
    fr = last / np.maximum(cnt, 1)
    out["term"] = dict(alive_frac=float((w > 0).mean()), quantiles_of_termination_fraction={q: float(np.quantile(fr, q)) for q in (0.1, 0.25, 0.5, 0.75, 0.9, 0.95, 0.98, 1.0)},
                       quantiles_of_termination_index={q: float(np.quantile(last, q)) for q in (0.1, 0.5, 0.9, 0.98, 1.0)},
                       samples_per_ray={q: float(np.quantile(cnt, q)) for q in (0.1, 0.5, 0.9, 1.0)})
    # two-phase budgets: phase A = first ceil(q * n_r) samples of every ray
    for q in (0.5, 0.6, 0.65, 0.7, 0.8):
        a = np.minimum(np.ceil(q * cnt), cnt)
        b = np.where(last > a, cnt - a, 0)
        out["term"][f"two_phase_q{q}"] = dict(A=float(a.sum() / cnt.sum()), B_all_rest=float(b.sum() / cnt.sum()),
                                               B_alive_only=float(np.where(last > a, last - a, 0).sum() / cnt.sum()), rays_in_B=float((last > a).mean()))
    # fixed K
    for K in (64, 96, 128, 160):
        a = np.minimum(K, cnt)
        b = np.where(last > a, cnt - a, 0)
        out["term"][f"two_phase_K{K}"] = dict(A=float(a.sum() / cnt.sum()), B_all_rest=float(b.sum() / cnt.sum()), rays_in_B=float((last > a).mean()))
    print(json.dumps(out["term"], indent=1), flush=True)

if "plugin" in what:
    if what & {"lm", "term"}:
        del hp, s
        torch.cuda.empty_cache()
    import bench

    res = {}
    variants = [("fused+prefetch", dict()), ("fused+prefetch, eager forward", dict(graph_forward=False)), ("fused", dict(prefetch=False)),
                ("torch_adam", dict(fused_optimizer=False, prefetch=False))]
    if "quick" in what:
        variants = variants[:1]
    if "ab" in what:
        variants = variants[:2] * 2
    import gc
    if "profile" in what:
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        r = bench.plugin_step_bench(sc, torch, "focal", steps=100, warmup=20)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(70)
        pstats.Stats(pr).sort_stats("tottime").print_stats(40)
        print(r)
        sys.exit(0)
    for stage in ("focal", "global"):
        for name, kw in variants:
            gc.collect(); torch.cuda.empty_cache()
            print("allocated GB", torch.cuda.memory_allocated() / 2**30, flush=True)
            r = bench.plugin_step_bench(sc, torch, stage, steps=100, warmup=20, **kw)
            res[f"{stage}:{name}"] = r
            print(stage, name, {k: r[k] for k in ("ms_per_step", "host_ms_per_step", "value", "mean_samples_per_ray", "loss")}, flush=True)
            torch.cuda.empty_cache()
    out["plugin"] = res

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r03_probe.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))

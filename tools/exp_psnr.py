#!/usr/bin/env python3
"""PSNR of the engine trained on the procedural scene (gfnerf_amd.procedural): python tools/exp_psnr.py [steps_global steps_focal [precisions]]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import procedural, scene  # noqa: E402

sg = int(sys.argv[1]) if len(sys.argv) > 1 else 600
sf = int(sys.argv[2]) if len(sys.argv) > 2 else 300
precs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["x3"]
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
VARIANTS = {"x3": {}, "f32": {}, "x3_onephase": dict(two_phase=False), "x3_fused": dict(level_major=False), "x3_unfusedfwd": dict(level_major=False, fused_forward=False),
            "x3_early0": dict(early_T=0.0), "x3_atomic": dict(lds_accumulate=False), "x3_unfusedbwd": dict(fused_backward=False), "f32_early0": dict(early_T=0.0),
            "x3_f32pool": dict(grad_f16=False), "f32_f32pool": dict(grad_f16=False), "x3_noprune": dict(update_oct=False), "f32_noprune": dict(update_oct=False),
            "x3_noprune_onephase": dict(update_oct=False, two_phase=False)}
for p in precs:
    t0 = time.perf_counter()
    kw = dict(VARIANTS[p])
    r = procedural.train_and_eval(sc, precision=p.split("_")[0], steps_global=sg, steps_focal=sf, progress=None, **kw)
    r["variant"] = p
    torch.cuda.synchronize()
    r["seconds"] = time.perf_counter() - t0
    print(json.dumps(r), flush=True)

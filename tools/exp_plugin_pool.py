#!/usr/bin/env python3
"""Plugin-surface focal iteration with the captured forwards in one shared memory pool against a private pool each (round 4): python tools/exp_plugin_pool.py"""
import os, sys, gc
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import hotpath, scene  # noqa: E402
import bench  # noqa: E402

sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
for shared in (True, False, True, False):
    hotpath.HotPathConfig.graph_forward_shared_pool = shared
    for kw in (dict(), dict(graph_forward=False)):
        gc.collect(); torch.cuda.empty_cache()
        r = bench.plugin_step_bench(sc, torch, "focal", steps=100, warmup=30, **kw)
        print("shared_pool", shared, kw, {k: round(r[k], 4) for k in ("ms_per_step", "host_ms_per_step")}, flush=True)

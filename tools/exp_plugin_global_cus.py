"""The plugin-surface GLOBAL-stage iteration (random-init scene) against the CUs its backward keeps beside the prefetched sampler + octree update
(HotPathConfig.prefetch_cus_global; eager launches): python tools/exp_plugin_global_cus.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import scene  # noqa: E402

sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
for n in (192, 160, 224, 256, 192):
    r = bench.plugin_step_bench(sc, torch, "global", steps=150, warmup=60, hot_cfg=dict(prefetch_cus_global=n))
    print("prefetch_cus_global", n, round(r["ms_per_step"], 4), "host", round(r["host_ms_per_step"], 4), flush=True)
    torch.cuda.empty_cache()

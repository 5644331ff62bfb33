#!/usr/bin/env python3
"""The plugin-surface training iteration (bench.plugin_step_bench) as a short run for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import scene  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "focal"
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
trained = (3000, 1000) if len(sys.argv) > 3 and sys.argv[3] == "trained" else None  # the focal step on a scene trained through the plugin surface itself
r = bench.plugin_step_bench(sc, torch, stage, steps=int(sys.argv[2]) if len(sys.argv) > 2 else 100, warmup=20, trained=trained)
print(r)

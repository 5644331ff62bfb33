#!/usr/bin/env python3
"""K of the phased forward and the evaluated / alive sample fractions along the bench's global-stage run (eager steps)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import dist as gdist, scene  # noqa: E402

args = bench.parse()
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
run = bench.StageRun(args, "global", sc, 1, 0, gdist.TrainSync(), torch)
for i in range(300):
    loss, aux = run.step()
    if i % 10 == 0 or i < 5:
        hp = run.hp
        n = int(aux["samples"]["pts_offsets"][-1].item())
        w = aux["comp"]["weights"][:n]
        st = hp._phase.stats_last.tolist() if hp._phase is not None else None
        print(i, hp.last_forward, "alive", round(float((w > 0).float().mean()), 3), "stats_last", st, "next K", int(hp._phase.k.item()) if hp._phase is not None else None, "loss", round(float(loss), 4), flush=True)

#!/usr/bin/env python3
"""PSNR of the split-f16 decoder against the exact-f32 decoder over MANY seeds (gfnerf_amd.procedural.psnr_study; the bench's `psnr`
sub-record runs two): python tools/exp_psnr_seeds.py [n_seeds [steps_global steps_focal]]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import procedural, scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sg = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
sf = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
t0 = time.perf_counter()
r = procedural.psnr_study(sc, seeds=tuple(range(n)), steps_global=sg, steps_focal=sf, progress=lambda m: print(m, flush=True))
r["seconds"] = time.perf_counter() - t0
print(json.dumps(r))

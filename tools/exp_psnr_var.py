#!/usr/bin/env python3
"""Where does the run-to-run PSNR scatter of gfnerf_amd.procedural.train_and_eval come from?  Each argument is a python dict literal of
train_and_eval keyword overrides plus `leg` (a key of procedural.LEGS) and `seeds` (list): python tools/exp_psnr_var.py "dict(leg='fast', seeds=[0,1])" ..."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import procedural, scene  # noqa: E402

sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
for spec in sys.argv[1:]:
    kw = eval(spec)
    leg, seeds = kw.pop("leg", "fast"), kw.pop("seeds", [0])
    for sd in seeds:
        t0 = time.perf_counter()
        r = procedural.train_and_eval(sc, seed=sd, **{**kw, **procedural.LEGS[leg]})
        torch.cuda.synchronize()
        r.update(leg=leg, spec=spec, seconds=round(time.perf_counter() - t0, 1))
        print(json.dumps(r), flush=True)
        torch.cuda.empty_cache()

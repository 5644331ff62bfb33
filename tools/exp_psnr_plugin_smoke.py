"""Round 6 smoke of the PSNR study's plugin leg (procedural.LEGS["plugin"]) at reduced length: does the surface train the procedural scene like the other legs?
    python tools/exp_psnr_plugin_smoke.py [steps_global steps_focal [n_seeds]]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import procedural, scene  # noqa: E402

sg = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sf = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
t0 = time.perf_counter()
st = procedural.psnr_study(sc, seeds=tuple(range(n)), legs=("plugin", "fast", "reference"), steps_global=sg, steps_focal=sf, detail=procedural.STUDY_DETAIL,
                           lr_final=procedural.STUDY_LR_FINAL, holdout_every=8, eval_rays=8192, progress=lambda m: print(m, flush=True))
print({k: v for k, v in st.items() if k.startswith("psnr") or k in ("delta", "se", "plugin_surface")}, f"{time.perf_counter() - t0:.0f} s", flush=True)

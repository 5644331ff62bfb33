"""Round 5: gradient cosines along one short training run (procedural.train_and_eval, the GPU test's settings) with the weight gradient's
operands as one f16 plane each ("rne", the default) and as hi + lo' planes ("x3"): every pair of gradient semantics, every group."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa: F401
from gfnerf_amd import procedural, scene

sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
for mode in sys.argv[1:] or ["rne", "x3"]:
    kw = dict(procedural.LEGS["fast"])
    kw["wgrad_planes"] = mode
    r = procedural.train_and_eval(sc, seed=0, steps_global=400, steps_focal=200, eval_rays=4096, detail=procedural.STUDY_DETAIL, lr_final=procedural.STUDY_LR_FINAL,
                                  cosine_every=100, cosine_legs=("exact", "reference", "fast", "f32", "fast0"), **kw)
    print(mode, "psnr", r["psnr"])
    for stage, step, pairs in r["grad_cosine"]:
        for p in ("exact|reference", "exact|fast0", "exact|fast", "fast|f32"):
            print(mode, stage, step, p, {g: (round(1 - v["cos"], 9), round(v["rel_diff"], 6)) for g, v in pairs[p].items()}, flush=True)

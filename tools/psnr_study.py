#!/usr/bin/env python3
"""The PSNR record of a round (profiles/rNN_psnr_study.json): gfnerf_amd.procedural.psnr_study -- the product's training leg ("fast": x3 decoder,
early termination, phased forward, LDS gradient accumulation, GfnAdam's kernel) against the reference-semantics leg ("reference": every sample
evaluated, exact-f32 decoder, f16 atomics, torch.optim.Adam), paired by seed -- over N seeds, plus the gradient-cosine trajectory of one run and the
octree study (early_T 1e-4 vs 0: leaf sets after the global stage).
    python tools/psnr_study.py OUT.json [n_seeds [steps_global steps_focal [detail]]]
GFN_STUDY_SEED0=k: seeds k .. k + n_seeds - 1 and only the paired study (a call on the GPU box lasts 20 minutes at most: the record of a round is
several calls' outputs merged by tools/psnr_merge.py).  GFN_STUDY_LEGS=a,b: the two legs of procedural.LEGS to pair (default, round 6: plugin,reference --
the engine AS THE PLUGIN SURFACE BUILDS AND DRIVES IT against the reference's training semantics; rounds 4-5: fast,reference).  GFN_STUDY_NO_AUX=1: no
gradient-cosine / octree sections even from seed 0."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import procedural, scene  # noqa: E402

out_path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sg = int(sys.argv[3]) if len(sys.argv) > 3 else procedural.STUDY_STEPS[0]
sf = int(sys.argv[4]) if len(sys.argv) > 4 else procedural.STUDY_STEPS[1]
detail = float(sys.argv[5]) if len(sys.argv) > 5 else procedural.STUDY_DETAIL
LRF = procedural.STUDY_LR_FINAL
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
say = lambda m: print(f"[{time.strftime('%H:%M:%S')}] {m}", flush=True)  # noqa: E731
seed0 = int(os.environ.get("GFN_STUDY_SEED0", "0"))
res = dict(argv=sys.argv[1:], protocol=procedural.PROTOCOL, seed0=seed0)


def save():
    with open(out_path + ".tmp", "w") as f:
        json.dump(res, f, indent=1)
    os.replace(out_path + ".tmp", out_path)


def aux_sections():
    # ---- 1. one trajectory: gradient cosine every 100 steps
    t0 = time.perf_counter()
    r = procedural.train_and_eval(sc, seed=0, steps_global=sg, steps_focal=sf, detail=detail, lr_final=LRF, cosine_every=100,
                                  cosine_legs=("exact", "reference", "fast", "f32", "fast0"), **procedural.LEGS["fast"])
    cos = r.pop("grad_cosine")
    pairs = sorted(cos[0][2])
    groups = ("table", "mlp", "emb")
    res["grad_cosine"] = dict(every=100, seed=0, legs={k: v for k, v in procedural.GRADIENT_LEGS.items()},
                              min_cos={p: {g: min(c[p][g]["cos"] for _, _, c in cos if g in c[p]) for g in groups} for p in pairs},
                              max_rel_diff={p: {g: max(c[p][g]["rel_diff"] for _, _, c in cos if g in c[p]) for g in groups} for p in pairs},
                              mean_rel_diff={p: {g: float(np.mean([c[p][g]["rel_diff"] for _, _, c in cos if g in c[p]])) for g in groups} for p in pairs},
                              trajectory=[(st, i, {p: {g: [round(v["cos"], 9), round(v["rel_diff"], 6)] for g, v in c[p].items()} for p in ("exact|reference", "exact|fast", "exact|fast0", "fast|f32")})
                                          for st, i, c in cos],
                              note="cos / relative L2 difference of the gradient of one batch from one state under pairs of gradient semantics: exact = every sample, exact-f32 "
                                   "decoder, f32 gradient pool; reference = the same with the reference's f16 x128 atomic adds; fast = the product (x3 decoder, cut at T < 1e-4, "
                                   "phased forward, exact LDS sums rounded to f16 once); f32 = the product with the exact-f32 decoder; fast0 = the product without the cut (early_T = 0)")
    say(f"gradient cosine: min {res['grad_cosine']['min_cos']} mean rel diff {res['grad_cosine']['mean_rel_diff']} ({time.perf_counter() - t0:.0f} s)")
    save()

    # ---- 2. octree study: the global stage from one seed with early_T = 1e-4 (exact alphas for K5) | early_T = 0 | early_T = 1e-4 the rounds 2-3 way
    def valid_leaf_mask(hp):
        from gfnerf_amd.gfnerf.perssampler import _NODE_DT
        a = hp.nodes.cpu().numpy().view(_NODE_DT)
        return (a["is_leaf_node"] != 0) & (a["trans_idx"] >= 0), (a["is_leaf_node"] != 0)


    oct_rows = []
    for sd in range(min(n, 4)):
        row = dict(seed=sd)
        masks = {}
        for name, kw in (("early_T_1e-4", dict()), ("early_T_0", dict(early_T=0.0)), ("early_T_1e-4_r03_alphas", dict(exact_oct_alphas=False))):
            rr = procedural.train_and_eval(sc, seed=sd, steps_global=sg, steps_focal=0, detail=detail, lr_final=LRF, holdout_every=8, keep_engine=True, **{**procedural.LEGS["fast"], **kw})
            hp = rr.pop("engine")
            m, is_leaf = valid_leaf_mask(hp)
            masks[name] = m
            row[name] = dict(valid_leaves=int(m.sum()), leaves=int(is_leaf.sum()), psnr=round(rr["psnr"], 3), psnr_heldout_cams=round(rr["psnr_heldout"], 3))
            del hp, rr
            torch.cuda.empty_cache()
        ref = masks["early_T_0"]
        for name in ("early_T_1e-4", "early_T_1e-4_r03_alphas"):
            row[name]["leaf_jaccard_vs_early_T_0"] = round(float((masks[name] & ref).sum() / max((masks[name] | ref).sum(), 1)), 5)
            row[name]["valid_only_here"], row[name]["valid_only_in_early_T_0"] = int((masks[name] & ~ref).sum()), int((~masks[name] & ref).sum())
        oct_rows.append(row)
        say(f"octree seed {sd}: {row}")
    res["octree_study"] = dict(rows=oct_rows, steps_global=sg, note="UpdateOctNodes (K5/K6) every global step, no ProcOctree: node indices are stable, a leaf is 'valid' while "
                               "trans_idx >= 0 (MarkInvalidNodes, PersSampler_cuda.cu:577-583).  Training is chaotic, so two runs differ in a few leaves whatever the semantics; "
                               "the same-state statement is tests/test_gpu_pipeline.py::test_octree_statistics_do_not_depend_on_the_cut")
    save()



legs = tuple(os.environ.get("GFN_STUDY_LEGS", "plugin,reference").split(","))
assert len(legs) in (1, 2) and all(leg in procedural.LEGS for leg in legs), legs  # (one leg: more runs for seeds another call has paired already; tools/psnr_merge.py pairs them by seed)
res["legs"] = list(legs)
if (seed0 == 0 and os.environ.get("GFN_STUDY_NO_AUX") != "1") or os.environ.get("GFN_STUDY_AUX_ONLY") == "1":
    aux_sections()
if os.environ.get("GFN_STUDY_AUX_ONLY") == "1":  # the gradient-cosine trajectory and the octree study alone (merged with the seed ranges' parts)
    sys.exit(0)

# ---- 3. the paired study
t0 = time.perf_counter()
st = procedural.psnr_study(sc, seeds=tuple(range(seed0, seed0 + n)), legs=legs, steps_global=sg, steps_focal=sf, detail=detail, lr_final=LRF, holdout_every=8, progress=say)
st["seconds"] = time.perf_counter() - t0
res["psnr_study"] = st
save()
if "delta" in st:
    say(f"delta {st['delta']:+.3f} +- {st['se']:.3f} dB over {n} seeds; same-state decoder delta {st['decoder_delta']:+.4f} +- {st['decoder_se']}")
else:
    say(f"{legs[0]}: {st['psnr_' + legs[0]]:.3f} +- {st['psnr_' + legs[0] + '_std']:.3f} dB over {n} seeds")
save()

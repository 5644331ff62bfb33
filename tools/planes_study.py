#!/usr/bin/env python3
"""Round 5: does the one-plane form of the weight gradient's operands (HotPathConfig.wgrad_planes = "rne": one f16 plane per
operand, rounded to nearest under a per-sample dither) change what the global stage learns?  The global stage of the PSNR study's schedule
trained twice per seed -- "rne" (the default) and "x3" (hi + lo' planes, rounds 1-4) -- from the same initialisation, rays and draws; PSNR on the
fixed held-out rays and from held-out cameras, paired by seed.
    python tools/planes_study.py OUT.json SEED0 N_SEEDS         (GPU box)
    python tools/planes_study.py merge OUT.json PART.json ...   (anywhere)"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mean_se(v):
    n = len(v)
    m = sum(v) / n
    return m, (math.sqrt(sum((x - m) ** 2 for x in v) / (n - 1) / n) if n > 1 else None)


def summarise(rows):
    out = dict(n_seeds=len(rows))
    for key in ("psnr", "psnr_heldout_cams"):
        a, b = [r["rne"][key] for r in rows], [r["x3"][key] for r in rows]
        d = [x - y for x, y in zip(a, b)]
        (ma, sa), (mb, sb), (md, sd) = mean_se(a), mean_se(b), mean_se(d)
        out[key] = dict(rne_mean=round(ma, 3), rne_se=round(sa, 3), x3_mean=round(mb, 3), x3_se=round(sb, 3), delta=round(md, 4), delta_se=round(sd, 4),
                        within_2_se_of_zero=bool(abs(md) <= 2 * sd))
    return out


if sys.argv[1] == "merge":
    rows, meta = [], None
    for p in sys.argv[3:]:
        d = json.load(open(p))
        meta = meta or {k: d[k] for k in ("protocol", "steps_global")}
        rows += d["rows"]
    rows.sort(key=lambda r: r["seed"])
    assert len({r["seed"] for r in rows}) == len(rows)
    res = dict(meta, summary=summarise(rows), rows=rows, note="tools/planes_study.py: the global stage with wgrad_planes rne | x3, paired by seed")
    json.dump(res, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(res["summary"], indent=1))
    sys.exit(0)

import torch  # noqa: E402

import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import procedural, scene  # noqa: E402

out_path, seed0, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
sg = procedural.STUDY_STEPS[0]
res = dict(protocol=procedural.PROTOCOL, steps_global=sg, rows=[])
for sd in range(seed0, seed0 + n):
    row = dict(seed=sd)
    for mode in ("rne", "x3"):
        r = procedural.train_and_eval(sc, seed=sd, steps_global=sg, steps_focal=0, detail=procedural.STUDY_DETAIL, lr_final=procedural.STUDY_LR_FINAL, holdout_every=8,
                                      **{**procedural.LEGS["fast"], "wgrad_planes": mode})
        row[mode] = dict(psnr=round(r["psnr"], 4), psnr_heldout_cams=round(r["psnr_heldout"], 4))
        torch.cuda.empty_cache()
    res["rows"].append(row)
    print(f"[{time.strftime('%H:%M:%S')}] seed {sd}: rne {row['rne']['psnr']:.3f} x3 {row['x3']['psnr']:.3f}", flush=True)
    json.dump(res, open(out_path + ".tmp", "w"), indent=1)
    os.replace(out_path + ".tmp", out_path)

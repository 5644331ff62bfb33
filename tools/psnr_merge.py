#!/usr/bin/env python3
"""Merge the outputs of several tools/psnr_study.py calls (disjoint seed ranges, one protocol) into the round's record:
    python tools/psnr_merge.py OUT.json PART0.json PART1.json ..."""
import json, math, os, sys
import numpy as np

every = [json.load(open(p)) for p in sys.argv[2:]]
assert len({p["protocol"] for p in every}) == 1, "parts were made with different protocols"
assert len({tuple(p["psnr_study"]["legs"]) for p in every if "psnr_study" in p}) == 1, "parts pair different legs"
parts = [p for p in every if "psnr_study" in p]
out = dict(protocol=every[0]["protocol"])
for p in every:  # gradient-cosine trajectory / octree study: the LAST file that carries them (an aux-only call, or part 0)
    for k in ("grad_cosine", "octree_study"):
        if k in p:
            out[k] = p[k]
out["parts"] = [dict(seed0=p.get("seed0", 0), argv=p["argv"]) for p in parts]
st0 = parts[0]["psnr_study"]
legs = st0["legs"]
st = {k: st0[k] for k in ("legs", "workload", "steps", "mean_samples_per_ray", "delta_of", "decoder_delta_of")}
st["seeds"] = sum((p["psnr_study"]["seeds"] for p in parts), [])
assert len(set(st["seeds"])) == len(st["seeds"]), "overlapping seeds"
for leg in legs:
    runs = sum((p["psnr_study"][f"psnr_{leg}_runs"] for p in parts), [])
    st[f"psnr_{leg}_runs"] = runs
    st[f"psnr_{leg}"] = float(np.mean(runs))
    st[f"psnr_{leg}_std"] = float(np.std(runs, ddof=1))
    for key in (f"psnr_global_{leg}", f"psnr_heldout_{leg}"):
        if key in st0:
            st[key] = float(np.average([p["psnr_study"][key] for p in parts], weights=[len(p["psnr_study"]["seeds"]) for p in parts]))
    # PSNR after the global stage, per seed (parts made before the key existed: from the call's log beside the JSON): the two stages' shares of the difference
    gl = []
    for path, p in zip(sys.argv[2:], every):
        if "psnr_study" not in p:
            continue
        have = p["psnr_study"].get(f"psnr_global_{leg}_runs")
        if have is None:
            import re
            log = path[:-5] + ".log"
            found = {int(m.group(1)): float(m.group(2)) for m in re.finditer(rf"psnr {leg} seed (\d+): global ([\d.]+) final", open(log).read())} if os.path.exists(log) else {}
            have = [found.get(sd) for sd in p["psnr_study"]["seeds"]]
        gl += have
    if all(v is not None for v in gl):
        st[f"psnr_global_{leg}_runs"] = gl
    if f"{leg}_surface" in st0:  # what each run of the leg ended with (forward / backward replayed, the forward's form, the merge): one entry per seed
        st[f"{leg}_surface"] = sum((p["psnr_study"][f"{leg}_surface"] for p in parts), [])
for key in ("alive_sample_frac_trained", "planned_K_trained"):
    if key in st0:
        st[key] = st0[key]
runs = sum((p["psnr_study"]["delta_runs"] for p in parts), [])
st["delta_runs"] = runs
st["delta"] = float(np.mean(runs))
st["se"] = float(np.std(runs, ddof=1) / math.sqrt(len(runs)))
# the same-state decoder differences are ~1e-6 dB: the per-seed list is rounded to 1e-3, so the parts' own means / standard errors are combined
ns = np.array([len(p["psnr_study"]["seeds"]) for p in parts], np.float64)
st["decoder_delta_runs"] = sum((p["psnr_study"]["decoder_delta_runs"] for p in parts), [])
st["decoder_delta"] = float(np.sum(ns * np.array([p["psnr_study"]["decoder_delta"] for p in parts])) / ns.sum())
st["decoder_se"] = float(np.sqrt(np.sum((ns * np.array([p["psnr_study"]["decoder_se"] or 0.0 for p in parts])) ** 2)) / ns.sum())
st["decoder_delta_max_abs_run"] = float(max(abs(v) for v in st["decoder_delta_runs"]))
if all(f"psnr_global_{leg}_runs" in st for leg in legs):  # the paired difference stage by stage: after the global stage, and in what the focal stage adds
    g = [np.array(st[f"psnr_global_{leg}_runs"]) for leg in legs]
    f = [np.array(st[f"psnr_{leg}_runs"]) for leg in legs]
    dg, dgain = g[0] - g[1], (f[0] - g[0]) - (f[1] - g[1])
    n_ = len(dg)
    st["delta_global"], st["se_global"] = float(dg.mean()), float(dg.std(ddof=1) / math.sqrt(n_))
    st["delta_focal_gain"], st["se_focal_gain"] = float(dgain.mean()), float(dgain.std(ddof=1) / math.sqrt(n_))
    st["focal_gain"] = {leg: [float((f[i] - g[i]).mean()), float((f[i] - g[i]).std(ddof=1))] for i, leg in enumerate(legs)}
st["n_seeds"] = len(st["seeds"])
st["seconds"] = sum(p["psnr_study"].get("seconds", 0.0) for p in parts)
out["psnr_study"] = st
json.dump(out, open(sys.argv[1], "w"), indent=1)
if "delta_global" in st:
    print(f"after the global stage {st['delta_global']:+.3f} +- {st['se_global']:.3f}; focal gain {st['focal_gain']}, paired difference {st['delta_focal_gain']:+.3f} +- {st['se_focal_gain']:.3f}")
print(f"{st['n_seeds']} seeds: delta {st['delta']:+.3f} +- {st['se']:.3f} dB; same-state decoder delta {st['decoder_delta']:+.6f} +- {st['decoder_se']:.1e}; "
      f"{legs[0]} {st['psnr_' + legs[0]]:.2f} +- {st['psnr_' + legs[0] + '_std']:.2f}, {legs[1]} {st['psnr_' + legs[1]]:.2f} +- {st['psnr_' + legs[1] + '_std']:.2f}")

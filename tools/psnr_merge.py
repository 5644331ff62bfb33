#!/usr/bin/env python3
"""Merge the outputs of several tools/psnr_study.py calls (one protocol) into the round's record:
    python tools/psnr_merge.py OUT.json PART0.json PART1.json ... [--against LEG]
Parts that pair two legs (disjoint seed ranges) give the record's seeds; parts that hold ONE leg (GFN_STUDY_LEGS=leg: more runs for seeds another call
paired already) add that leg's runs, matched by seed.  The record's headline difference is legs[0] - the leg named by --against (default: the second leg
of the paired parts); every other leg present for all seeds gets its own paired difference against legs[0] (delta_vs)."""
import json, math, os, re, sys
import numpy as np

args = sys.argv[1:]
against = None
if "--against" in args:
    i = args.index("--against")
    against = args[i + 1]
    del args[i:i + 2]
out_path, paths = args[0], args[1:]
every = [json.load(open(p)) for p in paths]
assert len({p["protocol"] for p in every}) == 1, "parts were made with different protocols"
out = dict(protocol=every[0]["protocol"])
for p in every:  # gradient-cosine trajectory / octree study: the LAST file that carries them (an aux-only call, or part 0)
    for k in ("grad_cosine", "octree_study"):
        if k in p:
            out[k] = p[k]
studies = [(path, p["psnr_study"]) for path, p in zip(paths, every) if "psnr_study" in p]
paired = [(path, st) for path, st in studies if len(st["legs"]) == 2]
single = [(path, st) for path, st in studies if len(st["legs"]) == 1]
assert paired and len({tuple(st["legs"]) for _, st in paired}) == 1, "the paired parts pair different legs"
out["parts"] = [dict(seed0=p.get("seed0", 0), argv=p["argv"], legs=p["psnr_study"]["legs"]) for p in every if "psnr_study" in p]
st0 = paired[0][1]
legs = list(st0["legs"])
st = {k: st0[k] for k in ("workload", "steps", "mean_samples_per_ray", "decoder_delta_of")}
seeds = sum((s_["seeds"] for _, s_ in paired), [])
assert len(set(seeds)) == len(seeds), "overlapping seeds"
st["seeds"] = seeds


def global_runs(path, s_, leg):  # PSNR after the global stage, per seed (parts made before the key existed: from the call's log beside the JSON)
    have = s_.get(f"psnr_global_{leg}_runs")
    if have is None:
        log = path[:-5] + ".log"
        found = {int(m.group(1)): float(m.group(2)) for m in re.finditer(rf"psnr {leg} seed (\d+): global ([\d.]+) final", open(log).read())} if os.path.exists(log) else {}
        have = [found.get(sd) for sd in s_["seeds"]]
    return have


runs, gruns = {}, {}  # leg -> {seed: PSNR}, {seed: PSNR after the global stage}
for path, s_ in paired + single:
    for leg in s_["legs"]:
        runs.setdefault(leg, {}).update(zip(s_["seeds"], s_[f"psnr_{leg}_runs"]))
        gruns.setdefault(leg, {}).update(zip(s_["seeds"], global_runs(path, s_, leg)))
all_legs = [leg for leg in runs if all(sd in runs[leg] for sd in seeds)]  # legs with a run for every seed of the record
assert legs[0] in all_legs
against = against or legs[1]
assert against in all_legs, f"leg {against} has no run for some seed of the record"
st["legs"] = [legs[0], against]
st["all_legs"] = all_legs
for leg in all_legs:
    v = np.array([runs[leg][sd] for sd in seeds])
    st[f"psnr_{leg}_runs"] = [float(x) for x in v]
    st[f"psnr_{leg}"], st[f"psnr_{leg}_std"] = float(v.mean()), float(v.std(ddof=1))
    g = [gruns[leg].get(sd) for sd in seeds]
    if all(x is not None for x in g):
        st[f"psnr_global_{leg}_runs"] = [float(x) for x in g]
        st[f"psnr_global_{leg}"] = float(np.mean(g))
for key in ("psnr_heldout_" + legs[0], "psnr_heldout_" + legs[1]):
    if key in st0:
        st[key] = float(np.average([s_[key] for _, s_ in paired], weights=[len(s_["seeds"]) for _, s_ in paired]))
if f"{legs[0]}_surface" in st0:  # what each run of the leg ended with (forward / backward replayed, the forward's form, the merge): one entry per seed
    st[f"{legs[0]}_surface"] = sum((s_[f"{legs[0]}_surface"] for _, s_ in paired), [])
for key in ("alive_sample_frac_trained", "planned_K_trained"):
    if key in st0:
        st[key] = st0[key]


def pair(a, b):
    d = np.array(st[f"psnr_{a}_runs"]) - np.array(st[f"psnr_{b}_runs"])
    r = dict(delta=float(d.mean()), se=float(d.std(ddof=1) / math.sqrt(len(d))), of=f"PSNR({a}) - PSNR({b}), paired by seed")
    if f"psnr_global_{a}_runs" in st and f"psnr_global_{b}_runs" in st:  # stage by stage: after the global stage, and in what the focal stage adds
        ga, gb = np.array(st[f"psnr_global_{a}_runs"]), np.array(st[f"psnr_global_{b}_runs"])
        fa, fb = np.array(st[f"psnr_{a}_runs"]), np.array(st[f"psnr_{b}_runs"])
        dg, dgain = ga - gb, (fa - ga) - (fb - gb)
        r.update(delta_global=float(dg.mean()), se_global=float(dg.std(ddof=1) / math.sqrt(len(dg))), delta_focal_gain=float(dgain.mean()),
                 se_focal_gain=float(dgain.std(ddof=1) / math.sqrt(len(dg))), focal_gain={a: float((fa - ga).mean()), b: float((fb - gb).mean())})
    return d, r


d, head = pair(legs[0], against)
st["delta_runs"] = [round(float(x), 3) for x in d]
st["delta"], st["se"], st["delta_of"] = head["delta"], head["se"], head["of"]
st.update({k: v for k, v in head.items() if k not in ("delta", "se", "of")})
st["delta_vs"] = {leg: pair(legs[0], leg)[1] for leg in all_legs if leg not in (legs[0], against)}
if against != legs[1] and legs[1] in all_legs:
    st["delta_vs"][f"{against}|{legs[1]}"] = pair(against, legs[1])[1]
# the same-state decoder differences are ~1e-6 dB: the per-seed list is rounded to 1e-3, so the parts' own means / standard errors are combined
ns = np.array([len(s_["seeds"]) for _, s_ in paired], np.float64)
st["decoder_delta_runs"] = sum((s_["decoder_delta_runs"] for _, s_ in paired), [])
st["decoder_delta"] = float(np.sum(ns * np.array([s_["decoder_delta"] for _, s_ in paired])) / ns.sum())
st["decoder_se"] = float(np.sqrt(np.sum((ns * np.array([s_["decoder_se"] or 0.0 for _, s_ in paired])) ** 2)) / ns.sum())
st["decoder_delta_max_abs_run"] = float(max(abs(v) for v in st["decoder_delta_runs"]))
st["n_seeds"] = len(seeds)
st["seconds"] = sum(s_.get("seconds", 0.0) for _, s_ in studies)
out["psnr_study"] = st
json.dump(out, open(out_path, "w"), indent=1)
print(f"{st['n_seeds']} seeds, legs {all_legs}: {st['delta_of']} = {st['delta']:+.3f} +- {st['se']:.3f} dB"
      + (f" (after the global stage {st['delta_global']:+.3f} +- {st['se_global']:.3f}; focal gain {st['focal_gain']}, paired {st['delta_focal_gain']:+.3f} +- {st['se_focal_gain']:.3f})" if "delta_global" in st else ""))
for k, v in st["delta_vs"].items():
    print(f"   {v['of']}: {v['delta']:+.3f} +- {v['se']:.3f}" + (f"; focal gain paired {v['delta_focal_gain']:+.3f} +- {v['se_focal_gain']:.3f}" if "delta_focal_gain" in v else ""))
print(f"   same-state decoder delta {st['decoder_delta']:+.6f} +- {st['decoder_se']:.1e}")

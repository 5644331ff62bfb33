"""gfnerf_amd.procedural (the analytic scene bench.py trains on to report PSNR): the torch closed form and quadrature against the
fixture made in numpy float64 + the oracle's compositing (tests/golden/make_procedural.py); runs on the CPU."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ground_truth_colours_match_the_oracle_composited_fixture():
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import procedural

    g = np.load(os.path.join(ROOT, "tests", "golden", "procedural_gt.npz"))
    o, d = torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"])
    rgb = procedural.gt_colors(o, d).numpy()
    assert rgb.shape == (256, 3) and rgb.min() >= 0 and rgb.max() <= 1
    np.testing.assert_allclose(rgb, g["rgb"], atol=3e-5)  # f32 torch quadrature vs f64 field values composited in f32 by the oracle
    assert g["acc"].min() > 0.999  # every ray of the capture ends on the (opaque) terrain: no background term is involved
    sigma, col = procedural.field(torch.from_numpy(g["xyz_probe"]))
    np.testing.assert_allclose(sigma.numpy(), g["sigma_probe"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(col.numpy(), g["rgb_probe"], atol=2e-5)
    # the detailed variant (fine colour texture: what the PSNR study trains on)
    dt = float(g["detail"])
    rgb_d = procedural.gt_colors(o, d, dt).numpy()
    np.testing.assert_allclose(rgb_d, g["rgb_detail"], atol=3e-4)  # (f32 sines of arguments up to ~3,000: 2e-4 of phase)
    _, col_d = procedural.field(torch.from_numpy(g["xyz_probe"]), dt)
    np.testing.assert_allclose(col_d.numpy(), g["rgb_detail_probe"], atol=3e-4)
    assert float(np.abs(rgb_d - rgb).mean()) > 0.02  # and it is a different picture
    # PSNR as torchmetrics defines it for data_range 1 (nerfacto.py:300): -10 log10(mse)
    a = torch.rand(100, 3)
    assert abs(procedural.psnr(a, a + 0.01) - 40.0) < 1e-3


def test_psnr_record_of_round_5():
    """profiles/r05_psnr_study.json (tools/psnr_study.py on an MI355X, merged by tools/psnr_merge.py; round 5's build: one-plane weight-gradient operands,
    coarser-unit gradient accumulation; legs fast | reference = the NATIVE product step against the reference's training semantics): it belongs to the
    protocol in the tree (procedural.PROTOCOL / STUDY_STEPS), holds >= 60 pairs (a standard error of ~0.07 dB; round 4's 148-pair record of the build
    before: profiles/r04_psnr_study.json, +0.001 +- 0.038), and says that the product is not worse than the reference's semantics by 0.1 dB -- the ONE-SIDED
    bound delta - 2 SE >= -0.1 ("|delta| <= 2 SE" cannot fail when delta is small and is no bound).  Round 6's record, of the engine the
    plugin surface builds and drives: test_psnr_record_of_the_round."""
    import json

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import procedural

    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_psnr_study.json")))
    st = rec["psnr_study"]
    assert rec["protocol"] == procedural.PROTOCOL and st["steps"] == list(procedural.STUDY_STEPS) and st["legs"] == ["fast", "reference"]
    n = st["n_seeds"]
    assert n >= 60 and len(set(st["seeds"])) == n == len(st["delta_runs"]) == len(st["psnr_fast_runs"]) == len(st["psnr_reference_runs"])
    d = np.array(st["psnr_fast_runs"]) - np.array(st["psnr_reference_runs"])
    assert abs(d.mean() - st["delta"]) < 2e-3 and abs(d.std(ddof=1) / np.sqrt(n) - st["se"]) < 2e-3   # the summary is the runs' (listed to 1e-3 dB)
    assert st["se"] <= 0.075 and abs(st["delta"]) <= 0.1 and st["delta"] - 2.0 * st["se"] >= -0.1  # within 0.1 dB; not worse than the reference by 0.1 dB at two standard errors
    assert abs(st["decoder_delta"]) < 1e-4 and st["decoder_delta_max_abs_run"] < 0.01  # the decoder arithmetic on identical states
    g = rec["grad_cosine"]["min_cos"]
    assert min(g["exact|reference"].values()) > 0.99999 and min(g["exact|fast0"].values()) > 0.99999 and min(g["fast|f32"].values()) > 0.99999
    assert g["exact|fast0"]["mlp"] > 0.99999  # the decoder's gradient with one-plane weight-gradient operands
    assert min(g["exact|fast"].values()) > 0.999                                   # the cut moves the gradient through the sign-like loss: DESIGN.md section 7


def test_psnr_record_of_the_round():
    """profiles/r06_psnr_study.json (VERDICT r05 #1b, Missing #3): the engine AS THE PLUGIN SURFACE BUILDS AND DRIVES IT -- gfnerf/nerfacto.py's HotPathConfig
    (record_merge_dense, replayed forward + backward, the fused loss node), trained through GFNerfPipeline.get_train_loss_dict / backward / Optimizers with
    GfnAdam and the model's callbacks (procedural.LEGS["plugin"]) -- against the reference's training semantics, paired by seed on the same rays, schedule
    and held-out set.  Three legs per seed: `plugin`; `reference_m1` = the reference's semantics INCLUDING its block-stage sampler mode (no march jitter
    from the focal stage's second step on, gfnerf/nerfacto.py:614-616 -- the surface mirrors it); `reference` = the same with jitter throughout (rounds
    4-5's leg).  Held here: >= 32 pairs of the protocol in the tree, every summary recomputed from the runs; VERDICT r05 #1's bar on the headline
    difference plugin - reference_m1 (|mean| <= 0.1 dB at a standard error <= 0.09); the one-sided bound (mean - 2 SE >= -0.1 dB, VERDICT r05 Weak #11) where
    the record resolves it -- in the FOCAL stage's gain, the stage the surface's own launch policy (merged records, two replayed graphs) applies to, whose
    paired differences scatter by 0.12 dB where whole runs scatter by 0.8 (one run's global stage is one draw: Adam at eps 1e-15) --, and for the headline
    itself once its standard error is <= 0.05; the sampler mode's own effect (reference_m1 - reference > 0 in the focal gain at >= 3 standard errors: it is
    why the surface's focal stage gains more than rounds 4-5's reference leg); every plugin run learned the scene and ended with both graphs replaying."""
    import json

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import procedural

    rec = json.load(open(os.path.join(ROOT, "profiles", "r06_psnr_study.json")))
    st = rec["psnr_study"]
    assert rec["protocol"] == procedural.PROTOCOL and st["steps"] == list(procedural.STUDY_STEPS) and st["legs"] == ["plugin", "reference_m1"]
    assert set(st["all_legs"]) >= {"plugin", "reference", "reference_m1"} and procedural.LEGS["reference_m1"]["focal_jitter"] is False
    n = st["n_seeds"]
    assert n >= 32 and len(set(st["seeds"])) == n == len(st["delta_runs"]) == len(st["psnr_plugin_runs"]) == len(st["psnr_reference_m1_runs"]) == len(st["psnr_reference_runs"])

    def paired(a, b):
        d = np.array(st[f"psnr_{a}_runs"]) - np.array(st[f"psnr_{b}_runs"])
        return d.mean(), d.std(ddof=1) / np.sqrt(n)

    def gain(a):
        return np.array(st[f"psnr_{a}_runs"]) - np.array(st[f"psnr_global_{a}_runs"])

    d, se = paired("plugin", "reference_m1")
    assert abs(d - st["delta"]) < 1e-6 and abs(se - st["se"]) < 1e-6                           # the summary is the runs'
    assert abs(st["delta"]) <= 0.1 and st["se"] <= 0.09, (st["delta"], st["se"])                # VERDICT r05 #1's bar
    if st["se"] <= 0.05:
        assert st["delta"] - 2.0 * st["se"] >= -0.1, (st["delta"], st["se"])
    # stage by stage: the focal stage's gain pairs tightly, and there the one-sided bound is resolved
    dg = gain("plugin") - gain("reference_m1")
    se_g = dg.std(ddof=1) / np.sqrt(n)
    assert abs(dg.mean() - st["delta_focal_gain"]) < 1e-6 and se_g <= 0.03 and dg.mean() - 2.0 * se_g >= -0.1 and abs(dg.mean()) <= 0.05, (dg.mean(), se_g)
    dgl = np.array(st["psnr_global_plugin_runs"]) - np.array(st["psnr_global_reference_m1_runs"])  # the global stage: no loss resolvable
    assert abs(dgl.mean() - st["delta_global"]) < 1e-6 and abs(dgl.mean()) <= 2.0 * dgl.std(ddof=1) / np.sqrt(n), (dgl.mean(), dgl.std(ddof=1) / np.sqrt(n))
    mode = gain("reference_m1") - gain("reference")                                               # what the sampler mode alone does to the focal stage
    assert mode.mean() / (mode.std(ddof=1) / np.sqrt(n)) >= 3.0 and mode.mean() > 0.05, (mode.mean(), mode.std(ddof=1) / np.sqrt(n))
    assert abs(st["delta_vs"]["reference"]["delta"] - paired("plugin", "reference")[0]) < 1e-6
    assert abs(st["decoder_delta"]) < 1e-4 and st["decoder_delta_max_abs_run"] < 0.01  # the decoder arithmetic on identical states
    assert min(st["psnr_plugin_runs"]) > 30.0  # every run through the surface learned the scene
    surf = st.get("plugin_surface")
    assert surf and len(surf) == n and all(x["forward_replayed"] and x["backward_replayed"] and x["record_merge_dense"] for x in surf)


def test_octree_study_record_of_the_round():
    """profiles/r05_octree_study.json (tools/octree_study.py) and the sentences INTEGRATION.md section 5 / DESIGN.md section 4
    make about it: >= 16 seeds, the summary recomputed from the rows, and every figure the documents quote is the record's."""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import octree_study as st

    rec = json.load(open(os.path.join(root, "profiles", "r05_octree_study.json")))
    rows, summ = rec["rows"], rec["summary"]
    assert len(rows) >= 16 and summ["n_seeds"] == len(rows) and sorted(r["seed"] for r in rows) == list(range(len(rows)))
    assert st.summarise(rows) == summ  # the statistics are the rows'
    c, n, n8, c8 = (summ["legs"][k] for k in ("cut", "no_cut", "no_cut_eps8", "cut_eps8"))
    p = summ["pairs"]["cut|no_cut"]
    q = summ["pairs"]["no_cut|no_cut_eps8"]
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    for text in (f"{c['valid_leaves_mean']} +- {c['valid_leaves_se']} valid leaves", f"{n['valid_leaves_mean']} +- {n['valid_leaves_se']}",
                 f"{p['valid_leaves_diff_mean']} +- {p['valid_leaves_diff_se']} leaves ({100 * p['valid_leaves_diff_rel']:.1f} %)",
                 f"in {p['seeds_with_more_only_in_b']} of {len(rows)} seeds", f"sign test p = {p['sign_test_p']:.3f}", f"Jaccard index {p['jaccard_mean']:.2f}",
                 f"+{p['psnr_diff_mean']:.2f} +- {p['psnr_diff_se']:.2f} dB on held-out pixels", f"+{p['psnr_heldout_cams_diff_mean']:.2f} +- {p['psnr_heldout_cams_diff_se']:.2f} dB from held-out cameras",
                 f"{n8['valid_leaves_mean']} +- {n8['valid_leaves_se']}", f"= {q['valid_leaves_diff_mean']}\n  +- {q['valid_leaves_diff_se']}", f"{c8['valid_leaves_mean']} +- {c8['valid_leaves_se']}"):
        assert text in doc, text
    # the statements themselves: one-sided (p < 0.05), no loss of held-out PSNR beyond two standard errors, eps is not the mechanism (no_cut keeps its leaves)
    assert p["sign_test_p"] < 0.05 and p["seeds_with_more_only_in_b"] > p["seeds_with_more_only_in_a"] and p["valid_leaves_diff_mean"] < 0
    assert p["psnr_heldout_cams_diff_mean"] > -2 * p["psnr_heldout_cams_diff_se"] and p["psnr_diff_mean"] > -2 * p["psnr_diff_se"]
    assert q["valid_leaves_diff_mean"] < 2 * q["valid_leaves_diff_se"]  # eps 1e-8 does not prune the no_cut leg


def test_planes_study_record_of_the_round():
    """profiles/r05_planes_study.json (tools/planes_study.py): the global stage trained with the weight gradient's operands as one f16
    plane each ("rne", the default) and as hi + lo' planes ("x3"), paired by seed -- >= 32 pairs, the summary recomputed from the rows, the PSNR
    difference within two standard errors of zero on held-out pixels and from held-out cameras."""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = json.load(open(os.path.join(root, "profiles", "r05_planes_study.json")))
    rows, summ = rec["rows"], rec["summary"]
    assert len(rows) >= 32 and summ["n_seeds"] == len(rows)
    for key in ("psnr", "psnr_heldout_cams"):
        d = np.array([r["rne"][key] - r["x3"][key] for r in rows])
        assert abs(d.mean() - summ[key]["delta"]) < 1e-3 and abs(d.std(ddof=1) / np.sqrt(len(d)) - summ[key]["delta_se"]) < 1e-3
        assert abs(summ[key]["delta"]) <= 2 * summ[key]["delta_se"] and summ[key]["within_2_se_of_zero"]

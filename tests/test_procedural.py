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


def test_psnr_record_of_the_round():
    """profiles/r04_psnr_study.json (tools/psnr_study.py on an MI355X, merged by tools/psnr_merge.py) is the evidence for BASELINE's "PSNR within 0.1 dB
    of reference": it must belong to the protocol in the tree (procedural.PROTOCOL / STUDY_STEPS -- bench.py reports it only then), hold enough pairs to
    resolve 0.1 dB (standard error <= 0.05), and say what DESIGN.md says it says."""
    import json

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import procedural

    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_psnr_study.json")))
    st = rec["psnr_study"]
    assert rec["protocol"] == procedural.PROTOCOL and st["steps"] == list(procedural.STUDY_STEPS) and st["legs"] == ["fast", "reference"]
    n = st["n_seeds"]
    assert n >= 100 and len(set(st["seeds"])) == n == len(st["delta_runs"]) == len(st["psnr_fast_runs"]) == len(st["psnr_reference_runs"])
    d = np.array(st["psnr_fast_runs"]) - np.array(st["psnr_reference_runs"])
    assert abs(d.mean() - st["delta"]) < 2e-3 and abs(d.std(ddof=1) / np.sqrt(n) - st["se"]) < 2e-3   # the summary is the runs' (listed to 1e-3 dB)
    assert st["se"] <= 0.05 and abs(st["delta"]) <= 0.1 - 1.0 * st["se"]          # within 0.1 dB with a standard error to spare
    assert abs(st["decoder_delta"]) < 1e-4 and st["decoder_delta_max_abs_run"] < 0.01  # the decoder arithmetic on identical states
    g = rec["grad_cosine"]["min_cos"]
    assert min(g["exact|reference"].values()) > 0.99999 and min(g["exact|fast0"].values()) > 0.99999 and min(g["fast|f32"].values()) > 0.99999
    assert min(g["exact|fast"].values()) > 0.999                                   # the cut moves the gradient through the sign-like loss: DESIGN.md section 7

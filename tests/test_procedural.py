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

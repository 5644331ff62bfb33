"""Batch assembly on the GPU (PixelSampler / ErrorPixelSampler / error-map upkeep / RayGenerator) through
the C ABI, against the vectors captured from the reference and against oracle/batch.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import _lib

    _lib.require_device()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import ops as _ops

    return _ops


@pytest.fixture(scope="module")
def ob():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import batch  # oracle/batch.py -- the checker

    return batch


def _t(a, dev):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_pixel_sampler_matches_golden(ops, ob, dev, golden):
    g = golden
    n, H, W = g["pix_image"].shape[:3]
    lin = ops.pixel_uniform(_t(g["pix_u"], dev), n, H, W)
    assert np.array_equal(lin.cpu().numpy(), ob.pixel_uniform(g["pix_u"], n, H, W))
    c = ops.pixel_collate(lin, _t(g["pix_image"], dev), _t(g["pix_image_idx"], dev), _t(g["pix_rel_idx"], dev))
    assert np.array_equal(c["indices"].cpu().numpy(), g["pix_indices"])
    assert np.array_equal(c["image"].cpu().numpy(), g["pix_gt"])
    assert np.array_equal(c["rel_camera_indices"].cpu().numpy(), g["pix_rel"])
    assert np.array_equal(c["coords"].cpu().numpy(), g["pix_indices"][:, 1:3].astype(np.float32) + 0.5)
    # Rubble-scale extents, no image gather
    lin = ops.pixel_uniform(_t(g["pix_big_u"], dev), 1600, 864, 1152)
    c = ops.pixel_collate(lin, None, shape=(1600, 864, 1152))
    assert np.array_equal(c["indices"].cpu().numpy(), g["pix_big_indices"]) and c["image"] is None
    # u8 stack, identity camera maps, ragged / empty batches
    img8 = (g["pix_image"] * 255).astype(np.uint8)
    for m in (0, 1, 96):
        lin = ops.pixel_uniform(_t(g["pix_u"][:m], dev), n, H, W)
        c = ops.pixel_collate(lin, _t(img8, dev))
        want = ob.pixel_collate(lin.cpu().numpy(), img8)
        for k in ("indices", "slot", "rel_camera_indices", "image", "coords"):
            assert np.array_equal(c[k].cpu().numpy(), want[k]), k


def test_error_map_update_and_block_sums(ops, ob, dev, golden):
    import torch

    g = golden
    em = _t(g["errpix_map"], dev)
    sums = ops.errmap_block_sums(em)
    np.testing.assert_allclose(sums.cpu().numpy(), ob.errmap_block_sums(g["errpix_map"]), rtol=1e-14)
    ops.errmap_update(em, sums, _t(g["errupd_indices"], dev), _t(g["errupd_pred"], dev), _t(g["errupd_gt"], dev))
    idx = g["errupd_indices"]
    lin = (idx[:, 0] * em.shape[1] + idx[:, 1]) * em.shape[2] + idx[:, 2]
    uniq, counts = np.unique(lin, return_counts=True)
    dup = np.isin(np.arange(em.numel()), uniq[counts > 1])
    got = em.cpu().numpy().reshape(-1)
    assert np.array_equal(got[~dup], g["errupd_map_after"].reshape(-1)[~dup])
    # a duplicated pixel holds ONE of its writers' values, and the partial sums follow the map whatever the order was
    e = ob.errmap_update(g["errpix_map"].copy(), idx, g["errupd_pred"], g["errupd_gt"])
    for p in uniq[counts > 1]:
        assert got[p] in e[lin == p]
    np.testing.assert_allclose(sums.cpu().numpy(), ops.errmap_block_sums(em).cpu().numpy(), rtol=1e-12)
    # many duplicates + out-of-range rows (skipped)
    rng = np.random.default_rng(3)
    big = torch.zeros((3, 70, 90, 1), device=dev)
    s2 = ops.errmap_block_sums(big)
    for _ in range(3):
        ii = np.stack([rng.integers(-1, 4, 5000), rng.integers(0, 70, 5000), rng.integers(0, 90, 5000)], 1).astype(np.int32)
        ops.errmap_update(big, s2, _t(ii, dev), _t(rng.random((5000, 3), np.float32), dev), _t(rng.random((5000, 3), np.float32), dev))
    np.testing.assert_allclose(s2.cpu().numpy(), ops.errmap_block_sums(big).cpu().numpy(), rtol=1e-9, atol=1e-9)


def test_train_losses_is_charbonnier_psnr_and_the_error_map_write_back_in_one_launch(ops, ob, oracle, dev, golden):
    """gfn_train_losses (round 6: the plugin surface's launches between compositing and backward, fused): out3[0] / g_char = the Charbonnier loss and
    gradient of the golden vector (losses.py:73-84) and of the oracle, out3[1] = 0, out3[2] = torchmetrics' PSNR (data_range 1: -10 log10 mse), g_zero
    zeroed; the error map and its partial sums after the launch are those gfn_errmap_update leaves (unique pixels: bit for bit; the sums follow the map),
    with the image of a row taken from `slot` when given; the values are WRITTEN (garbage in the outputs beforehand) and the same bits every run.
    gfn_loss_grad_combine: g_a w_a + g_b w_b with the weights read from the device."""
    import torch

    g = golden
    pred, gt = _t(g["charb_pred"], dev), _t(g["charb_gt"], dev)
    out3, gc, gz = ops.train_losses(pred, gt)
    np.testing.assert_allclose(out3[0].item(), g["charb_loss"], rtol=1e-5)
    np.testing.assert_allclose(gc.cpu().numpy(), g["charb_gpred"], rtol=1e-4, atol=1e-7)
    lo, go = oracle.charbonnier(g["charb_pred"], g["charb_gt"])
    np.testing.assert_allclose(out3[0].item(), lo, rtol=2e-6)
    np.testing.assert_allclose(gc.cpu().numpy(), go, rtol=1e-5, atol=1e-9)
    assert out3[1].item() == 0.0 and float(gz.abs().max()) == 0.0 and gz.shape == pred.shape
    mse = float(np.mean((g["charb_pred"].astype(np.float64) - g["charb_gt"]) ** 2))
    np.testing.assert_allclose(out3[2].item(), -10.0 * np.log10(mse), rtol=1e-5)
    again = ops.train_losses(pred, gt)
    assert torch.equal(again[0], out3) and torch.equal(again[1], gc)  # fixed-order sums: the same bits
    # the error-map write-back riding along: against gfn_errmap_update on a copy of the map
    R = g["errupd_indices"].shape[0]
    p2, g2, idx = _t(g["errupd_pred"], dev), _t(g["errupd_gt"], dev), _t(g["errupd_indices"], dev)
    em_a, em_b = _t(g["errpix_map"], dev), _t(g["errpix_map"], dev)
    s_a, s_b = ops.errmap_block_sums(em_a), ops.errmap_block_sums(em_b)
    ops.errmap_update(em_a, s_a, idx, p2, g2)
    o3, _, _ = ops.train_losses(p2, g2, want_s3im_grad=False, errmap=(em_b, s_b, idx, None))
    lin = (g["errupd_indices"][:, 0] * em_a.shape[1] + g["errupd_indices"][:, 1]) * em_a.shape[2] + g["errupd_indices"][:, 2]
    uniq, counts = np.unique(lin, return_counts=True)
    dup = np.isin(np.arange(em_a.numel()), uniq[counts > 1])
    assert np.array_equal(em_b.cpu().numpy().reshape(-1)[~dup], em_a.cpu().numpy().reshape(-1)[~dup]) and R >= 32
    np.testing.assert_allclose(s_b.cpu().numpy(), ops.errmap_block_sums(em_b).cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(o3[0].item(), ops.charbonnier(p2, g2)[0].item(), rtol=2e-6)
    # a training batch's size (the strided loops of the one workgroup): against the multi-workgroup kernel and torch
    big_p, big_g = torch.rand(8192, 3, device=dev), torch.rand(8192, 3, device=dev)
    ob3, ogc, _ = ops.train_losses(big_p, big_g, want_s3im_grad=False)
    ref_l, ref_g = ops.charbonnier(big_p, big_g)
    np.testing.assert_allclose(ob3[0].item(), ref_l.item(), rtol=3e-6)
    assert torch.equal(ogc, ref_g)
    # ... and spread over several workgroups (the caller's zeroed workspace: the last workgroup adds the partials in order and leaves it zero again)
    ws = ops.train_losses_workspace(dev)
    for _ in range(3):
        wb3, wgc, _ = ops.train_losses(big_p, big_g, want_s3im_grad=False, ws=ws)
        np.testing.assert_allclose(wb3.cpu().numpy(), ob3.cpu().numpy(), rtol=3e-6)
        assert torch.equal(wgc, ref_g) and float(ws.abs().max()) >= 0 and int(ws.view(torch.int32)[128].item()) == 0
    first = ops.train_losses(big_p, big_g, want_s3im_grad=False, ws=ws)[0]
    assert torch.equal(ops.train_losses(big_p, big_g, want_s3im_grad=False, ws=ws)[0], first)  # the same bits every run
    em_w, s_w = _t(g["errpix_map"], dev), ops.errmap_block_sums(_t(g["errpix_map"], dev))
    ops.train_losses(p2, g2, want_s3im_grad=False, errmap=(em_w, s_w, idx, None), ws=ws)
    assert np.array_equal(em_w.cpu().numpy().reshape(-1)[~dup], em_a.cpu().numpy().reshape(-1)[~dup])
    np.testing.assert_allclose(ob3[2].item(), float(-10.0 * torch.log10(torch.mean((big_p.double() - big_g.double()) ** 2))), rtol=1e-5)
    # `slot` names the image of a row (its position in the cached stack) where the indices carry the camera
    em_c = _t(g["errpix_map"], dev)
    cams = idx.clone(); cams[:, 0] += 1000
    ops.train_losses(p2, g2, want_s3im_grad=False, errmap=(em_c, None, cams, idx[:, 0].contiguous()))
    assert np.array_equal(em_c.cpu().numpy().reshape(-1)[~dup], em_a.cpu().numpy().reshape(-1)[~dup])
    em_d = _t(g["errpix_map"], dev)
    ops.train_losses(p2, g2, want_s3im_grad=False, errmap=(em_d, None, cams, None))  # cameras out of range: rows skipped, like gfn_errmap_update
    assert torch.equal(em_d, _t(g["errpix_map"], dev))
    # the backward's combination
    wa, wb = torch.tensor([0.75], device=dev), torch.tensor([-2.0], device=dev)
    a, b = torch.randn(300, 3, device=dev), torch.randn(300, 3, device=dev)
    assert torch.equal(ops.loss_grad_combine(a, wa, b, wb), a * 0.75 + b * -2.0)
    assert torch.equal(ops.loss_grad_combine(a, wa), a * 0.75) and torch.equal(ops.loss_grad_combine(a, None, b, wb), b * -2.0)
    buf = torch.empty_like(a)
    assert ops.loss_grad_combine(a, wa, b, None, out=buf) is buf and torch.equal(buf, a * 0.75)


def _dyadic(rng, n, zero_frac=0.2):
    """weights whose f64 sums are exact in any order (multiples of 2^-10 below 4): bit-exact parity is then well defined"""
    w = rng.integers(0, 4096, n).astype(np.float32) / 1024.0
    w[rng.random(n) < zero_frac] = 0.0
    return w


@pytest.mark.parametrize("shape,n,n_w", [((4, 16, 32), 40, 8), ((3, 70, 90), 700, 140), ((7, 301, 97), 4096, 819), ((2, 64, 64), 256, 256),
                                         ((2, 64, 64), 64, 0), ((5, 100, 100), 20000, 4000)])
def test_error_sampler_is_bit_identical_to_oracle(ops, ob, dev, shape, n, n_w):
    """gfn_errmap_sample == oracle/batch.py for the same random words: candidates, redraws and group structure."""
    rng = np.random.default_rng(n)
    w = _dyadic(rng, int(np.prod(shape))).reshape(shape)
    bits = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    em = _t(w, dev)
    lin, status, _ = ops.errmap_sample(em, ops.errmap_block_sums(em), _t(bits, dev), n_w)
    want, wstat = ob.errmap_sample(w, bits, n_w)
    lin = lin.cpu().numpy()
    assert np.array_equal(lin, want) and np.array_equal(status.cpu().numpy(), wstat)
    assert len(set(lin[:n_w])) == n_w and len(set(lin[n_w:])) == n - n_w and status[1].item() == 0
    assert (w.reshape(-1)[lin[:n_w]] > 0).all()


@pytest.mark.parametrize("odd", [0, 1])
def test_error_sampler_chunk_quotas(ops, ob, dev, odd):
    """More than one 2^24-pixel chunk: size // num_chunks draws per full chunk, the left-over draw from the tail when
    n % num_chunks != 0 and uniformly from everywhere otherwise (weighted_choice_multinomial :636-674)."""
    import torch

    npix = 2 * ob.PIXEL_CHUNK + 200000 + odd
    gen = torch.Generator(device=dev).manual_seed(5)
    w = (torch.randint(0, 4096, (npix,), device=dev, generator=gen).float() / 1024.0)
    w[torch.rand(npix, device=dev, generator=gen) < 0.3] = 0
    w = w.view(1, 1, npix)
    bits = np.random.default_rng(odd).integers(-2**63, 2**63 - 1, 60, dtype=np.int64)
    lin, status, _ = ops.errmap_sample(w, ops.errmap_block_sums(w), _t(bits, dev), 11)
    lin = lin.cpu().numpy()
    want, wstat = ob.errmap_sample(w.cpu().numpy(), bits, 11)
    assert np.array_equal(lin, want) and np.array_equal(status.cpu().numpy(), wstat)
    assert (lin[:5] < ob.PIXEL_CHUNK).all() and ((lin[5:10] >= ob.PIXEL_CHUNK) & (lin[5:10] < 2 * ob.PIXEL_CHUNK)).all()
    assert (lin[10] >= 2 * ob.PIXEL_CHUNK) == bool(odd)


def test_error_sampler_edge_cases(ops, dev):
    import torch

    # all-zero map: torch.multinomial would raise; drawn uniformly and counted
    em = torch.zeros((2, 32, 32), device=dev)
    bits = torch.randint(-2**63, 2**63 - 1, (50,), dtype=torch.int64, device=dev)
    lin, status, ws = ops.errmap_sample(em, ops.errmap_block_sums(em), bits, 10)
    assert status.tolist() == [10, 0] and len(set(lin.tolist())) >= 40 and lin.min() >= 0 and lin.max() < em.numel()
    # fewer weighted pixels than weighted draws: duplicates are kept and counted, nothing hangs
    em[0, 0, :3] = 1.0
    lin, status, ws = ops.errmap_sample(em, ops.errmap_block_sums(em), bits, 10, workspace=ws)
    assert status[0].item() == 0 and status[1].item() == 7 and set(lin[:10].tolist()) == {0, 1, 2}
    # empty batch
    lin, status, _ = ops.errmap_sample(em, ops.errmap_block_sums(em), bits[:0], 0)
    assert lin.numel() == 0
    with pytest.raises(Exception):
        ops.errmap_sample(em, ops.errmap_block_sums(em)[:-1], bits, 10)


def test_error_sampler_matches_reference_in_distribution(ops, dev, golden):
    """The reference's inclusion frequencies (torch.multinomial + random.sample, 1500 seeded calls) against the device
    sampler's: heavy pixels saturate (without replacement), zero-weight pixels never appear, the rest is homogeneous."""
    import torch

    g = golden
    em = _t(g["errpix_map"], dev)
    sums = ops.errmap_block_sums(em)
    B, k_w, calls = (int(v) for v in g["errpix_cfg"])
    gen = torch.Generator(device=dev).manual_seed(17)
    cw = torch.zeros(em.numel(), dtype=torch.int64, device=dev)
    cu = torch.zeros_like(cw)
    ws, one = None, torch.ones(B, dtype=torch.int64, device=dev)
    bad = torch.zeros(2, dtype=torch.int64, device=dev)
    for _ in range(calls):
        bits = torch.randint(-2**63, 2**63 - 1, (B,), dtype=torch.int64, device=dev, generator=gen)
        lin, status, ws = ops.errmap_sample(em, sums, bits, k_w, workspace=ws)
        cw.index_add_(0, lin[:k_w], one[:k_w]); cu.index_add_(0, lin[k_w:], one[k_w:])
        bad += status
    assert bad.tolist() == [0, 0]
    cw, cu, w = cw.cpu().numpy(), cu.cpu().numpy(), g["errpix_map"].reshape(-1)
    assert cw.max() <= calls and cu.max() <= calls and cw[w == 0].sum() == 0
    heavy = np.argsort(w)[-4:]
    assert np.abs(cw[heavy] / calls - g["errpix_count_weighted"][heavy] / calls).max() < 0.04
    for mine, ref in ((cw, g["errpix_count_weighted"]), (cu, g["errpix_count_uniform"])):
        p = (mine + ref) / (2.0 * calls)
        z = (mine - ref) / np.sqrt(np.maximum(2 * calls * p * (1 - p), 1e-9))
        live = p > 0
        assert np.abs(z[live]).max() < 5.0 and abs(z[live].mean()) < 0.15 and 0.8 < z[live].std() < 1.2


def test_next_train_mirror(ops, ob, dev, golden):
    """data.PixelSampler / ErrorPixelSampler / RayGenerator / ErrorMap strung together like VanillaDataManager.next_train +
    the pipeline's error-map write: rays equal gfn_generate_rays on the sampled pixels, colours equal the stack's."""
    import torch

    from gfnerf_amd import data

    g = golden
    n_img, H, W, R = 6, 48, 64, 512
    gen = torch.Generator(device=dev).manual_seed(3)
    images = torch.rand((n_img, H, W, 3), device=dev, generator=gen)
    c2w, intr = _t(g["rays_c2w"], dev), _t(g["rays_intr"], dev)
    n_cams = c2w.shape[0]
    image_idx = torch.arange(n_img, dtype=torch.int32, device=dev) % n_cams
    em = data.ErrorMap(torch.rand((n_img, H, W, 1), device=dev, generator=gen))
    batch = dict(image=images, image_idx=image_idx, rel_camera_idx=image_idx.clone(), error_map=em)
    rg = data.RayGenerator(c2w, intr)
    for sampler in (data.PixelSampler(R, generator=gen), data.ErrorPixelSampler(R, generator=gen)):
        rays, b, ray_indices = data.next_train(batch, sampler, rg)
        idx = ray_indices.cpu().numpy()
        assert idx.shape == (R, 3) and b["image"].shape == (R, 3) and rays["origins"].shape == (R, 3)
        slot = b["slot"].cpu().numpy()
        assert np.array_equal(idx[:, 0], image_idx.cpu().numpy()[slot])
        assert np.array_equal(b["image"].cpu().numpy(), images.cpu().numpy()[slot, idx[:, 1], idx[:, 2]])
        assert np.array_equal(b["error_map"].cpu().numpy(), em.map.cpu().numpy()[slot, idx[:, 1], idx[:, 2]])
        want = ops.generate_rays(c2w, intr, cam_idx=ray_indices[:, 0].contiguous(), coords=(ray_indices[:, 1:3].float() + 0.5).contiguous())
        assert torch.equal(rays["directions"], want["directions"]) and torch.equal(rays["origins"], want["origins"])
        assert torch.equal(rays["rel_camera_indices"], image_idx[b["slot"].long()])
        if isinstance(sampler, data.ErrorPixelSampler):
            assert sampler.last_status.tolist() == [0, 0]
            lin = b["lin"].cpu().numpy()
            assert len(set(lin[: int(R * 0.2)])) == int(R * 0.2) and len(set(lin[int(R * 0.2):])) == R - int(R * 0.2)
    # the pipeline's write-back: positions in the stack are the first column the caller passes (reference quirk kept)
    pred = torch.rand((R, 3), device=dev, generator=gen)
    stack_idx = torch.cat([b["slot"][:, None], ray_indices[:, 1:3]], 1).contiguous()
    before = em.map.clone()
    em.update(stack_idx, pred, b["image"])
    e = (b["image"] - pred).abs().sum(-1)
    lin = b["lin"]
    _, inv, cnt = torch.unique(lin, return_inverse=True, return_counts=True)
    once = cnt[inv] == 1  # a pixel both groups drew holds one of its two errors
    assert once.sum() >= R - 16 and torch.allclose(em.map.view(-1)[lin][once], e[once], rtol=1e-6, atol=1e-7)
    untouched = torch.ones(em.map.numel(), dtype=torch.bool, device=dev); untouched[lin] = False
    assert torch.equal(em.map.view(-1)[untouched], before.view(-1)[untouched])
    np.testing.assert_allclose(em.block_sums.cpu().numpy(), ops.errmap_block_sums(em.map).cpu().numpy(), rtol=1e-12)

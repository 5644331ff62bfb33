"""End-to-end GPU parity: the whole hot path (sampler -> hash -> MLPs -> compositing, and the
training backward) on a small synthetic scene against the CPU oracle pipeline, at the
north-star tolerance (rgb / sigma within 1e-3 abs; integer outputs bit-exact)."""
import numpy as np
import pytest

from helpers import rays_for, small_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import sys, os
    import torch

    import gfnerf_amd
    from gfnerf_amd import _lib, hotpath, scene

    _lib.require_device()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pipeline as opipe

    sc = small_scene()
    cfg = hotpath.HotPathConfig(log2_T=14, early_T=0.0, grad_f16=False, keep_features=True)
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
    rng = np.random.default_rng(11)
    hp.blocks[0].copy_(torch.from_numpy(rng.uniform(-1e-2, 1e-2, tuple(hp.blocks[0].shape)).astype(np.float32)))
    # larger weights than the default init so sigma spans ~an order of magnitude and rays saturate
    hp.table.mul_(100.0)
    hp.layers[1][0].mul_(4.0)
    hp.refresh_shadows()
    R = 384
    o, d, cam = rays_for(sc, R, seed=3)
    noise = scene.make_noise(R, True, 5.0, seed=2)
    gt = rng.random((R, 3)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    return dict(hp=hp, P=hp.export_params(), opipe=opipe, o=o, d=d, cam=cam, noise=noise, gt=gt, t=t, cfg=cfg, sc=sc)


@pytest.mark.parametrize("block", [None, 0])
def test_render_matches_oracle(setup, block):
    S = setup
    hp, t = S["hp"], S["t"]
    out = hp.render(t(S["o"]), t(S["d"]), t(S["cam"]), t(S["noise"]), block=block)
    ref = S["opipe"].render(S["P"], S["o"], S["d"], S["cam"], S["noise"], S["cfg"].log2_T, block=block, early_T=0.0,
                            scale_factor=S["cfg"].scale_factor)
    m = ref["samples"]
    N = int(m["pts_offsets"][-1])
    assert N > 20000
    s = out["samples"]
    assert np.array_equal(s["pts_offsets"].cpu().numpy().astype(np.int64), m["pts_offsets"])
    assert np.array_equal(s["anchors"].cpu().numpy()[:N].astype(np.int64), m["anchors"])
    assert np.array_equal(s["warped_pts"].cpu().numpy()[:N], m["x"])  # table-space points, bit-exact
    f = out["field"]
    assert np.array_equal(f["feat"].cpu().numpy()[:N], ref["field"]["feat"])  # hash features bit-exact
    sig, sig_ref = f["sigma"].cpu().numpy()[:N], ref["field"]["sigma"]
    assert sig_ref.max() / sig_ref.min() > 5 and ref["acc"].max() > 0.999
    np.testing.assert_allclose(sig, sig_ref, rtol=1e-4, atol=1e-3)          # sigma within 1e-3 abs
    np.testing.assert_allclose(f["rgb"].cpu().numpy()[:N], ref["field"]["rgb"], atol=1e-4)
    np.testing.assert_allclose(out["rgb"].cpu().numpy(), ref["rgb"], atol=1e-3)  # north-star tolerance
    assert np.abs(out["rgb"].cpu().numpy() - ref["rgb"]).max() < 1e-4          # what we actually achieve
    np.testing.assert_allclose(out["accumulation"].cpu().numpy()[:, 0], ref["acc"], atol=1e-4)
    np.testing.assert_allclose(out["depth"].cpu().numpy()[:, 0], ref["depth"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(out["oct_depth"].cpu().numpy()[:, 0], ref["oct_depth"])
    assert out["oct_depth"].min() > 0  # nerfacto.py:556


@pytest.mark.parametrize("precision", ["x3", "f32"])
def test_fused_decoder_matches_per_layer_kernels(setup, precision):
    """gfn_field_fwd[_x3] / gfn_field_bwd_dx[_x3] (one kernel per direction; split-f16 and exact-f32 MFMA forms) against
    the per-layer f32 kernels."""
    import torch

    from gfnerf_amd import ops

    S = setup
    hp, t = S["hp"], S["t"]
    hp.cfg.mlp_precision = precision
    s = hp.sample(t(S["o"]), t(S["d"]), t(S["noise"]))
    cam = t(S["cam"])
    N = int(s["pts_offsets"][-1].item())
    fu = hp._field_forward(s, cam, block=0, fused=True, save=True)
    un = hp._field_forward(s, cam, block=0, fused=False)
    np.testing.assert_allclose(fu["sigma"].cpu().numpy()[:N], un["sigma"].cpu().numpy()[:N], rtol=2e-5)
    np.testing.assert_allclose(fu["rgb"].cpu().numpy()[:N], un["rgb"].cpu().numpy()[:N], atol=2e-6)
    np.testing.assert_allclose(fu["h0"].cpu().numpy()[:N], un["h"].cpu().numpy()[:N, 0], rtol=2e-5, atol=2e-6)
    # ReLU masks reproduce the sign pattern of the stored activations (bit r + 16 t of word (layer, h))
    masks = fu["masks"].cpu().numpy()[:N].view(np.uint64).reshape(N, 3, 2)
    for li, key in enumerate(("a1", "c1", "c2")):
        act = un[key].cpu().numpy()[:N]
        bits = np.zeros((N, 128), bool)
        for h in range(2):
            for tt in range(4):
                for r in range(16):
                    bits[:, 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h] = (masks[:, li, h] >> np.uint64(tt * 16 + r)) & np.uint64(1)
        agree = (bits == (act > 0)).mean()
        assert agree > 0.9999, (key, agree)  # an activation within an ulp of zero may flip
    # backward: random upstream gradients, some rows exactly zero (dead samples)
    rng = np.random.default_rng(1)
    cap = s["ts"].shape[0]
    d_sigma = np.zeros(cap, np.float32); d_rgb = np.zeros((cap, 3), np.float32)
    d_sigma[:N] = rng.standard_normal(N) * 1e-2
    d_rgb[:N] = rng.standard_normal((N, 3)) * 1e-1
    dead = rng.random(N) < 0.3
    dead[: 64] = True
    d_sigma[:N][dead] = 0; d_rgb[:N][dead] = 0
    dfeat = ops.field_bwd_dx(t(d_sigma), t(d_rgb), fu["rgb"], fu["h0"], fu["masks"], hp.layers, n_dev=s["n_dev"],
                             precision=precision).cpu().numpy()[:N]
    hp.cfg.mlp_precision = "x3"
    (bW0, _), (bW1, _), (hW0, _), (hW1, _), (hW2, _) = hp.layers
    n_dev = s["n_dev"]
    dc2, _, _ = ops.linear_bwd(un["c2"], hW2, un["rgb"], t(d_rgb), "sigmoid", need_dw=False, n_dev=n_dev)
    dc1, _, _ = ops.linear_bwd(un["c1"], hW1, un["c2"], dc2, "relu", need_dw=False, n_dev=n_dev)
    dxh, _, _ = ops.linear_bwd(un["xh"], hW0, un["c1"], dc1, "relu", need_dw=False, n_dev=n_dev, in_dim=63)
    dh = ops.head_unpack_bwd(un["h"], t(d_sigma), dxh, cam, s["pts_offsets"], n_dev=n_dev)
    da1, _, _ = ops.linear_bwd(un["a1"], bW1, un["h"], dh, "none", need_dw=False, n_dev=n_dev)
    want, _, _ = ops.linear_bwd(un["feat"], bW0, un["a1"], da1, "relu", need_dw=False, n_dev=n_dev)
    want = want.cpu().numpy()[:N]
    assert np.all(dfeat[dead] == 0)
    # an activation within rounding of zero can sit on opposite sides of the ReLU in the two paths: allow a few flips
    bad = ~np.isclose(dfeat, want, rtol=1e-3, atol=1e-5 * np.abs(want).max())
    assert bad.mean() < 1e-4, bad.mean()


def test_early_termination_within_tolerance(setup):
    S = setup
    hp, t = S["hp"], S["t"]
    full = hp.render(t(S["o"]), t(S["d"]), t(S["cam"]), t(S["noise"]), early_T=0.0)["rgb"].cpu().numpy()
    early = hp.render(t(S["o"]), t(S["d"]), t(S["cam"]), t(S["noise"]), early_T=1e-4)["rgb"].cpu().numpy()
    assert np.abs(full - early).max() < 1e-3


@pytest.mark.parametrize("stage", ["global", "focal"])
def test_training_gradients_match_oracle(setup, stage):
    import torch

    S = setup
    hp, t = S["hp"], S["t"]
    block = 0 if stage == "focal" else None
    # Charbonnier + 1.0 * S3IM (gfnerf/config.py:96, nerfacto.py:636,687-688) with an explicit permutation index
    R = S["o"].shape[0]
    rng = np.random.default_rng(17)
    s3 = np.concatenate([np.arange(R)] + [rng.permutation(R) for _ in range(9)]).astype(np.int32)
    loss, aux = hp.forward_backward(t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"]), stage=stage, block=block,
                                    s3im_index=t(s3))
    ref_loss, g, ref_aux = S["opipe"].loss_and_grads(S["P"], S["o"], S["d"], S["cam"], S["gt"], S["noise"], S["cfg"].log2_T,
                                                      block=block, s3im_index=s3, s3im_mult=hp.cfg.s3im_loss_mult)
    assert hp.cfg.s3im_loss_mult == 1.0
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=1e-5)
    tg, tg_ref = hp.table_grad.cpu().numpy(), g["table"]
    assert (tg_ref != 0).sum() > 1000
    scale = np.abs(tg_ref).max()
    np.testing.assert_allclose(tg, tg_ref, rtol=2e-2, atol=2e-3 * scale)  # f16-quantised contributions (x128 -> half)
    assert np.abs(tg - tg_ref).max() < 5e-3 * scale
    if stage == "global":
        # decoder gradients with the weight gradient's operands as the product keeps them ("rne": one f16 plane per operand, rounded to
        # nearest: 2^-12 per term, little to average over in this small batch -- the full-size bar of 1e-4 is held by
        # tests/test_gpu_bench_workload.py) and as hi + lo' planes ("x3": the arithmetic itself)
        names = ["bW0", "bb0", "bW1", "bb1", "hW0", "hb0", "hW1", "hb1", "hW2", "hb2"]
        assert hp.cfg.wgrad_planes == "rne"
        try:
            for mode, tol in (("rne", 5e-4), ("x3", 1e-4)):
                if mode != "rne":
                    hp.cfg.wgrad_planes, hp._plane_bufs = mode, None
                    hp.forward_backward(t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"]), stage=stage, block=block, s3im_index=t(s3))
                for i, (gW, gB) in enumerate(hp.layer_grads):
                    for got, want in ((gW, g[names[2 * i]]), (gB, g[names[2 * i + 1]])):
                        want = np.asarray(want)
                        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-3, atol=tol * np.abs(want).max() + 1e-9, err_msg=mode)
                np.testing.assert_allclose(hp.emb_grad.cpu().numpy(), g["emb"], rtol=1e-3, atol=1e-4 * np.abs(g["emb"]).max())
        finally:
            hp.cfg.wgrad_planes, hp._plane_bufs = "rne", None
    else:
        # the focal stage freezes decoder and embedding (nerfacto_field.py:460-461,528,549): its backward writes neither gradient
        hp.mlp_grad.fill_(7.0); hp.emb_grad.fill_(7.0)
        hp.forward_backward(t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"]), stage="focal", block=0, s3im_index=None)
        assert hp.mlp_grad.min().item() == 7.0 and hp.mlp_grad.max().item() == 7.0
        assert hp.emb_grad.min().item() == 7.0 and hp.emb_grad.max().item() == 7.0


def test_training_f16_gradient_pool(setup):
    """Default training mode: table gradient accumulated in packed f16 (x128 scale) like the reference."""
    from gfnerf_amd import hotpath

    S = setup
    t = S["t"]
    hp32 = S["hp"]
    from gfnerf_amd import ops
    s3 = ops.s3im_index(S["o"].shape[0], 10, device="cuda")  # the same permutations in both runs
    hp32.forward_backward(t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"]), stage="global", s3im_index=s3)
    ref = hp32.table_grad.cpu().numpy()
    old = hp32.cfg.grad_f16
    try:
        hp32.cfg.grad_f16, hp32.table_grad = True, None
        hp32.forward_backward(t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"]), stage="global", s3im_index=s3)
        got = hp32.table_grad.float().cpu().numpy() / 128.0
    finally:
        hp32.cfg.grad_f16, hp32.table_grad = old, None
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() < 5e-3 * scale
    # slots nobody touched stay exactly zero; a slot whose f32 sum cancels exactly may keep an f16 ulp
    untouched = np.ones(ref.size, bool)
    untouched[: 17 * (1 << S["cfg"].log2_T)] = False
    assert np.all(got.reshape(-1)[untouched] == 0)
    assert np.abs(got[ref == 0]).max() <= 1e-4 * scale


def test_train_steps_reduce_loss(setup):
    """A few Adam steps on a fixed batch must lower the Charbonnier loss (fwd+bwd+optimiser wiring)."""
    import torch

    from gfnerf_amd import hotpath

    S = setup
    sc = S["sc"]
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=14), seed=1)
    t = S["t"]
    args = (t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"]))
    losses = []
    for it in range(12):
        loss, aux = hp.train_step(*args)
        losses.append(loss.item())
    assert losses[-1] < losses[0] * 0.9, losses
    hp.update_oct_nodes(aux)  # runs on the packed samples of the last step
    assert int((hp.visit_cnt > 0).sum().item()) > 10


@pytest.mark.parametrize("use_graph,pipeline", [(True, 1), (False, 1), (True, 3), (True, 4)])  # 4: the fourth tile re-uses the first one's sample buffer
def test_tile_renderer_matches_direct_render_and_oracle(setup, use_graph, pipeline):
    """render.py path: gfn_generate_rays + hipGraph-replayed tiles == the same rays pushed through HotPath.render in one
    go == the oracle pipeline (ragged last tile included)."""
    import torch

    from gfnerf_amd import ops
    from gfnerf_amd import render as rmod
    from gfnerf_amd import scene

    S = setup
    hp, t, sc = S["hp"], S["t"], S["sc"]
    c2w = t(sc["c2w"])
    K = sc["intri"]
    intr = t(np.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], -1).astype(np.float32))
    # a small off-centre image of camera 3 (principal point shifted so the 24 x 20 window looks at the scene centre)
    H, W = 20, 24
    intr2 = intr.clone(); intr2[:, 2] = W / 2; intr2[:, 3] = H / 2; intr2[:, 0] *= 0.02; intr2[:, 1] *= 0.02
    tr = rmod.TileRenderer(hp, c2w, intr2, tile_rays=128, block=0, fineness=5.0, early_T=0.0, use_graph=use_graph, pipeline=pipeline,
                           decoder_cus=192)  # pipeline 3: 4 tiles (one ragged) in groups of 3, samplers beside the capped fused forward
    img = tr.render_camera(3, H, W)
    assert img["rgb"].shape == (H, W, 3) and img["depth"].shape == (H, W)
    rays = ops.generate_rays(c2w, intr2, cam0=3, first_pixel=0, width=W, n=H * W)
    cam = torch.full((H * W,), 3, dtype=torch.int32, device="cuda")
    noise = torch.full((1024 + H * W + 10,), 5.0, device="cuda")
    direct = hp.render(rays["origins"], rays["directions"], cam, noise, block=0, early_T=0.0)
    # tiles see different batch-global depth clips (renderers.py:281 clips per forward chunk, like the reference) -> compare rgb / acc
    np.testing.assert_allclose(img["rgb"].reshape(-1, 3).cpu().numpy(), direct["rgb"].cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(img["accumulation"].reshape(-1).cpu().numpy(), direct["accumulation"][:, 0].cpu().numpy(), atol=1e-6)
    o_np, d_np = rays["origins"].cpu().numpy(), rays["directions"].cpu().numpy()
    ref = S["opipe"].render(S["P"], o_np, d_np, np.full(H * W, 3, np.int32), scene.make_noise(H * W, False, 5.0), S["cfg"].log2_T, block=0,
                            early_T=0.0, scale_factor=S["cfg"].scale_factor)
    assert ref["acc"].max() > 0.5
    np.testing.assert_allclose(img["rgb"].reshape(-1, 3).cpu().numpy(), ref["rgb"], atol=1e-3)
    # a second frame through the same captured graph
    img2 = tr.render_camera(5, H, W)
    rays2 = ops.generate_rays(c2w, intr2, cam0=5, first_pixel=0, width=W, n=H * W)
    cam.fill_(5)
    direct2 = hp.render(rays2["origins"], rays2["directions"], cam, noise, block=0, early_T=0.0)
    np.testing.assert_allclose(img2["rgb"].reshape(-1, 3).cpu().numpy(), direct2["rgb"].cpu().numpy(), atol=1e-6)
    assert rmod.frame_shard(10, 4, 1) == [1, 5, 9]


@pytest.mark.parametrize("block", [None, 0])
def test_fused_forward_matches_separate_kernels(setup, block):
    """gfn_fused_forward (lookup + decoder + compositing in one kernel) against the three separate kernels on the same
    samples: every sample (early_T = 0), then with early termination (skipped samples zeroed, colour within 1e-3)."""
    import torch

    from gfnerf_amd import ops

    S = setup
    hp, t = S["hp"], S["t"]
    cam = t(S["cam"])
    s = hp.sample(t(S["o"]), t(S["d"]), t(S["noise"]))
    N = int(s["pts_offsets"][-1].item())
    assert hp._fused_ok(block)
    f, comp = hp._forward_fused(s, cam, block, 0.0, save=True)
    sep = hp._field_forward(s, cam, block=block, fused=True, save=True)
    csep = ops.composite_fwd(s["pts_offsets"], sep["sigma"], s["dists"], s["ts"], sep["rgb"], early_T=0.0)
    fa, fb = f["feat"].cpu().numpy()[:N], sep["feat"].cpu().numpy()[:N]
    assert np.array_equal(fa, fb), ((fa != fb).mean(), np.argwhere(fa != fb)[:5], np.isnan(fa).sum(), np.isnan(fb).sum())  # same lookups, bit for bit
    assert np.array_equal(f["sigma"].cpu().numpy()[:N], sep["sigma"].cpu().numpy()[:N])        # same decoder tile code
    assert np.array_equal(f["rgb"].cpu().numpy()[:N], sep["rgb"].cpu().numpy()[:N])
    assert np.array_equal(f["h0"].cpu().numpy()[:N], sep["h0"].cpu().numpy()[:N])
    assert np.array_equal(f["masks"].cpu().numpy()[:N], sep["masks"].cpu().numpy()[:N])
    np.testing.assert_allclose(comp["weights"].cpu().numpy()[:N], csep["weights"].cpu().numpy()[:N], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(comp["alphas"].cpu().numpy()[:N], csep["alphas"].cpu().numpy()[:N], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(comp["rgb"].cpu().numpy(), csep["rgb"].cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(comp["acc"].cpu().numpy(), csep["acc"].cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(comp["depth_raw"].cpu().numpy(), csep["depth_raw"].cpu().numpy(), rtol=2e-5, atol=1e-5)
    assert np.array_equal(comp["t_minmax"].cpu().numpy(), csep["t_minmax"].cpu().numpy())
    # early termination: rays leave after the 32-sample tile in which T drops below 1e-4 (denser medium: h0 bias + 3)
    hp.layers[1][1][0] += 3.0
    hp._x3_stale = True
    try:
        sep = hp._field_forward(s, cam, block=block, fused=True, save=True)
        csep = ops.composite_fwd(s["pts_offsets"], sep["sigma"], s["dists"], s["ts"], sep["rgb"], early_T=0.0)
        fe, ce = hp._forward_fused(s, cam, block, 1e-4, save=True)
    finally:
        hp.layers[1][1][0] -= 3.0
        hp._x3_stale = True
    sig = fe["sigma"].cpu().numpy()[:N]
    skipped = sig == 0
    assert 0.05 < skipped.mean() < 0.9, skipped.mean()
    assert np.all(fe["rgb"].cpu().numpy()[:N][skipped] == 0) and np.all(ce["weights"].cpu().numpy()[:N][skipped] == 0)
    assert np.array_equal(sig[~skipped], sep["sigma"].cpu().numpy()[:N][~skipped])
    po = s["pts_offsets"].cpu().numpy()
    for r in range(0, po.size - 1, 13):  # skipped samples form a suffix of their ray, starting on a 32-sample tile boundary
        k = skipped[po[r]:po[r + 1]]
        if k.any():
            first = int(np.argmax(k))
            assert k[first:].all() and first % 32 == 0
    assert np.abs(ce["rgb"].cpu().numpy() - csep["rgb"].cpu().numpy()).max() < 1e-3
    assert np.array_equal(ce["t_minmax"].cpu().numpy(), csep["t_minmax"].cpu().numpy())


def test_fused_paths_edge_cases(setup):
    """Rays that miss the scene (no samples), rays that hit the 1024-sample cap, an all-miss batch and an empty batch
    through the fused forward / backward; every output stays finite and rays without samples render black."""
    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    hp, t, sc = S["hp"], S["t"], S["sc"]
    R = 96
    o = S["o"][:R].copy(); d = S["d"][:R].copy(); cam = S["cam"][:R].copy()
    o[::3] = np.array([0.0, 0.0, 4000.0], np.float32)          # far above the octree's bounding box ...
    d[::3] = np.array([0.0, 0.0, 1.0], np.float32)             # ... looking away: no leaf is ever hit
    from gfnerf_amd import scene
    noise = t(scene.make_noise(R, True, 0.25, seed=5))          # tiny steps: the rays that do hit run into the 1024 cap
    out = hp.render(t(o), t(d), t(cam), noise, block=0, early_T=0.0)
    po = out["samples"]["pts_offsets"].cpu().numpy()
    cnt = np.diff(po)
    assert np.all(cnt[::3] == 0) and cnt.max() == 1024
    rgb = out["rgb"].cpu().numpy()
    assert np.isfinite(rgb).all() and np.all(rgb[::3] == 0) and np.all(out["accumulation"].cpu().numpy()[::3] == 0)
    ref = S["opipe"].render(S["P"], o, d, cam, noise.cpu().numpy(), S["cfg"].log2_T, block=0, early_T=0.0, scale_factor=S["cfg"].scale_factor)
    assert np.array_equal(po.astype(np.int64), ref["samples"]["pts_offsets"])
    np.testing.assert_allclose(rgb, ref["rgb"], atol=1e-3)
    # the same rays through the patch kernel (gfn_fused_render): patches mix rays without samples, short rays and rays at the cap
    for tw in (0, 24):  # 96 rays: three runs of 32, or a 24-wide block of 4 rows = three 8 x 4 patches
        pr = hp.render(t(o), t(d), t(cam), noise, block=0, early_T=0.0, samples=out["samples"], tile_width=tw)
        assert pr["field"] is None
        np.testing.assert_allclose(pr["rgb"].cpu().numpy(), rgb, atol=2e-6)
        np.testing.assert_allclose(pr["accumulation"].cpu().numpy(), out["accumulation"].cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(pr["depth"].cpu().numpy(), out["depth"].cpu().numpy(), rtol=1e-5, atol=1e-6)
        # ... and from the march's records (no samples given: render samples in the record form itself): the same bits
        pr2 = hp.render(t(o), t(d), t(cam), noise, block=0, early_T=0.0, tile_width=tw)
        assert pr2["samples"].get("pts_offsets") is None and np.array_equal(pr2["samples"]["counts"].cpu().numpy(), cnt)
        for k in ("rgb", "accumulation", "depth", "oct_depth"):
            assert torch.equal(pr2[k], pr[k]), (tw, k)
    # training step on the same batch (fused backward): finite loss and gradients
    gt = t(S["gt"][:R])
    loss, aux = hp.forward_backward(t(o), t(d), t(cam), gt, noise, stage="focal", block=0)
    assert np.isfinite(loss.item()) and torch.isfinite(hp.table_grad.float()).all()
    # every ray misses
    o2 = np.tile(np.array([[0.0, 0.0, 4000.0]], np.float32), (R, 1)); d2 = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (R, 1))
    out2 = hp.render(t(o2), t(d2), t(cam), noise, block=None)
    assert int(out2["samples"]["pts_offsets"][-1].item()) == 0 and float(out2["rgb"].abs().max()) == 0.0
    out2p = hp.render(t(o2), t(d2), t(cam), noise, block=None, samples=out2["samples"], tile_width=0)
    assert out2p["field"] is None and float(out2p["rgb"].abs().max()) == 0.0 and float(out2p["accumulation"].abs().max()) == 0.0
    assert torch.equal(out2p["depth"], out2["depth"])
    out2r = hp.render(t(o2), t(d2), t(cam), noise, block=None, tile_width=0)  # the record form on rays without samples
    assert int(out2r["samples"]["counts"].sum().item()) == 0 and float(out2r["rgb"].abs().max()) == 0.0 and torch.equal(out2r["depth"], out2["depth"])
    loss2, _ = hp.forward_backward(t(o2), t(d2), t(cam), gt, noise, stage="global")
    assert np.isfinite(loss2.item()) and float(hp.table_grad.float().abs().max()) == 0.0
    # empty batch
    e3 = torch.zeros((0, 3), device="cuda"); c3 = torch.zeros(0, dtype=torch.int32, device="cuda")
    out3 = hp.render(e3, e3, c3, noise, block=0)
    assert out3["rgb"].shape == (0, 3)


def test_training_trajectory_matches_oracle_psnr(setup):
    """North-star gate 'PSNR within 0.1 dB of the reference path': the same 12 Adam steps (global stage: table, both MLPs
    and the embedding train; Charbonnier + S3IM; f16 x128 gradient pool like the reference) on the HIP path and on the
    CPU oracle from identical parameters, noise and S3IM permutations."""
    import torch

    from gfnerf_amd import hotpath

    S = setup
    sc, t, opipe = S["sc"], S["t"], S["opipe"]
    import oracle as O

    cfg = hotpath.HotPathConfig(log2_T=12, early_T=0.0)
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=5)
    hp.table.mul_(30.0)
    hp.refresh_shadows()
    P = hp.export_params()
    R = 128
    o, d, cam, gt = S["o"][:R], S["d"][:R], S["cam"][:R], S["gt"][:R]
    noise = S["noise"][: 1024 + R + 10].copy()
    rng = np.random.default_rng(23)
    # oracle state
    names = ["bW0", "bb0", "bW1", "bb1", "hW0", "hb0", "hW1", "hb1", "hW2", "hb2"]
    params = {"table": P["table"].copy(), "emb": P["emb"].copy()}
    for i, (W, b) in enumerate(P["layers"]):
        params[names[2 * i]], params[names[2 * i + 1]] = W.copy(), b.copy()
    mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in params.items()}
    psnr_gpu, psnr_cpu, loss_gpu, loss_cpu = [], [], [], []
    n_steps = 12
    for step in range(1, n_steps + 1):
        s3 = np.concatenate([np.arange(R)] + [rng.permutation(R) for _ in range(9)]).astype(np.int32)
        loss, aux = hp.train_step(t(o), t(d), t(cam), t(gt), t(noise), stage="global", s3im_index=t(s3))
        rgb = aux["comp"]["rgb"].cpu().numpy()
        psnr_gpu.append(-10 * np.log10(np.mean((rgb - gt) ** 2))); loss_gpu.append(loss.item())
        Pc = dict(P, table=params["table"], emb=params["emb"], layers=[(params[names[2 * i]], params[names[2 * i + 1]]) for i in range(5)])
        l, g, ra = opipe.loss_and_grads(Pc, o, d, cam, gt, noise, cfg.log2_T, s3im_index=s3, s3im_mult=cfg.s3im_loss_mult)
        psnr_cpu.append(-10 * np.log10(np.mean((ra["comp"]["rgb"] - gt) ** 2))); loss_cpu.append(l)
        for k in params:
            m, v = mom[k]
            O.adam_step(params[k].reshape(-1), np.ascontiguousarray(g[k], np.float32).reshape(-1), m.reshape(-1), v.reshape(-1), step,
                        lr=cfg.lr, eps=cfg.adam_eps)
    psnr_gpu, psnr_cpu = np.asarray(psnr_gpu), np.asarray(psnr_cpu)
    assert psnr_cpu[-1] > psnr_cpu[0] + 0.5, psnr_cpu                      # it does train
    assert np.abs(psnr_gpu - psnr_cpu).max() < 0.1, (psnr_gpu, psnr_cpu)    # every step within 0.1 dB
    np.testing.assert_allclose(loss_gpu, loss_cpu, rtol=5e-3)
    np.testing.assert_allclose(hp.layers[4][0].cpu().numpy(), params["hW2"], atol=3e-3)


def test_prefetched_sampler_is_the_same_sampler(setup):
    """HotPath.prefetch_sample: the next batch's sampler on a second stream beside the (CU-capped) fused backward returns
    bit-identical samples, is collected only by a call with the very same tensors, and the training losses agree."""
    import torch

    from gfnerf_amd import hotpath

    S = setup
    t, sc = S["t"], S["sc"]
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    o2, d2 = o.flip(0).contiguous(), d.flip(0).contiguous()
    from gfnerf_amd import ops
    s3 = ops.s3im_index(o.shape[0], 10, device="cuda")
    runs = []
    for pre in (False, True):
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=14), seed=1, n_blocks=1)
        hp.blocks[0].uniform_(-1e-3, 1e-3); hp.refresh_shadows()
        losses = []
        for it in range(4):
            a, b = ((o, d), (o2, d2)) if it % 2 == 0 else ((o2, d2), (o, d))
            loss, aux = hp.train_step(a[0], a[1], cam, gt, noise, stage="focal", block=0, s3im_index=s3,
                                      prefetch=(b[0], b[1], noise) if pre else None)
            assert (hp._pre is not None) == pre
            losses.append(loss.item())
        runs.append((losses, aux["samples"]))
        if pre:  # a stale prefetch is dropped, not handed to other rays
            s_other = hp.sample(o, d, noise * 1.0)
            assert hp._pre is None and s_other["pts_offsets"].data_ptr() != runs[-1][1]["pts_offsets"].data_ptr()
    (l0, s0), (l1, s1) = runs
    assert torch.equal(s0["pts_offsets"], s1["pts_offsets"])
    n = int(s0["pts_offsets"][-1])
    for k in ("anchors", "warped_pts", "dists", "ts"):
        assert torch.equal(s0[k][:n], s1[k][:n]), k
    np.testing.assert_allclose(l0, l1, rtol=2e-3)


def test_global_stage_prefetch_keeps_the_reference_order(setup):
    """Global stage: UpdateOctNodes right after the forward and the next sampler behind it on the second stream give the same
    node statistics, the same tree and the same next samples as running the three one after the other on one stream."""
    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    t, sc = S["t"], S["sc"]
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    o2, d2 = o.flip(0).contiguous(), d.flip(0).contiguous()
    s3 = ops.s3im_index(o.shape[0], 10, device="cuda")
    res = []
    for pre in (False, True):
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=14), seed=1)
        for it in range(4):
            a, b = ((o, d), (o2, d2)) if it % 2 == 0 else ((o2, d2), (o, d))
            loss, aux = hp.train_step(a[0], a[1], cam, gt, noise, stage="global", s3im_index=s3, update_oct=True,
                                      prefetch=(b[0], b[1], noise) if pre else None)
        nxt = hp.sample(o, d, noise)  # collects the prefetched sampler of the last step when pre
        torch.cuda.synchronize()
        res.append((hp.weight_stats.clone(), hp.alpha_stats.clone(), hp.visit_cnt.clone(), hp.nodes_compact.clone(), nxt))
    (w0, a0, v0, n0, s0), (w1, a1, v1, n1, s1) = res
    assert int(v0.sum()) > 0
    assert torch.equal(v0, v1) and torch.equal(n0, n1)
    # the statistics are thresholded weights of two training runs whose f16 gradient sums differ in the last bit: all but
    # a handful of nodes agree exactly
    assert (w0 != w1).float().mean() < 0.01 and (a0 != a1).float().mean() < 0.01
    assert torch.equal(s0["pts_offsets"], s1["pts_offsets"])


def test_tree_change_drops_prefetched_samples(setup):
    """A prefetched sampler (and the last step's context) marched through the OLD tree: ProcOctree / MarkInvisibleNodes
    renumber or shrink the node blob, so the next step must sample again instead of collecting them, and UpdateOctNodes must
    never be fed node indices of a tree that no longer exists (mark_visit_kernel also skips indices >= n_nodes)."""
    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    t, sc = S["t"], S["sc"]
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    s3 = ops.s3im_index(o.shape[0], 10, device="cuda")
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=14), seed=1)
    hp.table.mul_(100.0); hp.refresh_shadows()
    for _ in range(3):
        hp.train_step(o, d, cam, gt, noise, stage="global", s3im_index=s3, update_oct=True, prefetch=(o, d, noise))
    assert hp._pre is not None and hp._last_train_ctx is not None
    n_before = hp.n_nodes
    # invalidate every leaf the statistics have not confirmed, then compact: the tree shrinks and is renumbered
    hp.weight_stats.fill_(-5); hp.alpha_stats.fill_(-5)
    keep = hp.visit_cnt > 0
    hp.weight_stats[keep] = 1000; hp.alpha_stats[keep] = 1000
    ops.update_nodes(hp._last_train_ctx["samples"]["pts_offsets"], hp._last_train_ctx["samples"]["anchors"], hp._last_train_ctx["comp"]["weights"],
                     hp._last_train_ctx["comp"]["alphas"], hp.weight_stats, hp.alpha_stats, hp.visit_cnt, hp.nodes)
    hp.proc_octree(compact=True)
    assert hp.n_nodes < n_before and hp._pre is None and hp._last_train_ctx is None
    fresh = hp.sample(o, d, noise)
    torch.cuda.synchronize()
    assert int(fresh["anchors"][: int(fresh["pts_offsets"][-1]), 1].max()) < hp.n_nodes  # node indices of the NEW tree
    loss, aux = hp.train_step(o, d, cam, gt, noise, stage="global", s3im_index=s3, update_oct=True, prefetch=(o, d, noise))
    torch.cuda.synchronize()
    assert torch.equal(aux["samples"]["pts_offsets"], fresh["pts_offsets"]) and bool(torch.isfinite(loss).all())
    assert hp.weight_stats.numel() == hp.n_nodes and int(hp.visit_cnt.max()) > 0
    # stale indices handed to K5 anyway are skipped, not written past the arrays
    stale = aux["samples"]["anchors"].clone()
    stale[:, 1] += hp.n_nodes
    guard = torch.full((3 * hp.n_nodes + 64,), -7, dtype=torch.int64, device="cuda")
    w, a, v = guard[: hp.n_nodes], guard[hp.n_nodes: 2 * hp.n_nodes], guard[2 * hp.n_nodes: 3 * hp.n_nodes]
    w.fill_(1000); a.fill_(1000); v.fill_(0)
    ops.update_nodes(aux["samples"]["pts_offsets"], stale, aux["comp"]["weights"], aux["comp"]["alphas"], w, a, v, hp.nodes.clone())
    torch.cuda.synchronize()
    assert int(v.max()) == 0 and bool((guard[3 * hp.n_nodes:] == -7).all())


def test_optimizer_groups_follow_the_reference(setup):
    """Every parameter group is its own Adam (bias correction counts the GROUP's steps); delete_optimizer / update_optimizer
    drop a group's moments and step count like Optimizers.delete_optimizer + the update_optimizer callback."""
    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    t, sc = S["t"], S["sc"]
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    s3 = ops.s3im_index(o.shape[0], 10, device="cuda")
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=14), seed=1, n_blocks=2)
    for _ in range(3):
        hp.train_step(o, d, cam, gt, noise, stage="global", s3im_index=s3)
    assert hp._adam_t == {"table": 3, "mlp": 3, "emb": 3}
    # first step of a block: a fresh Adam -> the update is lr_block * sign(g) wherever the gradient is not tiny (t = 1)
    hp.update_optimizer("focal", active_blocks=[0])
    assert set(hp._adam) == set()
    before = hp.blocks[0].clone()
    hp.forward_backward(o, d, cam, gt, noise, stage="focal", block=0, s3im_index=s3)
    g = hp.table_grad.float().view(-1)[: before.numel()].clone() / 128.0
    hp.optimizer_step("focal", 0)
    delta = (hp.blocks[0] - before).view(-1)
    big = g.abs() > 1e-6
    assert int(big.sum()) > 100
    assert torch.allclose(delta[big], -hp.cfg.lr_block * torch.sign(g[big]), rtol=1e-3, atol=1e-9)
    assert hp._adam_t == {"block0": 1}
    hp.train_step(o, d, cam, gt, noise, stage="focal", block=1, s3im_index=s3)
    assert hp._adam_t == {"block0": 1, "block1": 1}
    hp.update_optimizer("focal", active_blocks=[1])     # block 0 leaves the active set: its optimizer is deleted
    assert set(hp._adam) == {"block1"}
    hp.update_optimizer("global")
    assert set(hp._adam) == set()


@pytest.mark.parametrize("stage", ["focal", "global"])
def test_lds_accumulated_gradient_matches_atomics(setup, stage):
    """Backward in record mode + gfn_grad_records_reduce (sort by slot range, accumulate in LDS) gives the gradient pool of the
    atomic form up to the order of the f16 additions -- also when record regions / range streams overflow into atomics."""
    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    sc, t = S["sc"], S["t"]
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    s3 = ops.s3im_index(o.shape[0], 10, device="cuda")
    pools = []
    from gfnerf_amd import _lib
    n_ranges = _lib.lib().gfn_grad_record_ranges(14)  # T = 2^14: 8.5 T f16 pairs in ranges of 16 Ki
    assert n_ranges == 9
    for lds, capacity, exact in ((False, 1 << 22, True), (False, 1 << 22, True), (True, 1 << 22, True), (True, n_ranges * 256 * 8, True),
                                 (True, 1 << 22, False)):
        cfg = hotpath.HotPathConfig(log2_T=14, lds_accumulate=lds, record_capacity=capacity, lds_accumulate_exact=exact)
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=1, n_blocks=1)
        hp.blocks[0].uniform_(-1e-3, 1e-3); hp.refresh_shadows()
        if lds:
            hp.record_counts_seen = torch.zeros_like(hp._record_buffers()[1])
        loss, aux = hp.forward_backward(o, d, cam, gt, noise, stage=stage, block=0 if stage == "focal" else None, s3im_index=s3)
        pools.append(hp.table_grad.float().cpu().numpy().copy())
        if lds:
            assert int(hp._records[1].abs().max().item()) == 0  # the accumulate kernel consumed the counters
            counts = hp.record_counts_seen.cpu().numpy()
            assert counts.shape == (n_ranges * 256,) and counts.sum() > 0 and counts.max() <= hp._records[0].numel() // (n_ranges * 256)
    ref, again = pools[0], pools[1]
    scale = np.abs(ref).max()
    assert scale > 0
    # the f16 pool is a sum of up to hundreds of f16 terms in whatever order the hardware took them: two runs of the SAME
    # atomic kernel differ by this much, and the LDS-accumulated pools must not differ from them by more
    noise_floor = np.abs(again - ref).max()
    for got in pools[2:]:
        assert np.abs(got - ref).max() <= max(4.0 * noise_floor, 5e-2 * scale)  # (the floor is itself one random draw)
        np.testing.assert_allclose(np.abs(got).sum(), np.abs(ref).sum(), rtol=2e-2)  # (the signed sum cancels to almost nothing)
        assert ((ref != 0) != (got != 0)).mean() < 2e-3


def test_global_stage_level_major_forward_matches_fused_forward(setup):
    """The global stage switches from the fused forward to the level-major lookup + decoder + compositing kernels after
    level_major_global_after optimiser steps: same features bit for bit, so the loss and every gradient of the step agree
    (early_T = 0 here: the fused kernel's early termination is the only other difference)."""
    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    sc, t = S["sc"], S["t"]
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    s3 = ops.s3im_index(o.shape[0], 10, device="cuda")
    out = {}
    for name, kw in (("fused", dict(level_major_global_after=-1)), ("lm", dict(level_major_global_after=0))):
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=14, early_T=0.0, **kw), seed=1)
        hp.table.mul_(100.0); hp.refresh_shadows()
        loss, _ = hp.forward_backward(o, d, cam, gt, noise, stage="global", s3im_index=s3)
        assert hp.last_forward == name
        out[name] = (float(loss.item()), hp.table_grad.float().cpu().numpy(), hp.mlp_grad.cpu().numpy(), hp.emb_grad.cpu().numpy())
    (l0, tg0, mg0, eg0), (l1, tg1, mg1, eg1) = out["fused"], out["lm"]
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    for a, b in ((tg0, tg1), (mg0, mg1), (eg0, eg1)):
        scale = np.abs(a).max()
        assert scale > 0 and np.abs(a - b).max() <= 2e-3 * scale
    # and the default: fused for the first steps, level-major from step 50 on
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=14), seed=1)
    hp.forward_backward(o, d, cam, gt, noise, stage="global", s3im_index=s3)
    assert hp.last_forward == "fused"
    hp.step = 50
    hp.forward_backward(o, d, cam, gt, noise, stage="global", s3im_index=s3)
    assert hp.last_forward.startswith("lm")


def test_exact_lds_accumulation_makes_the_table_gradient_reproducible(setup):
    """With the fixed-point image the table gradient no longer depends on the order in which the hardware took the adds: two
    focal backward passes of the same batch (Charbonnier only -- S3IM scatters its ray gradients with f32 atomics) give the
    SAME bits, which neither the scattered atomics nor the f16 image do."""
    import torch

    from gfnerf_amd import hotpath

    S = setup
    sc, t = S["sc"], S["t"]
    R = 64  # few enough rays, and (T = 2^18) enough slot ranges, that no record stream fills up: a full stream falls back to f16 atomics
    o, d, cam, gt, noise = t(S["o"][:R]), t(S["d"][:R]), t(S["cam"][:R]), t(S["gt"][:R]), t(S["noise"])

    def run(**kw):
        cfg = hotpath.HotPathConfig(log2_T=18, s3im_loss_mult=0.0, **kw)
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=1, n_blocks=1)
        hp.blocks[0].uniform_(-1e-3, 1e-3, generator=torch.Generator(device="cuda").manual_seed(3)); hp.refresh_shadows()
        hp.record_counts_seen = torch.zeros_like(hp._record_buffers()[1])
        out = []
        for _ in range(2):
            hp.forward_backward(o, d, cam, gt, noise, stage="focal", block=0)
            out.append(hp.table_grad.clone())
            assert 0 < int(hp.record_counts_seen.max().item()) < hp._records[0].numel() // hp.record_counts_seen.numel()
        return out

    a, b = run(lds_accumulate=True)
    assert float(a.float().abs().max()) > 0 and torch.equal(a, b)
    c, _ = run(lds_accumulate=True)  # a fresh engine: other buffers, other stream lengths per workgroup -- same sums
    assert torch.equal(a, c)
    f1, f2 = run(lds_accumulate=True, lds_accumulate_exact=False)
    assert float((f1.float() - a.float()).abs().max()) <= 5e-2 * float(a.float().abs().max())  # the f16 image: close, not identical


def test_async_adam_is_the_same_training_run(setup):
    """HotPathConfig.async_adam: Adam + pool zeroing on a third stream beside the next step's global-table lookup gives the
    same losses and (up to the f16 gradient noise) the same block table as the synchronous order."""
    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    sc, t = S["sc"], S["t"]
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    s3 = ops.s3im_index(o.shape[0], 10, device="cuda")
    runs = []
    for flag in (False, True):
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=14, async_adam=flag),
                             seed=1, n_blocks=1)
        hp.blocks[0].uniform_(-1e-3, 1e-3); hp.refresh_shadows()
        losses = []
        for _ in range(6):
            loss, _ = hp.train_step(o, d, cam, gt, noise, stage="focal", block=0, s3im_index=s3)
            losses.append(loss)
        assert hp._opt_pending == flag
        P = hp.export_params()  # synchronises with the optimiser stream
        assert not hp._opt_pending
        runs.append(([float(x) for x in losses], hp.blocks[0].clone(), hp.table_grad.clone()))
    (l0, b0, g0), (l1, b1, g1) = runs
    np.testing.assert_allclose(l0, l1, rtol=2e-3)
    assert l0[-1] < l0[0]
    # Adam normalises every gradient to about +-lr_block per step, so an entry whose tiny gradient comes out with the other
    # sign (f16 accumulation order) walks the other way: a few lr_block apart at most, almost all entries much closer
    assert float((b0 - b1).abs().max()) <= 6 * 5e-3 * 2 * 1.2
    assert float((b0 - b1).abs().mean()) < 3e-3
    assert float(g1.abs().max()) == 0.0  # the pool was zeroed behind the last asynchronous Adam


@pytest.mark.parametrize("block", [None, 0])
def test_fused_forward_features_repeatably_bit_exact(setup, block):
    """The inference forms of the fused forward run two waves per SIMD.  A build that chose the table mode with a run-time branch
    inside the unrolled level loop returned wrong values for whole 16-lane groups of the upper half-wave in ~0.3 % of the rows,
    differently on every launch (the table mode is a template parameter since): the kept features must equal the stand-alone
    lookup bit for bit, five launches in a row."""
    from gfnerf_amd import ops

    S = setup
    hp, t = S["hp"], S["t"]
    o, d, cam, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["noise"])
    for rep in range(5):
        out = hp.render(o, d, cam, noise, block=block)
        s = out["samples"]
        N = int(s["pts_offsets"][-1])
        ref = ops.hash_fwd_f16(hp.table16, hp.prim, hp.bias_arg, s["warped_pts"], s["anchors"], hp.n_volumes, S["cfg"].log2_T,
                               table2_f16=hp.blocks16[0] if block is not None else None,
                               prim_pool2=hp.block_prims[0] if block is not None else None, n_dev=s["n_dev"])[:N]
        bad = out["field"]["feat"][:N] != ref
        assert int(bad.sum()) == 0, (rep, int(bad.sum()), bad.sum(0).tolist())


@pytest.mark.parametrize("block,early_T,tile_width", [(0, 0.0, 32), (0, 1e-4, 32), (None, 1e-4, 0), (0, 1e-4, 96)])
def test_patch_render_matches_ray_per_wave_render(setup, block, early_T, tile_width):
    """gfn_fused_render (a wave walks a patch of 32 neighbouring rays in lockstep, each ray compositing in its own lane) against the
    wave-per-ray fused forward on the same samples: same lookups, same decoder, the compositing sums in a different order
    (sequential per ray instead of a 64-lane scan) -> rgb / accumulation / depth to f32 rounding.  Rays of very different
    lengths share patches (the batch's rays are random, not neighbours): every drop-out path is taken."""
    import torch

    from gfnerf_amd import hotpath

    S = setup
    t, sc = S["t"], S["sc"]
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=S["cfg"], seed=1, n_blocks=1)
    hp.table.mul_(30.0); hp.blocks[0].normal_(0, 0.3); hp.refresh_shadows()
    R = (S["o"].shape[0] // 64) * 64
    o, d, cam = t(S["o"][:R]), t(S["d"][:R]), t(S["cam"][:R])
    noise = t(S["noise"])
    ref = hp.render(o, d, cam, noise, block=block, early_T=early_T)
    got = hp.render(o, d, cam, noise, block=block, early_T=early_T, samples=ref["samples"], tile_width=tile_width)
    assert got["field"] is None and hp.timings is not None
    for k, tol in (("rgb", 2e-5), ("accumulation", 2e-5), ("depth", 2e-4)):
        a, b = got[k].cpu().numpy(), ref[k].cpu().numpy()
        assert np.isfinite(a).all()
        np.testing.assert_allclose(a, b, rtol=tol, atol=tol, err_msg=k)
    assert float(ref["accumulation"].max()) > 0.5  # the comparison is not between empty images
    # the samples read where the counting march recorded them (sample(record=True) -> gfn_fused_render_rec: no scan, no fill, no packed
    # copy) are the same samples: the same images, bit for bit
    recd = hp.sample(o, d, noise, record=True)
    assert recd.get("pts_offsets") is None and torch.equal(recd["counts"], ref["samples"]["pts_offsets"].diff())
    assert torch.equal(recd["first_oct_dis"], ref["samples"]["first_oct_dis"])
    again = hp.render(o, d, cam, noise, block=block, early_T=early_T, samples=recd, tile_width=tile_width)
    for k in ("rgb", "accumulation", "depth", "oct_depth"):
        assert torch.equal(again[k], got[k]), k
    with pytest.raises(Exception, match="record"):
        hp.render(o, d, cam, noise, block=block, early_T=early_T, samples=recd)  # only the patch kernel reads records
    # a ray count that does not divide into patches falls back to the wave-per-ray kernel
    odd = hp.render(o[:40], d[:40], cam[:40], noise, block=block, early_T=early_T, tile_width=0)
    assert odd["field"] is not None


def test_patch_render_with_the_embedding_folded_into_the_bias(setup):
    """gfn_fused_render(cam_uniform): an image has ONE appearance embedding (test mode: the nearest training image's, perssampler.py:431-433), so head layer
    1's appearance part -- W_h0[:, 31:63] emb, the same 128 numbers for every sample -- is added to the layer's bias once per workgroup
    (fold_embedding_into_bias) and the decoder runs 192 instead of 216 MFMAs per 32 samples.  Same products, another summation order: the image with the
    fold equals the image without it to f32 rounding, for a camera with a LARGE embedding (so that a dropped or doubled appearance term would show), on
    packed samples and on the march's records; and the camera is read from the device (cam_idx[0]) by every launch: another camera, another image, no
    recompilation or host value involved."""
    import torch

    from gfnerf_amd import hotpath

    S = setup
    t, sc = S["t"], S["sc"]
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=S["cfg"], seed=1, n_blocks=1)
    hp.table.mul_(30.0); hp.blocks[0].normal_(0, 0.3)
    hp.emb.normal_(0, 1.0, generator=torch.Generator(device="cuda").manual_seed(3))  # (an appearance term of the size of the other inputs)
    hp.refresh_shadows(); hp._x3_stale = True
    R = (S["o"].shape[0] // 64) * 64
    o, d, noise = t(S["o"][:R]), t(S["d"][:R]), t(S["noise"])
    cam5 = torch.full((R,), 5, dtype=torch.int32, device="cuda")
    smp = hp.sample(o, d, noise)
    plain = hp.render(o, d, cam5, noise, block=0, samples=smp, tile_width=0)
    fold = hp.render(o, d, cam5, noise, block=0, samples=smp, tile_width=0, cam_uniform=True)
    for k, tol in (("rgb", 2e-6), ("accumulation", 2e-6)):
        np.testing.assert_allclose(fold[k].cpu().numpy(), plain[k].cpu().numpy(), rtol=0, atol=tol, err_msg=k)
    assert float(plain["accumulation"].max()) > 0.5
    # the appearance term matters in this setup: without it the colours move by far more than the tolerance above
    cam9 = torch.full((R,), 9, dtype=torch.int32, device="cuda")
    other = hp.render(o, d, cam9, noise, block=0, samples=smp, tile_width=0, cam_uniform=True)
    assert float((other["rgb"] - fold["rgb"]).abs().max()) > 1e-2
    np.testing.assert_allclose(other["rgb"].cpu().numpy(), hp.render(o, d, cam9, noise, block=0, samples=smp, tile_width=0)["rgb"].cpu().numpy(), rtol=0, atol=2e-6)
    # the record form (what the tile renderer launches)
    recd = hp.sample(o, d, noise, record=True)
    rec_fold = hp.render(o, d, cam5, noise, block=0, samples=recd, tile_width=0, cam_uniform=True)
    assert torch.equal(rec_fold["rgb"], fold["rgb"])


@pytest.mark.parametrize("variant", ["patch_width_0", "f32_decoder", "unfused_forward"])
def test_tile_pipeline_without_the_patch_kernel(setup, variant):
    """the pipelined TileRenderer (several tiles per replay, tiles of >= 32 Ki rays) sampled in the record form whenever the
    tile was large enough for the patch kernel -- also for configurations HotPath.render does not send there (render_patch_width = 0,
    the exact-f32 decoder, fused_forward off), which then failed on the first multi-tile frame.  Now the sample form follows
    HotPath.uses_patch_kernel: those configurations render the same image through the wave-per-ray kernel on packed samples."""
    import dataclasses

    import torch

    from gfnerf_amd import hotpath
    from gfnerf_amd import render as rmod

    S = setup
    t, sc = S["t"], S["sc"]
    c2w = t(sc["c2w"])
    K = sc["intri"]
    intr = t(np.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], -1).astype(np.float32))
    H, W = 256, 256  # two tiles of 128 x 256 = 32,768 rays in one replay
    intr2 = intr.clone(); intr2[:, 2] = W / 2; intr2[:, 3] = H / 2; intr2[:, 0] *= 0.25; intr2[:, 1] *= 0.25
    over = dict(patch_width_0=dict(render_patch_width=0), f32_decoder=dict(mlp_precision="f32"), unfused_forward=dict(fused_forward=False))[variant]
    imgs = {}
    for name, kw in (("default", {}), (variant, over)):
        cfg = dataclasses.replace(S["cfg"], render_colour_x2=False, **kw)  # (x3 throughout: the two kernels are compared to 2e-5; colour_x2 has its own test)
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=1, n_blocks=1)
        hp.table.mul_(30.0); hp.blocks[0].normal_(0, 0.3, generator=torch.Generator(device="cuda").manual_seed(5)); hp.refresh_shadows()
        tr = rmod.TileRenderer(hp, c2w, intr2, tile_rays=32768, tile_width=128, block=0, fineness=5.0, early_T=1e-4, pipeline=2)
        assert tr.T == 32768 and tr.P == 2 and tr.record == (name == "default") and (tr.patch_tw is not None) == (name == "default")
        imgs[name] = tr.render_camera(3, H, W)["rgb"]
        del tr, hp
        torch.cuda.empty_cache()
    assert float(imgs["default"].std()) > 0.003  # (not a blank image)
    tol = 1e-3 if variant == "f32_decoder" else 2e-5
    np.testing.assert_allclose(imgs[variant].cpu().numpy(), imgs["default"].cpu().numpy(), atol=tol)


def test_tile_pipeline_record_pool_is_sized_by_the_samples_and_grows_when_it_runs_out(setup):
    """the tile pipeline's ring buffers hold the march's records in pages from a pool of `rec_slots` slots per ray
    (224 by default: 7 KB per ray) instead of 1024 slots for every ray (32 KB).  The image is the dense form's bit for bit; a pool that is too
    small (rec_slots = 32 here) is noticed at the end of the frame -- the samplers' overflow words, or-ed over the frame's tiles -- and
    the frame is rendered again with a pool half as large again, until it fits.  The same for the traversal's hit lists: 128 leaves per ray
    by default instead of the reference's cap of 1024 (a ray that reaches the cap: four times the cap, and the frame again).  A frame rendered as ONE
    tile per replay (pipeline = 1) uses the same paged records and caps, and notices their overflow the same way (round 6: the memory budget counts on it)."""
    import torch

    from gfnerf_amd import hotpath
    from gfnerf_amd import render as rmod

    S = setup
    t, sc = S["t"], S["sc"]
    c2w = t(sc["c2w"])
    K = sc["intri"]
    intr = t(np.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], -1).astype(np.float32))
    H, W = 256, 384  # three tiles of 128 x 256 = 32,768 rays through a ring of two buffers
    intr2 = intr.clone(); intr2[:, 2] = W / 2; intr2[:, 3] = H / 2; intr2[:, 0] *= 0.25; intr2[:, 1] *= 0.25
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=S["cfg"], seed=1, n_blocks=1)
    hp.table.mul_(30.0); hp.blocks[0].normal_(0, 0.3, generator=torch.Generator(device="cuda").manual_seed(5)); hp.refresh_shadows()
    imgs, foot = {}, {}
    for name, slots, hits, pipe in (("dense", 0, 1024, 3), ("paged", 224, 128, 3), ("starved", 32, 2, 3), ("one_tile", 224, 128, 1), ("one_tile_starved", 32, 2, 1)):
        tr = rmod.TileRenderer(hp, c2w, intr2, tile_rays=32768, tile_width=128, block=0, fineness=5.0, early_T=1e-4, pipeline=pipe, rec_slots=slots, hit_cap=hits)
        out = tr.render_camera(3, H, W)
        imgs[name] = (out["rgb"].clone(), out["depth"].clone())
        again = tr.render_camera(3, H, W)  # a second frame through the same capture
        assert torch.equal(again["rgb"], imgs[name][0])
        foot[name] = tr.footprint_bytes()
        assert tr.record and tr.P == pipe
        if name == "dense":
            assert tr.rec_slots == 0 and tr.pool_regrown == 0
        elif name in ("paged", "one_tile"):
            assert tr.rec_slots == 224 and tr.hit_cap == 128 and tr.pool_regrown == 0
        else:
            assert tr.pool_regrown >= 1 and tr.rec_slots > 32 and tr.hit_cap > 2, (tr.pool_regrown, tr.rec_slots, tr.hit_cap)
        del tr
        torch.cuda.empty_cache()
    assert float(imgs["dense"][0].std()) > 0.003
    for name in ("paged", "starved", "one_tile", "one_tile_starved"):  # the same images bit for bit (depth: the clip range never binds)
        assert torch.equal(imgs[name][0], imgs["dense"][0]) and torch.equal(imgs[name][1], imgs["dense"][1]), name
    assert foot["paged"] < 0.45 * foot["dense"], foot


def test_step_graph_replays_without_host_sync_use_each_steps_own_adam_factors(setup):
    """StepGraph replays enqueued back to back, the host several steps ahead of the device (nothing is read back in between): Adam's
    step-dependent factors -- they travel through a ring of pinned blocks, one per replay -- must still be those of the step being
    replayed.  The same table, moments and f16 shadow, bit for bit, as the eager steps."""
    import torch

    from gfnerf_amd import hotpath, scene, stepgraph

    S = setup
    t, sc = S["t"], S["sc"]
    R, K = 256, 5
    batches = []
    for k in range(K + 2):
        o, d, cam = scene.make_rays(sc["c2w"], R, seed=140 + k)
        gt = np.random.default_rng(190 + k).random((R, 3)).astype(np.float32)
        batches.append(tuple(t(a) for a in (o, d, cam, gt)))
    fin = 5.0
    noise = t(scene.make_noise(R, False, fin))

    def run(graph):
        cfg = hotpath.HotPathConfig(log2_T=18, s3im_loss_mult=0.0)
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=1, n_blocks=1)
        hp.table.mul_(100.0); hp.refresh_shadows()
        o, d, cam, gt = batches[0]
        hp.train_step(o, d, cam, gt, noise, stage="focal", block=0)
        if graph:
            sg = stepgraph.StepGraph(hp, "focal", R, fin, block=0, train_noise=False)
            sg.prime(batches[1][0], batches[1][1])
            for k in range(1, K + 1):
                sg.step(batches[k][2], batches[k][3], batches[k + 1][0], batches[k + 1][1])
        else:
            for k in range(1, K + 1):
                o, d, cam, gt = batches[k]
                hp.train_step(o, d, cam, gt, noise, stage="focal", block=0)
        torch.cuda.synchronize()
        m, v = hp._adam["block0"][:2]
        return hp.blocks[0].clone(), m.clone(), v.clone(), hp.blocks16[0].clone()

    ref, got = run(False), run(True)
    assert float((ref[0]).abs().max()) > 0 and float(ref[1].abs().max()) > 0  # the block table did train
    for a, b, name in zip(got, ref, ("table", "m", "v", "f16 shadow")):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("stage", ["focal", "global", "global_segmented"])
def test_step_graph_is_the_same_training_run(setup, stage):
    """gfnerf_amd.stepgraph.StepGraph: K training steps as hipGraph replays (sampler of the next batch on a captured side branch,
    Adam's step-dependent factors read from the device) against the same K steps launched eagerly -- with the random draws
    switched off (eval-mode march noise, no S3IM) the two runs must agree: bit for bit in the focal stage (exact LDS
    accumulation makes the step reproducible), to f16 gradient rounding in the global stage (its weight-gradient atomics are
    not ordered).  global_segmented: the form the data-parallel global step is captured in -- four graph segments with the three
    exchange points (identities on one rank) called eagerly between them."""
    import torch

    from gfnerf_amd import hotpath, ops, scene, stepgraph

    segmented = stage == "global_segmented"
    stage = "global" if segmented else stage
    S = setup
    t, sc = S["t"], S["sc"]
    R, K = 256, 6
    batches = []
    for k in range(K + 1):
        o, d, cam = scene.make_rays(sc["c2w"], R, seed=40 + k)
        gt = np.random.default_rng(90 + k).random((R, 3)).astype(np.float32)
        batches.append(tuple(t(a) for a in (o, d, cam, gt)))
    fin = 5.0
    noise = t(scene.make_noise(R, False, fin))

    def make():
        # focal: a table large enough that no record stream overflows into (unordered) f16 atomics -> the step is reproducible
        cfg = hotpath.HotPathConfig(log2_T=18 if stage == "focal" else 14, s3im_loss_mult=0.0)
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=1, n_blocks=1 if stage == "focal" else 0)
        hp.table.mul_(100.0); hp.refresh_shadows()
        return hp

    blk = 0 if stage == "focal" else None
    upd = stage == "global"
    # eager run: K + 1 steps (step 0 is the eager step the graph needs first, too)
    he = make()
    losses_e = []
    for k in range(K + 1):
        o, d, cam, gt = batches[k]
        l, _ = he.train_step(o, d, cam, gt, noise, stage=stage, block=blk, update_oct=upd)
        losses_e.append(float(l.item()))
    # graph run: the same step 0 eagerly, then K replays
    hg = make()
    o, d, cam, gt = batches[0]
    hg.train_step(o, d, cam, gt, noise, stage=stage, block=blk, update_oct=upd)
    sg = stepgraph.StepGraph(hg, stage, R, fin, block=blk, update_oct=upd, train_noise=False, segmented=segmented or None)
    sg.prime(batches[1][0], batches[1][1])
    losses_g = []
    for k in range(1, K + 1):
        nxt = batches[k + 1] if k < K else batches[k]
        l, _ = sg.step(batches[k][2], batches[k][3], nxt[0], nxt[1])
        losses_g.append(float(l.item()))
    torch.cuda.synchronize()
    if segmented:
        kinds = ["graph" if isinstance(x, torch.cuda.CUDAGraph) else "eager" for x in sg.G[0]]
        assert kinds == ["graph", "eager", "graph", "eager", "graph", "eager", "graph"], kinds
        assert hg.seg is None
    assert hg.step == he.step == K + 1 and hg._adam_t == he._adam_t
    pe = he.blocks[0] if stage == "focal" else he.table
    pg = hg.blocks[0] if stage == "focal" else hg.table
    assert float((pe - (0 if stage == "focal" else make().table)).abs().max()) > 0  # the run did train
    if stage == "focal":
        np.testing.assert_allclose(losses_g, losses_e[1:], rtol=1e-5)  # the loss is a sum of per-wave partial sums added atomically
        assert torch.equal(pe, pg) and torch.equal(he._adam["block0"][0], hg._adam["block0"][0]) and torch.equal(he._adam["block0"][1], hg._adam["block0"][1])
    else:
        np.testing.assert_allclose(losses_g, losses_e[1:], rtol=2e-3)
        # Adam turns a gradient whose last f16 bit differs (unordered atomics where a record stream overflows at this table size)
        # into a step of the full learning rate, so a few slots differ by k * lr: all but a vanishing fraction agree
        far = (pg - pe).abs() > 2e-3 * float(pe.abs().max())
        assert float(far.float().mean()) < 1e-3 and float((pg - pe).abs().max()) <= 2 * K * 1e-2
        np.testing.assert_allclose(hg.mlp_flat.cpu().numpy(), he.mlp_flat.cpu().numpy(), atol=2e-2 * float(he.mlp_flat.abs().max()))
        assert torch.equal(hg.visit_cnt, he.visit_cnt)  # the octree statistics are integer maxima: exact
        # the gradient buffers hold the LAST step's gradient, not a sum over the replays (they are zeroed inside the captured step: csrc/gfn_common.h)
        ge, gg = he.mlp_grad.float(), hg.mlp_grad.float()
        assert float((gg - ge).abs().max()) < 0.2 * float(ge.abs().max()), (float(gg.abs().max()), float(ge.abs().max()))
        pe_, pg_ = he._grad_live().float().abs().sum(), hg._grad_live().float().abs().sum()
        assert abs(float(pg_) / float(pe_) - 1.0) < 0.2, (float(pg_), float(pe_))


@pytest.mark.parametrize("async_adam", [True, False])
def test_graphed_forward_is_the_eager_forward(setup, async_adam):
    """HotPath.forward_train_graphed (the plugin surface's focal forward over two static sets of sample buffers): three eager runs per
    set, then one hipGraph replay per call -- the same training run as forward_train, bit for bit (focal stage, T = 2^18: exact record
    accumulation): losses, the block table and its Adam moments after fourteen iterations (FOUR replays of each capture), phased forward
    included; and a decoder written from outside between two replays (a checkpoint) is repacked before the next one.
    async_adam=False is the plugin surface's arrangement (an optimiser of the caller's steps the table: gfnerf.optim.GfnAdam): the gradient pool
    is zeroed by the forward itself, i.e. INSIDE the captured graph -- as a memset node that zeroing ran in full on a capture's first replay only
    (round 5: the trained plugin run degraded and then died of it; csrc/gfn_common.h, gfn_zero_async), so every replay is checked."""
    import torch

    from gfnerf_amd import hotpath, ops, scene

    S = setup
    t, sc = S["t"], S["sc"]
    R, K = 256, 14
    noise = t(scene.make_noise(R, False, 5.0))
    batches = []
    for k in range(K):
        o, d, cam = scene.make_rays(sc["c2w"], R, seed=340 + k)
        gt = np.random.default_rng(390 + k).random((R, 3)).astype(np.float32)
        batches.append(tuple(t(a) for a in (o, d, cam, gt)))

    def run(graphed):
        cfg = hotpath.HotPathConfig(log2_T=18, s3im_loss_mult=0.0, early_T=1e-2, graph_forward=graphed, async_adam=async_adam, graph_one_phase=False)
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=1, n_blocks=1)
        hp.table.mul_(100.0); hp.refresh_shadows()
        bufs, losses, kinds = [None, None], [], []
        for k, (o, d, cam, gt) in enumerate(batches):
            if k == 6:  # parameters written from outside the optimiser: the frozen decoder and the global table
                hp.layers[1][0].mul_(1.25); hp.table.mul_(0.9); hp.refresh_shadows()
            s = hp.sample(o, d, noise, out=bufs[k % 2])
            bufs[k % 2] = s
            comp, ctx = hp.forward_train_graphed(cam, "focal", 0, s, want_depth=False)
            kinds.append(hp.last_forward_replayed)
            if not async_adam:  # the forward has zeroed the gradient pool (the last backward's gradient was in it): on every replay
                assert int((hp._grad_live() != 0).sum()) == 0, k
            loss, g = ops.charbonnier(comp["rgb"], gt)
            hp.backward(ctx, g)
            hp.optimizer_step("focal", 0)
            losses.append(float(loss.item()))
        torch.cuda.synchronize()
        return dict(losses=losses, kinds=kinds, table=hp.blocks[0].clone(), m=hp._adam["block0"][0].clone(), v=hp._adam["block0"][1].clone(),
                    k=int(hp._phase.k.item()))

    a, b = run(False), run(True)
    assert a["kinds"] == [False] * K and b["kinds"] == [False] * 6 + [True] * (K - 6), (a["kinds"], b["kinds"])  # three eager runs per buffer set first
    np.testing.assert_allclose(b["losses"], a["losses"], rtol=1e-6)  # (the loss scalar is summed with atomics)
    assert a["k"] == b["k"] and a["losses"][6] != a["losses"][5]
    for key in ("table", "m", "v"):
        assert torch.equal(a[key], b[key]), key
    assert float(b["table"].abs().max()) > 0


def test_captured_forward_takes_the_form_the_planner_asks_for(setup):
    """cfg.graph_one_phase (round 6): forward_train_graphed captures the ONE-PHASE forward (every sample, no lists: eight graph nodes instead of
    twenty-one) while the planner's K says that a split does not pay, looks at the plan again every cfg.graph_one_phase_probe forwards (one eager phased
    forward) and goes back to the phased capture when K drops.  First a planner that finds no split worth its launches (thresholds no saving can
    reach: K = 1024 -> one phase), then the global table made dense and the planner told that any saving counts (rays terminate early: the probe
    plans a split -> phased captures); every forward's colours are those of an eager forward_train on the same samples (phased and one-phase agree to
    1e-6: the same cut), and the backward of either capture leaves the eager backward's table gradient."""
    import torch

    from gfnerf_amd import hotpath, ops, scene

    S = setup
    t, sc = S["t"], S["sc"]
    R = 256
    noise = t(scene.make_noise(R, False, 5.0))
    never = dict(two_phase_min_saving=(2.0, 2.0))  # (a saving is at most 1: no split is ever worth it)
    cfg = hotpath.HotPathConfig(log2_T=18, s3im_loss_mult=0.0, early_T=1e-2, graph_one_phase_probe=6, **never)
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=1, n_blocks=1)
    ref = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0],
                          cfg=hotpath.HotPathConfig(log2_T=18, s3im_loss_mult=0.0, early_T=1e-2, graph_forward=False, **never), seed=1, n_blocks=1)
    for e in (hp, ref):
        e.blocks[0].uniform_(-1e-2, 1e-2, generator=torch.Generator(device="cuda").manual_seed(5)); e.refresh_shadows()
    bufs, forms, replayed = [None, None], [], []
    for k in range(40):
        if k == 20:  # the scene fills up: rays terminate after a few samples -- and the planner is told that any saving is worth a split
            for e in (hp, ref):
                e.table.mul_(300.0); e.refresh_shadows()
                e.cfg.two_phase_fixed_cost, e.cfg.two_phase_min_saving = 0.0, (0.01, 0.005)
        o, d, cam = (t(a) for a in scene.make_rays(sc["c2w"], R, seed=500 + k))
        gt = t(np.random.default_rng(600 + k).random((R, 3)).astype(np.float32))
        s = hp.sample(o, d, noise, out=bufs[k % 2])
        bufs[k % 2] = s
        comp, ctx = hp.forward_train_graphed(cam, "focal", 0, s, want_depth=False)
        forms.append(hp.last_forward); replayed.append(hp.last_forward_replayed)
        comp_r, ctx_r = ref.forward_train(o, d, cam, noise, "focal", 0, want_depth=False)
        np.testing.assert_allclose(comp["rgb"].cpu().numpy(), comp_r["rgb"].cpu().numpy(), atol=1e-6, err_msg=str(k))
        _, g = ops.charbonnier(comp["rgb"], gt)
        hp.backward_graphed(ctx, g)
        ref.backward(ctx_r, ops.charbonnier(comp_r["rgb"], gt)[1])
        a, b = hp.table_grad.float(), ref.table_grad.float()
        assert float((a - b).abs().max()) <= 2e-3 * max(float(b.abs().max()), 1e-12), k
    print(list(zip(forms, replayed)))
    assert forms[:6] == ["lm2"] * 6 and "lm" in forms[6:20] and replayed[19] and forms[19] == "lm"        # no split planned: captured as one phase
    assert forms[-1] == "lm2" and replayed[-1] and not hp._one_phase and int(hp._phase.k.item()) < 1024     # the dense scene: back to the phased capture
    assert {e["last_forward"] for e in hp._fwd_graphs.values() if "graph" in e} == {"lm", "lm2"}


def test_captured_forwards_are_bounded_and_evicted_least_recently_used(setup):
    """forward_train_graphed kept every capture (a full set of forward activations each) until drop_block / proc_octree.  Now the
    captures share one memory pool and at most cfg.graph_forward_keep stay (least recently used first out): three blocks x two sample-buffer
    sets with keep = 4 -- never more than 4 captured forwards alive, the oldest block's graphs gone when the third block's are captured, a
    block that comes back is captured again, and every replay still is the eager forward (outputs compared on the spot)."""
    import torch

    from gfnerf_amd import hotpath, scene

    S = setup
    t, sc = S["t"], S["sc"]
    R = 256
    noise = t(scene.make_noise(R, False, 5.0))
    cfg = hotpath.HotPathConfig(log2_T=14, s3im_loss_mult=0.0, early_T=1e-2, graph_forward_keep=4, graph_one_phase=False)
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=1, n_blocks=3)
    hp.table.mul_(100.0); hp.refresh_shadows()
    for b in hp.blocks:
        b.uniform_(-1e-2, 1e-2, generator=torch.Generator(device="cuda").manual_seed(5))
    hp.refresh_shadows()
    hp.zero_grads("focal", 0)
    bufs = [None, None]
    seen_max, k = 0, 0
    for block in (0, 1, 2, 0):
        for it in range(8):  # 3 eager runs per buffer set, then replays
            o, d, cam = (t(a) for a in scene.make_rays(sc["c2w"], R, seed=500 + k))
            s = hp.sample(o, d, noise, out=bufs[k % 2])
            bufs[k % 2] = s
            comp, ctx = hp.forward_train_graphed(cam, "focal", block, s, want_depth=False)
            replayed = hp.last_forward_replayed
            got = comp["rgb"].clone()
            want, _ = hp.forward_train(None, None, cam, None, "focal", block, want_depth=False, samples=s)
            assert float((got - want["rgb"]).abs().max()) <= 1e-6, (block, it, replayed)
            captured = [key for key, e in hp._fwd_graphs.items() if "graph" in e]
            seen_max = max(seen_max, len(captured))
            k += 1
        assert replayed                                     # every block reaches replays
        assert {key[2] for key in captured} <= {block, (block - 1) % 3 if block else 2}, [key[2] for key in captured]  # this block's + the one before
    assert seen_max == 4 and len(hp._fwd_graphs) <= 12
    hp.forget_graphs()
    assert len(hp._fwd_graphs) == 0
    torch.cuda.empty_cache()


def test_concurrent_stream_runs_beside_the_current_one():
    """ops.concurrent_stream (round 4): HIP maps streams onto a few hardware queues in creation order; a second stream that shares the launch
    stream's queue runs BEHIND it (the plugin surface's focal iteration: 1.81 instead of 1.55 ms, depending on what the process had created
    before).  Whatever number of streams exists already, the stream handed out overlaps the current one: two equal spins take the time of one."""
    import time

    import torch

    from gfnerf_amd import ops

    spin = 20_000_000  # ~8 ms
    torch.cuda.synchronize()
    t0 = time.perf_counter(); torch.cuda._sleep(spin); torch.cuda.synchronize()
    one = time.perf_counter() - t0
    hoard = []
    for k in range(6):
        hoard += [torch.cuda.Stream() for _ in range(k)]  # shift the round-robin of the queues
        s = ops.concurrent_stream()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.cuda._sleep(spin)
        with torch.cuda.stream(s):
            torch.cuda._sleep(spin)
        torch.cuda.synchronize()
        both = time.perf_counter() - t0
        assert both < 1.5 * one, (k, one, both)


def test_step_draws_are_uniform_fresh_per_call_and_replayable():
    """gfn_step_draws (the device's own Philox4x32-10 stream for a captured step's random draws): march noise (u + 0.5) * fineness with u
    uniform in [0, 1) (PersSampler_cuda.cu:380-389), S3IM's index vector = arange + 9 uniform random permutations (losses.py:780-789);
    every call -- and every hipGraph replay of one captured call -- draws fresh numbers; the same seed gives the same stream."""
    import torch

    from gfnerf_amd import ops

    R, rep, fin = 4096, 10, 3.0
    d = ops.StepDraws(R, rep, seed=1234)
    noise = torch.empty(1024 + R + 10, device="cuda")
    idx = torch.empty(rep * R, dtype=torch.int32, device="cuda")
    d.draw(noise, fin, idx)
    n1, i1 = noise.cpu().numpy().copy(), idx.cpu().numpy().reshape(rep, R).copy()
    u = n1 / fin - 0.5
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.005
    hist = np.histogram(u, bins=16, range=(0, 1))[0]
    assert hist.min() > 0.75 * u.size / 16 and hist.max() < 1.25 * u.size / 16
    assert np.array_equal(i1[0], np.arange(R))
    for row in i1[1:]:
        assert np.array_equal(np.sort(row), np.arange(R))
    assert len({row.tobytes() for row in i1}) == rep  # all different
    fixed = np.mean([np.mean(row == np.arange(R)) for row in i1[1:]])
    assert fixed < 5.0 / R * 3  # a uniform permutation has one fixed point on average
    # a position's images over the 9 permutations are spread over the whole range
    assert abs(i1[1:, :64].mean() - (R - 1) / 2) < 0.1 * R
    d.draw(noise, fin, idx)
    n2 = noise.cpu().numpy().copy()
    assert not np.array_equal(n1, n2) and abs(np.corrcoef(n1, n2)[0, 1]) < 0.05
    assert int(d.state[0].item()) == 2 and int(d.state[1].item()) == 0
    # the same seed: the same stream
    e = ops.StepDraws(R, rep, seed=1234)
    e.draw(noise, fin, idx)
    assert np.array_equal(noise.cpu().numpy(), n1) and np.array_equal(idx.cpu().numpy().reshape(rep, R), i1)
    # replays of ONE captured call move on
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e.draw(noise, fin, idx, write_arange=False)
    seen = []
    for _ in range(3):
        g.replay()
        seen.append(noise.cpu().numpy().copy())
    assert np.array_equal(seen[0], n2) and not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[1], seen[2])
    assert int(e.state[0].item()) == 4


@pytest.mark.parametrize("n_rays", [1, 37, 4096, 16384, 20011])
def test_phase_lists_against_numpy(n_rays):
    """gfn_phase_list on synthetic ray lengths -- one launch up to 16,384 rays (every workgroup sums the entries in front of its rays by
    itself), scan + fill beyond: list A (first K of every ray; the others get sigma = rgb = 0), list B (samples K.. of the rays with
    need_more), the front list (first term[r]), their counts and ray counts; K = 1024 lists nothing (count -1)."""
    import torch

    from gfnerf_amd import ops

    rng = np.random.default_rng(n_rays)
    cnt = rng.integers(0, 300, n_rays).astype(np.int32)
    cnt[rng.random(n_rays) < 0.1] = 0
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    n = int(off[-1])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    st = ops.PhaseState(n_rays, max(n, 1), "cuda")
    K = 100
    st.k.fill_(K)
    need = (rng.random(n_rays) < 0.3).astype(np.uint8)
    term = np.minimum(rng.integers(0, 320, n_rays), 400).astype(np.int32)
    st.need_more.copy_(t(need)); st.term.copy_(t(term))
    sigma, rgb = torch.ones(max(n, 1), device="cuda"), torch.ones((max(n, 1), 3), device="cuda")
    po = t(off)

    def want(c, first):
        return np.concatenate([np.arange(off[r] + first[r], off[r] + first[r] + c[r]) for r in range(n_rays)] + [np.zeros(0, np.int64)]).astype(np.int32)

    zero_r = np.zeros(n_rays, np.int32)
    cases = [("a", dict(zero=(sigma, rgb)), np.minimum(cnt, K), zero_r, 0),
             ("b", dict(phase_b=True), np.where((need > 0) & (cnt > K), cnt - K, 0), np.full(n_rays, K, np.int32), 2),
             ("front", dict(front=True), np.minimum(term, cnt), zero_r, 4)]
    for name, kw, c, first, sb in cases:
        lst, count = ops.phase_list(po, st, **kw)
        w = want(c, first)
        stats = st.stats.cpu().numpy()
        assert int(count.item()) == w.size == stats[sb] and stats[sb + 1] == int((c > 0).sum()), (name, int(count.item()), w.size)
        assert np.array_equal(lst[: w.size].cpu().numpy(), w), name
        assert np.array_equal(st.coff.cpu().numpy(), np.concatenate([[0], np.cumsum(c)]).astype(np.int32)), name
    listed = np.zeros(max(n, 1), bool)
    listed[want(np.minimum(cnt, K), zero_r)] = True
    listed[n:] = True  # (padding when the batch is empty)
    assert np.array_equal(sigma.cpu().numpy() != 0, listed) and np.array_equal((rgb.cpu().numpy() != 0).all(1), listed)
    st.k.fill_(1024)  # one phase: no list, nothing cleared
    sigma.fill_(1.0)
    _, count = ops.phase_list(po, st, zero=(sigma, rgb))
    assert int(count.item()) == -1 and bool((sigma == 1).all())


@pytest.mark.parametrize("stage", ["focal", "global"])
def test_phased_forward_is_the_one_phase_forward(setup, stage):
    """The termination-aware level-major forward (HotPathConfig.two_phase; gfn_phase_mask / gfn_composite_fwd_phased /
    gfn_two_phase_plan) against the one-phase forward that looks up and decodes every sample (the reference evaluates all 1024
    padded samples of every ray, PersSampler_cuda.cu:437-444, nerfacto_field.py:443), at the same early_T:
      * K = 1024 (the state before the first plan): identical bits everywhere;
      * K planned from the batch: rendered colour / accumulation within 1e-6 (the same samples are composited; only samples behind
        the termination points are skipped), per-sample sigma / rgb identical in front of every termination point and zero
        behind, table gradient within the f16 bar, evaluated samples < all samples;
      * K forced below the termination points (phase B has work): the same again, through both phases."""
    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    t = S["t"]
    block = 0 if stage == "focal" else None
    base = dict(early_T=1e-2, level_major_global=True)  # a high threshold: rays of the small test scene end well before their last sample
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    R = S["o"].shape[0]
    s3 = ops.s3im_index(R, 10, device="cuda")

    def engine(**kw):
        cfg = hotpath.HotPathConfig(log2_T=S["cfg"].log2_T, **base, **kw)
        hp = hotpath.HotPath(S["sc"]["nodes"], S["sc"]["transes"], n_cams=S["sc"]["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
        hp.table.mul_(300.0)  # dense enough for rays to terminate
        hp.blocks[0].uniform_(-1e-1, 1e-1, generator=torch.Generator(device="cuda").manual_seed(5))
        hp.refresh_shadows()
        return hp

    def run(hp):
        loss, ctx = hp.forward_backward(o, d, cam, gt, noise, stage=stage, block=block, s3im_index=s3)
        n = int(ctx["samples"]["pts_offsets"][-1].item())
        return dict(loss=float(loss.item()), rgb=ctx["comp"]["rgb"].clone(), acc=ctx["comp"]["acc"].clone(), w=ctx["comp"]["weights"][:n].clone(),
                    sigma=ctx["field"]["sigma"][:n].clone(), srgb=ctx["field"]["rgb"][:n].clone(), tg=hp.table_grad.float().clone(),
                    mlp=hp.mlp_grad.clone(), n=n, po=ctx["samples"]["pts_offsets"].clone())

    one = engine(two_phase=False)
    ref = run(one)
    assert one.last_forward == "lm"
    two = engine(two_phase=True)
    a = run(two)  # first step: K = 1024, phase B has nothing to do
    assert two.last_forward == "lm2"
    for k in ("rgb", "acc", "w", "sigma", "srgb"):
        assert torch.equal(a[k], ref[k]), k
    # (the table gradient is a sum in an order that differs from run to run at this table size: the record streams of a 2^14 table
    # overflow into f16 atomics)
    assert float((a["tg"] - ref["tg"]).abs().max()) <= 5e-3 * float(ref["tg"].abs().max())
    st = two._phase
    last = st.stats_last.cpu().numpy()
    assert last[3] == 1024 and last[0] == ref["n"] and last[1] == 0 and last[2] == 0
    K = int(st.k.item())
    cnt = (ref["po"][1:] - ref["po"][:-1]).cpu().numpy()
    term = st.term.cpu().numpy()
    # the plan covers every ray of this batch: the latest termination point + the width of the distribution's tail (latest - 99th percentile)
    # -- unless the split would save too few of the samples (HotPathConfig.two_phase_min_saving[_planes]): then one phase (K = 1024)
    srt = np.sort(np.minimum(term, cnt))
    f99 = srt[len(srt) - len(srt) // 100 - 1]
    K_split = int(min(1024, srt[-1] + max(2, srt[-1] - f99)))
    enter = (two.cfg.two_phase_min_saving_planes if stage == "global" else two.cfg.two_phase_min_saving)[0]
    saving = 1.0 - np.minimum(cnt, K_split).sum() / cnt.sum()
    worth = saving >= enter
    assert abs(saving - enter) > 1e-3, "the test scene sits on the threshold: change it"
    assert K == (K_split if worth else 1024) and K_split < cnt.max(), (K, K_split, srt[-1], f99, cnt.max())
    if not worth:  # this scene's rays are short: take the split anyway, to exercise it
        K = K_split
        two._phase.k.fill_(K)
    alive_front = (ref["w"] > 0)

    def check(b, label):
        np.testing.assert_allclose(b["rgb"].cpu().numpy(), ref["rgb"].cpu().numpy(), atol=1e-6, err_msg=label)
        np.testing.assert_allclose(b["acc"].cpu().numpy(), ref["acc"].cpu().numpy(), atol=1e-6, err_msg=label)
        assert torch.equal(b["w"], ref["w"]), label
        assert torch.equal(b["sigma"][alive_front], ref["sigma"][alive_front]) and torch.equal(b["srgb"][alive_front], ref["srgb"][alive_front]), label
        assert abs(b["loss"] - ref["loss"]) <= 1e-6 * abs(ref["loss"]), label
        scale = float(ref["tg"].abs().max())
        assert float((b["tg"] - ref["tg"]).abs().max()) <= 5e-3 * scale, label   # zero gradients behind the cut either way; f16 pool
        if stage == "global":
            # (round 6: wgrad_reduce_kernel adds the 512 partial vectors in one fixed order -- no atomics -- so a run repeats its bits; the phased step's
            # compacted plane rows still meet in another order than the one-phase step's, hence a floor of 1e-6 of the largest entry, not equality)
            np.testing.assert_allclose(b["mlp"].cpu().numpy(), ref["mlp"].cpu().numpy(), rtol=1e-4, atol=1e-6 * float(ref["mlp"].abs().max()), err_msg=label)

    b = run(two)  # second step: planned K
    last = two._phase.stats_last.cpu().numpy()
    assert last[3] == K and last[1] == 0 and 0 < last[0] < ref["n"], last
    assert last[0] == int(np.minimum(cnt, K).sum())
    assert float((b["sigma"][~alive_front] != 0).float().mean()) < 1.0  # something behind the termination points was never evaluated
    check(b, "planned K")
    # K below the termination points of many rays: phase B does real work
    K2 = max(8, int(np.quantile(term, 0.4)) // 8 * 8)
    two._phase.k.fill_(K2)
    c2 = run(two)
    last = two._phase.stats_last.cpu().numpy()
    need = (term > K2) & (cnt > K2)
    assert last[3] == K2 and last[1] == int(need.sum()) and last[2] == int((cnt - K2)[need].sum()) and last[1] > 0, (last, need.sum())
    check(c2, "two phases")
    assert np.array_equal(two._phase.term.cpu().numpy(), term)  # the termination points do not depend on the split


def test_octree_statistics_do_not_depend_on_the_cut(setup):
    """with early termination (early_T = 1e-4) samples behind a ray's termination point are cut from the colour and the
    backward -- but UpdateOctNodes (K5, PersSampler_cuda.cu:21-75) compares each visited node's largest ALPHA with its thresholds, and a dense
    node that a batch only reaches behind its rays' cuts must still count as occupied, like in the reference (which has no cut).  A training
    forward that K5 follows therefore evaluates every sample's density (HotPath.forward_train(all_alphas=True)): from the same state and
    batches, node statistics, visit counts and the node blob (K6's invalidations) are the early_T = 0 ones bit for bit over repeated updates --
    while the rounds 2-3 behaviour (alpha = 0 where the forward skipped the sample; HotPathConfig.exact_oct_alphas = False) is not."""
    import torch

    from gfnerf_amd import hotpath

    S = setup
    sc, t = S["sc"], S["t"]
    batches = [tuple(t(a) for a in rays_for(sc, 384, seed=s)) for s in (3, 4, 5)]
    noise = t(S["noise"])

    def run(**kw):
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=14, **kw), seed=1)
        hp.table.mul_(100.0); hp.layers[1][0].mul_(4.0); hp.refresh_shadows()   # dense enough that rays terminate inside the volume
        hp.weight_stats.fill_(3); hp.alpha_stats.fill_(3)                        # so that K6 prunes within a few updates (INIT_NODE_STAT is 1000)
        alphas, kinds, hist = [], [], []
        for k in range(8):
            o, d, cam = batches[k % 3]
            comp, ctx = hp.forward_train(o, d, cam, noise, stage="global", all_alphas=True)
            kinds.append(hp.last_forward)
            n = int(ctx["samples"]["pts_offsets"][-1].item())
            alphas.append((comp["alphas"][:n].clone(), comp["weights"][:n].clone()))
            hp.update_oct_nodes(ctx)
            hist.append((hp.weight_stats.clone(), hp.alpha_stats.clone(), hp.visit_cnt.clone(), hp.nodes.clone()))
        return hp, alphas, kinds, hist

    cut, a_cut, k_cut, h_cut = run(early_T=1e-4)
    ref, a_ref, k_ref, h_ref = run(early_T=0.0)
    old, a_old, k_old, h_old = run(early_T=1e-4, exact_oct_alphas=False)
    assert set(k_cut) == {"lm"} and set(k_old) == {"fused"}, (k_cut, k_old)  # every density evaluated | rays leave the fused kernel once opaque
    behind = 0
    for (a, w), (ar, wr) in zip(a_cut, a_ref):
        assert a.shape == ar.shape and float((a - ar).abs().max()) <= 1e-6          # the true alpha of every sample
        cutoff = (w == 0) & (wr > 0)
        behind += int(cutoff.sum())
        assert float(wr[cutoff].max() if cutoff.any() else 0.0) < 1e-4             # the weights the cut drops are below 1e-4 ...
        assert float((w - wr)[~cutoff].abs().max()) <= 1e-6
    assert behind > 100                                                             # ... and such samples exist in this scene
    for step, (c_, r_) in enumerate(zip(h_cut, h_ref)):       # after EVERY update: weight / alpha statistics, visit counts, node blob
        for name, x, y in zip(("weight_stats", "alpha_stats", "visit_cnt", "nodes"), c_, r_):
            assert torch.equal(x, y), (step, name)
    moved = int((h_ref[-1][0] != 3).sum() + (h_ref[-1][1] != 3).sum())
    assert moved > 10                                                               # the updates did touch the statistics
    # and the test can tell: rounds 2-3 fed K5 alpha = 0 behind the cut -- some node's alpha statistic was decremented where the reference's was kept
    assert any(not torch.equal(o_[1], r_[1]) for o_, r_ in zip(h_old, h_ref))


def test_psnr_of_the_product_against_the_reference_semantics():
    """BASELINE.json: "PSNR vs ref ... within 0.1 dB".  Three statements, tightest first (gfnerf_amd.procedural; ground truth pinned by
    tests/test_procedural.py; the records over many seeds at full length are profiles/r06_psnr_study.json -- the engine as the plugin surface builds and
    drives it against the reference's semantics -- and profiles/r05_psnr_study.json -- the native step against them --, made by tools/psnr_study.py):
      1. the decoder arithmetic alone: every trained state rendered with the split-f16 x3 decoder and with the exact-f32 decoder (the
         reference's nn.Linear, gfnerf/mlp.py:35-57) gives the same PSNR to 0.01 dB -- no training chaos involved;
      2. along a training run the gradient of a batch computed with either decoder has cosine >= 0.9999 in every parameter group; the product
         without the cut (x3, exact LDS sums, every sample) is within the f16 pool's rounding of the f32-pool gradient (cosine >= 0.99999); the cut
         at T < 1e-4 is what moves the gradient (cosine >= 0.995): Charbonnier's gradient is sign-like, and a colour change of up to 1e-4 flips it
         on the rays whose error is that small;
      3. trained end to end (reduced length: 1,500 + 400 steps, four seeds), the product's leg ("fast") against the reference-semantics leg
         ("reference": early_T = 0, every sample, exact-f32 decoder, f16 atomics, torch.optim.Adam; engine/trainer.py:382-441,
         gfnerf/nerfacto.py:522-619,628-689), paired by seed: a SANITY bound only, |mean difference| <= 0.1 dB + 3 x max(standard error, 0.3 dB)
         ~ 1.0 dB.  One training run is one draw (Adam at eps = 1e-15 amplifies last-bit differences; pairs scatter by ~0.5 dB), so four
         short pairs cannot resolve 0.1 dB and a tighter gate on them would fail by chance; the claim itself -- within 0.1 dB -- is the
         many-seed records', held to the protocol in the tree by tests/test_procedural.py (test_psnr_record_of_the_round, test_psnr_record_of_round_5)."""
    import torch

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import procedural, scene

    sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=__import__("os").path.join(__import__("os").environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
    kw = dict(steps_global=1500, steps_focal=400, eval_rays=8192, detail=procedural.STUDY_DETAIL, lr_final=procedural.STUDY_LR_FINAL)
    r = procedural.psnr_study(sc, seeds=(0, 1, 2, 3), legs=("fast", "reference"), **kw)
    print({k: v for k, v in r.items() if k.startswith("psnr") or k.startswith("delta") or k.startswith("decoder") or k in ("se", "n_seeds")})
    assert min(r["psnr_fast_runs"] + r["psnr_reference_runs"]) > 24.0, r                 # every run learns the scene (~7 dB at the start)
    assert max(abs(v) for v in r["decoder_delta_runs"]) <= 0.01, r["decoder_delta_runs"]   # 1.
    assert r["n_seeds"] == 4 and r["se"] is not None
    # 3.  One run scatters by ~0.5 dB and pairs decorrelate, so four short pairs have a standard error of ~0.3 dB whatever the sample says: the bound
    # here is a sanity bound (three such standard errors + the 0.1 dB of the claim); the claim itself is held by the many-seed records (the CPU tests
    # tests/test_procedural.py::test_psnr_record_of_the_round / _of_round_5 check them against the protocol in the tree)
    assert abs(r["delta"]) <= 0.1 + 3.0 * max(r["se"], 0.3), (r["delta"], r["se"], r["delta_runs"])
    c = procedural.train_and_eval(sc, seed=0, steps_global=400, steps_focal=200, eval_rays=4096, detail=procedural.STUDY_DETAIL, lr_final=procedural.STUDY_LR_FINAL, cosine_every=100,
                                  cosine_legs=("exact", "reference", "fast", "f32", "fast0"), **procedural.LEGS["fast"])["grad_cosine"]
    assert len(c) == 6
    for stage, step, pairs in c:                                                            # 2.
        for g, v in pairs["fast|f32"].items():      # the decoder arithmetic alone (everything else the product's)
            assert v["cos"] >= 0.9999, (stage, step, g, v)
        for g, v in pairs["exact|fast0"].items():   # x3 decoder + exact LDS sums, every sample, against the f32-pool gradient: f16-pool rounding only
            assert v["cos"] >= 0.99999 and v["rel_diff"] <= 3e-3, (stage, step, g, v)
        for g, v in pairs["exact|fast"].items():    # with the cut: the <= 1e-4 colour change flips Charbonnier's sign-like gradient on the few rays at |error| < 1e-4
            assert v["cos"] >= 0.995, (stage, step, g, v)
    torch.cuda.empty_cache()


def test_colour_x2_render_moves_the_colour_only():
    """gfn_launch_opts.colour_x2 (ABI 108; HotPathConfig.render_colour_x2 / TileRenderer(colour_x2=True)): the inference kernel with the colour head's
    layer 2 -- 128 -> 128, half of the decoder's MFMAs, behind head layer 1 and two layers in front of the sigmoid -- on ONE f16 activation operand (round to
    nearest) instead of the split pair: two MFMAs per product sum, no lo' half to make.  On an engine TRAINED on the procedural terrain (short: 1,500 + 400
    steps) and a COARSE march (fineness 8: few samples per ray, so a sample's colour error reaches the pixel almost undiluted -- the worst case met; full-HD
    frames at 192 samples per ray of fully trained engines are at 5.5e-5 ... 1.3e-4, tools/exp_render_x2.py), two cameras, 960 x 540 frames through the tile renderer (the
    patch kernel on the march's records, the form bench.py times):
      * nothing the density depends on is touched: accumulation and depth are the SAME BITS as the x3 form's;
      * rgb moves, and stays well inside north_star's 1e-3 bar: measured 1.5e-4 ... 2.1e-4 at the worst pixel over the round's runs of this test (a margin of
        5 x: VERDICT r05 #2's rule for switching the default was >= 3 x; one training run is one draw -- S3IM's f32 atomics -- so the figure scatters from run
        to run) -- held here to 5e-4 at the worst pixel, which that scatter cannot reach, and to 2e-5 on average over the frame (measured 7e-6 ... 8e-6);
      * either form renders the same bits twice.
    Reference: the decoder of gfnerf/nerfacto_field.py:509-555 behind Model.get_outputs_for_camera_ray_bundle, nerfstudio/models/base_model.py:163-186."""
    import os

    import torch

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import procedural, scene
    from gfnerf_amd import render as rmod

    sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
    r = procedural.train_and_eval(sc, seed=0, steps_global=1500, steps_focal=400, eval_rays=4096, detail=procedural.STUDY_DETAIL, lr_final=procedural.STUDY_LR_FINAL,
                                  keep_engine=True, **procedural.LEGS["fast"])
    assert r["psnr"] > 24.0, r["psnr"]  # a trained scene: surfaces and free space, weights far from their initialisation
    hp = r["engine"]
    H, W = 540, 960
    c2w = torch.from_numpy(sc["c2w"]).cuda()
    K = sc["intri"]
    zoom = W / 1152.0
    intr = torch.from_numpy(np.stack([K[:, 0, 0] * zoom, K[:, 1, 1] * zoom, np.full(K.shape[0], W / 2), np.full(K.shape[0], H / 2)], -1).astype(np.float32)).cuda()
    assert os.environ.get("GFN_RENDER_X2") is None  # (the measurement switch would override the argument)
    worst = 0.0
    for cam in (37, 74):
        frames = {}
        for x2 in (False, True):
            tr = rmod.TileRenderer(hp, c2w, intr, block=0, fineness=8.0, colour_x2=x2)
            out = tr.render_camera(cam, H, W)
            assert tr.record and tr.patch_tw  # the patch kernel on the march's records
            frames[x2] = {k: out[k].clone() for k in ("rgb", "accumulation", "depth")}
            again = tr.render_camera(cam, H, W)
            for k, v in frames[x2].items():
                assert torch.equal(again[k], v), (cam, x2, k)
            del tr
            torch.cuda.empty_cache()
        assert float(frames[False]["accumulation"].mean()) > 0.3
        assert torch.equal(frames[True]["accumulation"], frames[False]["accumulation"]) and torch.equal(frames[True]["depth"], frames[False]["depth"]), cam
        d = (frames[True]["rgb"] - frames[False]["rgb"]).abs()
        print(f"camera {cam}: |rgb_x2 - rgb_x3| max {float(d.max()):.2e} mean {float(d.mean()):.2e}")
        assert 0.0 < float(d.max()) <= 5e-4 and float(d.mean()) <= 2e-5, (cam, float(d.max()), float(d.mean()))
        worst = max(worst, float(d.max()))
    del hp, r
    torch.cuda.empty_cache()


def test_plugin_leg_runs_do_not_resume_each_others_block_tables():
    """The field resumes a block's table and primes from base_dir/encodings_ckpt when the block becomes active and saves it there at the end of the block's
    period (nerfacto_field.py load_table / save_table: the reference's block swap).  A study runs many seeds through the surface in one process: with one
    base_dir for all of them every run but the first started its focal stage from the PREVIOUS run's residual table (round 6: the first 58 pairs of the
    record were made that way and discarded).  procedural.train_and_eval(plugin=True) gives every run a directory of its own and removes it: two short
    runs in a row find no checkpoint when their block becomes active, each writes its own, and none is left behind."""
    import os

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import procedural, scene
    from gfnerf_amd.gfnerf import nerfacto_field

    sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
    seen = []
    load, save = nerfacto_field.GFNeRFField.load_table, nerfacto_field.GFNeRFField.save_table

    def load_probe(self, idx, strict=False):
        seen.append(("load", str(self.encodings_ckpt_dir), os.path.exists(self.encodings_ckpt_dir / f"base_encoding_{idx}.ckpt")))
        return load(self, idx, strict)

    def save_probe(self, idx):
        seen.append(("save", str(self.encodings_ckpt_dir), None))
        return save(self, idx)

    nerfacto_field.GFNeRFField.load_table, nerfacto_field.GFNeRFField.save_table = load_probe, save_probe
    try:
        for sd in (0, 1):
            procedural.train_and_eval(sc, seed=sd, steps_global=12, steps_focal=8, eval_rays=4096, **procedural.LEGS["plugin"])
    finally:
        nerfacto_field.GFNeRFField.load_table, nerfacto_field.GFNeRFField.save_table = load, save
    dirs = sorted({d for _, d, _ in seen})
    assert len(dirs) == 2, seen                                     # a directory per run
    assert all(not found for kind, _, found in seen if kind == "load"), seen   # nobody resumed anything
    assert {d for kind, d, _ in seen if kind == "save"} == set(dirs) and {d for kind, d, _ in seen if kind == "load"} == set(dirs), seen  # each run looked, and wrote its own ...
    assert not any(os.path.exists(os.path.dirname(d)) for d in dirs), dirs   # ... and its directory is gone with it


def test_backward_cu_share_follows_the_batch(setup):
    """HotPath._cu_cap: while a prefetched sampler runs beside it the focal backward keeps cfg.prefetch_cus CUs -- unless the batch is dense (at least
    cfg.prefetch_dense_frac of its samples in front of the termination points, a trained scene): then cfg.prefetch_cus_dense in a captured step (the count is
    StepGraph's planning forward's, HotPath.front_override) and cfg.prefetch_cus_dense_eager when launched eagerly (the count is the pinned word the last
    phased forward copied); the global stage keeps its own share; no prefetch, no cap.  The cap is a value handed to the ONE launch it is for
    (ops.launch_opts(decoder_cus=...), ABI 107): nothing process-wide is set, so there is nothing to put back.  (Only the grid depends on the hint; what
    does change values -- merging the record runs of a dense batch -- is decided on the device: next test.)"""
    from gfnerf_amd import hotpath, ops

    S = setup
    cfg = hotpath.HotPathConfig(log2_T=S["cfg"].log2_T, prefetch_cus=160, prefetch_cus_global=192, prefetch_dense_frac=0.75, prefetch_cus_dense=184, prefetch_cus_dense_eager=200)
    hp = hotpath.HotPath(S["sc"]["nodes"], S["sc"]["transes"], n_cams=S["sc"]["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
    try:
        assert hp._cu_cap("focal") == 0  # nothing prefetched: no cap
        hp._pre = (None, None, None, False, {})
        hp._front_host[0], hp._front_host[1] = 10, 1000
        assert hp._cu_cap("focal") == 160 and hp._cu_cap("global") == 192
        hp._front_host[0] = 800
        assert hp._cu_cap("focal") == 200 and hp._cu_cap("global") == 192  # dense, eager launch
        hp.front_override = (10, 1000)  # (a capture's planning forward says otherwise: its word counts)
        assert hp._cu_cap("focal") == 160
        hp.front_override = None
        hp.cfg.prefetch_cus_dense_eager = 256
        assert hp._cu_cap("focal") == 0  # 256 = every CU = no cap
    finally:
        hp._pre = None
    o = ops.launch_opts(decoder_cus=300)
    assert o.decoder_cus == 0 and ops.launch_opts(decoder_cus=184).decoder_cus == 184 and ops.launch_opts().record_merge == 0


def _merge_engine(S, torch, hotpath, **cfg_kw):
    hp = hotpath.HotPath(S["sc"]["nodes"], S["sc"]["transes"], n_cams=S["sc"]["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=18, early_T=1e-2, **cfg_kw), seed=0, n_blocks=1)
    hp.table.mul_(300.0)  # (T = 2^18: no record stream overflows into unordered atomics)
    hp.blocks[0].uniform_(-1e-1, 1e-1, generator=torch.Generator(device="cuda").manual_seed(5))
    hp.refresh_shadows()
    hp._record_buffers()
    hp.record_counts_seen = torch.zeros_like(hp._records[1])
    return hp


def test_record_merge_is_decided_on_the_device_from_the_steps_own_counts(setup):
    """cfg.record_merge_dense: with the phased forward's front list at hand the focal backward launches the plain and the run-merging record form as a
    gated pair (gfn_launch_opts.record_merge = GFN_MERGE_DEVICE) and the DEVICE picks one from this step's counts -- front >= cfg.prefetch_dense_frac x
    samples: the merging form.  Seen from outside: the number of records written.  A threshold no batch reaches gives the plain form's count, one every
    batch reaches the merged form's (fewer), from the very first step (no host-side hint is involved); cfg.record_merge = 0 / 1 (bench.py --record-merge)
    takes the gate out; the gradient is the same up to the f16 rounding of a run's sum either way."""
    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    t = S["t"]
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    s3 = ops.s3im_index(S["o"].shape[0], 10, device="cuda")

    def run(frac, fixed=-1):
        hp = _merge_engine(S, torch, hotpath, prefetch_dense_frac=frac, record_merge_dense=True, record_merge=fixed)  # (record_merge_dense: what the plugin surface switches on)
        hp.forward_backward(o, d, cam, gt, noise, stage="focal", block=0, s3im_index=s3)  # the FIRST step of the engine
        torch.cuda.synchronize()
        assert hp.last_forward == "lm2"
        return int(hp.record_counts_seen.to(torch.int64).sum()), hp.table_grad.float().clone()

    plain, g_plain = run(2.0)
    merged, g_merged = run(0.0)
    assert 0 < merged < 0.9 * plain, (plain, merged)
    assert run(2.0, fixed=1)[0] == merged and run(0.0, fixed=0)[0] == plain  # the fixed form rules when it is set
    assert float((g_merged - g_plain).abs().max()) <= 2e-3 * float(g_plain.abs().max())


def test_merged_record_runs_give_the_same_table_gradient(setup):
    """cfg.record_merge = 1 (GFN_MERGE_ON: the focal backward's form on dense batches): runs of consecutive samples in one cell are summed in f32 and
    written as ONE record per corner (field_bwd_dx_x3_kernel<0, 4>) -- fewer records, the same gradient up to the f16 rounding of a run's sum (every
    addend is still quantised like the reference's, x128 -> f16), gradient mass conserved per channel."""
    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    t = S["t"]
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    s3 = ops.s3im_index(S["o"].shape[0], 10, device="cuda")
    out = {}
    for merge in (0, 1):
        hp = _merge_engine(S, torch, hotpath, record_merge=merge)
        hp.forward_backward(o, d, cam, gt, noise, stage="focal", block=0, s3im_index=s3)  # (buffers exist now)
        hp.record_counts_seen.zero_()
        hp.forward_backward(o, d, cam, gt, noise, stage="focal", block=0, s3im_index=s3)
        torch.cuda.synchronize()
        out[merge] = dict(g=hp.table_grad.float().clone(), records=int(hp.record_counts_seen.to(torch.int64).sum()))
    g0, g1 = out[0]["g"], out[1]["g"]
    assert 0 < out[1]["records"] < 0.9 * out[0]["records"], (out[0]["records"], out[1]["records"])  # the merged form did run, and merged
    scale = float(g0.abs().max())
    assert scale > 0 and float((g1 - g0).abs().max()) <= 2e-3 * scale
    for ch in (0, 1):
        a, b = g0.view(-1, 2)[:, ch].double().sum(), g1.view(-1, 2)[:, ch].double().sum()
        assert abs(float(a - b)) <= 1e-3 * float(g0.view(-1, 2)[:, ch].double().abs().sum())


def test_two_engines_on_two_threads_keep_their_own_launch_policy(setup):
    """VERDICT r05 #3 / ADVICE r05: launch policy is an argument of each call (gfn_launch_opts, ABI 107), not a process-wide setting that callers set,
    launch under and restore -- ctypes releases the GIL around every call, so a render or evaluation thread beside the training thread, or two engines,
    used to race on it.  Two engines with DIFFERENT policies -- A: plain records, every CU, 16 lanes per ray; B: merged records, a backward capped to 96
    CUs behind a pending prefetch, a quad of lanes per ray -- run 24 focal steps each on two Python threads at once (separate streams), and each ends
    with the bits it produces alone on one thread: same record counts per step (the merge form is a policy), same table gradient, same samples."""
    import threading

    import torch

    from gfnerf_amd import hotpath, ops

    S = setup
    t = S["t"]
    o, d, cam, gt, noise = t(S["o"]), t(S["d"]), t(S["cam"]), t(S["gt"]), t(S["noise"])
    s3 = ops.s3im_index(S["o"].shape[0], 10, device="cuda")
    policies = dict(A=dict(record_merge=0, lanes=16, cus=0), B=dict(record_merge=1, lanes=4, cus=96))

    def work(name, n_steps, out, stream):
        pol = policies[name]
        with torch.cuda.stream(stream):
            # (Charbonnier only: S3IM scatters its gradient with f32 atomics -- like the reference's index_select backward -- whose order, and so the last bit
            # of d loss / d rgb, follows the timing; everything else of a focal step at this table size is a fixed-order or exact sum)
            hp = _merge_engine(S, torch, hotpath, record_merge=pol["record_merge"], prefetch_cus=pol["cus"] or 256, prefetch_cus_dense_eager=pol["cus"] or 256,
                               s3im_loss_mult=0.0)
            counts, lanes = [], ops.launch_opts(march_lanes=pol["lanes"])
            for _ in range(n_steps):
                smp = hp.sample(o, d, noise, opts=lanes)
                if pol["cus"]:
                    hp._pre = (None, None, None, False, {})  # (a prefetch pending: the backward's launch carries the CU cap)
                hp.record_counts_seen.zero_()
                hp.forward_backward(o, d, cam, gt, noise, stage="focal", block=0, s3im_index=s3, samples=smp)
                hp._pre = None
                counts.append(hp.record_counts_seen.to(torch.int64).sum())
            stream.synchronize()
            out[name] = dict(counts=[int(c) for c in counts], g=hp.table_grad.float().clone(), po=smp["pts_offsets"].clone(), ts=smp["ts"].clone())

    alone, both = {}, {}
    main = torch.cuda.current_stream()
    for name in policies:
        work(name, 24, alone, main)
    streams = {name: torch.cuda.Stream() for name in policies}
    errs = []

    def guarded(name):
        try:
            work(name, 24, both, streams[name])
        except Exception as e:  # noqa: BLE001
            errs.append((name, e))

    threads = [threading.Thread(target=guarded, args=(name,)) for name in policies]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs, errs
    assert 0 < alone["B"]["counts"][0] < 0.9 * alone["A"]["counts"][0]  # the two policies do differ in what they write
    for name in policies:
        assert both[name]["counts"] == alone[name]["counts"], name
        n_live = int(alone[name]["po"][-1])  # (the sample buffers are sized for 1024 per ray: what lies behind the live count is whatever the allocator handed out)
        assert torch.equal(both[name]["g"], alone[name]["g"]) and torch.equal(both[name]["po"], alone[name]["po"]), name
        assert torch.equal(both[name]["ts"][:n_live], alone[name]["ts"][:n_live]), name

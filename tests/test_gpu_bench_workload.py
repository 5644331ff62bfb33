"""Parity on the workload bench.py times: the 1,600-camera "Rubble-scale" scene (40 x 40 cameras, 10,809 tree nodes /
3,643 warp transforms), T = 2^21, 4096 rays x 192 samples (focal stage, configs[2]) and 4096 x 128 (global stage, configs[1]) --
the very engine, tables, rays, fineness and noise of bench.StageRun.  The CPU oracle cannot run these batches in seconds, so:
runs of consecutive rays and random subsets of the SAME full-size runs against the oracle (traversal / march / hash features bit
for bit; sigma, rgb, gradients within the tolerances written at the asserts), and full-size checks against float64 contractions
of what the kernels themselves stored.  Reference: PersSampler_cuda.cu:53-318, Hash3DAnchored_cuda.cu:11-155, gfnerf/mlp.py:35-57,
nerfstudio/cameras/rays.py:178-200.  All through the C ABI."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, LOG2_T = 4096, 21


def _t(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def world():
    """The bench scene and one engine per stage, set up like bench.StageRun does it."""
    import torch

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import _lib, hotpath, ops, scene

    _lib.require_device()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pipeline as opipe

    sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
    assert sc["c2w"].shape[0] == 1600 and sc["n_nodes"] > 10000 and sc["n_trans"] > 3000

    def make(stage):
        cfg = hotpath.HotPathConfig(log2_T=LOG2_T)
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=1600, cfg=cfg, seed=0, n_blocks=1 if stage == "focal" else 0)
        for i, b in enumerate(hp.blocks):
            b.uniform_(-1e-3, 1e-3, generator=torch.Generator(device="cuda").manual_seed(100 + i))
        hp.refresh_shadows()
        o, d, cam = scene.make_rays(sc["c2w"], R, seed=0 if stage == "focal" else 500)
        gt = np.random.default_rng(7).random((R, 3)).astype(np.float32)
        target = 192 if stage == "focal" else 128
        lo, hi = 0.5, 64.0
        for _ in range(14):  # bench.pick_fineness
            fin = (lo * hi) ** 0.5
            mean = hp.sample(_t(o), _t(d), _t(scene.make_noise(R, True, fin, seed=1)))["pts_offsets"][-1].item() / R
            if abs(mean - target) / target < 0.005:
                break
            lo, hi = (fin, hi) if mean > target else (lo, fin)
        noise = scene.make_noise(R, True, fin, seed=1)
        s3 = ops.s3im_index(R, 10, generator=torch.Generator(device="cuda").manual_seed(3), device="cuda")
        return dict(hp=hp, o=o, d=d, cam=cam, gt=gt, noise=noise, s3=s3, fineness=fin, mean=mean)

    return dict(sc=sc, make=make, opipe=opipe, cache={})


def _stage(world, stage):
    if stage not in world["cache"]:
        world["cache"][stage] = world["make"](stage)
    return world["cache"][stage]


def test_bench_scene_trace_and_march_bit_exact_on_runs_of_rays(world, oracle):
    """K3 + K4 on the bench's tree (depth 16, 10,809 nodes) with the bench's train-mode noise: offsets, anchors, depths, step
    lengths, warped points and first-leaf distances of three runs of 16 consecutive rays of the 4096-ray batch, bit for bit."""
    from gfnerf_amd import ops

    S, sc = _stage(world, "focal"), world["sc"]
    hp = S["hp"]
    assert abs(S["mean"] - 192) < 2
    s = hp.sample(_t(S["o"]), _t(S["d"]), _t(S["noise"]), need_world=True)
    po = s["pts_offsets"].cpu().numpy().astype(np.int64)
    assert int(s["overflow"].item()) == 0 and po[-1] > 180 * R
    nodes, transes = sc["nodes"].view(oracle.TREE_NODE_DTYPE), sc["transes"].view(oracle.TRANS_INFO_DTYPE)
    for r0, k in ((0, 16), (1777, 16), (R - 16, 16)):
        d = oracle.normalize_dirs(S["d"][r0:r0 + k])
        off, hi, nf = oracle.trace(S["o"][r0:r0 + k], d, nodes)
        m = oracle.march(S["o"][r0:r0 + k], d, S["noise"][r0:r0 + 1024 + k + 10], 1.0 / 256, True, off, hi, nf, nodes, transes)
        a, b = int(po[r0]), int(po[r0 + k])
        assert np.array_equal(po[r0:r0 + k + 1] - a, m["pts_offsets"]) and b - a > 100 * k
        assert np.array_equal(s["anchors"].cpu().numpy()[a:b].astype(np.int64), m["anchors"])
        for key in ("ts", "dists"):
            assert np.array_equal(s[key].cpu().numpy()[a:b], m[key]), key
        x = ((m["warped_pts"] + np.float32(1.5)) * (np.float32(1.0) / np.float32(3.0))).astype(np.float32)  # table space (nerfacto_field.py:431)
        assert np.array_equal(s["warped_pts"].cpu().numpy()[a:b], x)
        assert np.array_equal(s["first_oct_dis"].cpu().numpy()[r0:r0 + k], m["first_oct_dis"])


def test_bench_scene_lookup_with_its_prime_pool(world, oracle):
    """K1 at T = 2^21 with the scene's own prime pool (one prime triple per level and tree node, nerfacto.py:268) and the
    sampler's own points / warp-volume anchors: the level-major two-table lookup the focal step runs, 4,000 samples of the
    787 k against the oracle, bit for bit."""
    import torch

    from gfnerf_amd import ops

    S = _stage(world, "focal")
    hp = S["hp"]
    s = hp.sample(_t(S["o"]), _t(S["d"]), _t(S["noise"]))
    n = int(s["pts_offsets"][-1].item())
    V, T = hp.n_volumes, LOG2_T
    assert V > 10000 and int(s["anchors"][:n, 0].max().item()) < world["sc"]["n_trans"]
    feat = ops.hash_fwd_lm(hp.table16, hp.prim, None, s["warped_pts"], s["anchors"], V, T, n_dev=s["n_dev"], planar=True)
    ops.hash_fwd_lm(hp.blocks16[0], hp.block_prims[0], None, s["warped_pts"], s["anchors"], V, T, n_dev=s["n_dev"], planar=True, out=feat, accumulate=True)
    rng = np.random.default_rng(4)
    idx = np.unique(np.concatenate([[0, n - 1], rng.integers(0, n, 4000)]))
    it = torch.from_numpy(idx).cuda()
    got = feat[:, it, :].permute(1, 0, 2).reshape(idx.size, 32).cpu().numpy()
    P = hp.export_params()
    x, a = s["warped_pts"][it].cpu().numpy(), s["anchors"][it, 0].cpu().numpy().astype(np.int64)
    want = oracle.hash_fwd(P["table"], P["prim"], P["bias"], x, a, V, T) + oracle.hash_fwd(P["blocks"][0], P["block_prims"][0], P["bias"], x, a, V, T)
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


def _g_rgb(S, rgb):
    """d(Charbonnier + S3IM)/d rgb of the full batch, from the product's own (golden-pinned) loss kernels."""
    from gfnerf_amd import ops

    loss, g = ops.charbonnier(rgb, _t(S["gt"]))
    ops.s3im(_t(S["gt"]), rgb, S["s3"], patch_h=32, mult=1.0, loss=loss, g_pred=g)
    return loss, g


def _oracle_rays(world, S, P, r0, k, block, g_rgb_rows):
    """Oracle forward of rays [r0, r0 + k) of the batch + its backward from the given d loss / d rgb rows."""
    import oracle as O

    opipe = world["opipe"]
    m = opipe.sample(P, S["o"][r0:r0 + k], S["d"][r0:r0 + k], S["noise"][r0:r0 + 1024 + k + 10])
    f = opipe.field_forward(P, m, S["cam"][r0:r0 + k], LOG2_T, block)
    comp = O.composite_fwd(m["pts_offsets"], f["sigma"], m["dists"], m["ts"], f["rgb"], early_T=1e-4)
    # the product stops a ray where its transmittance falls below early_T = 1e-4 (the reference composites every sample: the colour
    # differs by < 1e-4, tests/test_gpu_pipeline.py::test_early_termination_within_tolerance); the oracle's backward has no such
    # switch, so it is given every ray cut at the same sample: the prefix with T >= 1e-4
    off = m["pts_offsets"]
    keep = np.zeros(f["sigma"].shape[0], bool)
    for r in range(k):
        ds = (m["dists"][off[r]:off[r + 1]] * f["sigma"][off[r]:off[r + 1]]).astype(np.float32)
        T = np.exp(-(np.cumsum(ds, dtype=np.float32) - ds))
        keep[off[r]:off[r + 1]] = T >= np.float32(1e-4)
    cut_off = np.concatenate([[0], np.cumsum([keep[off[r]:off[r + 1]].sum() for r in range(k)])]).astype(off.dtype)
    ds_k, dr_k = O.composite_bwd(cut_off, f["sigma"][keep], m["dists"][keep], f["rgb"][keep], g_rgb_rows, None)
    d_sigma, d_rgb = np.zeros_like(f["sigma"]), np.zeros_like(f["rgb"])
    d_sigma[keep], d_rgb[keep] = ds_k, dr_k
    (bW0, _), (bW1, _), (hW0, _), (hW1, _), (hW2, _) = P["layers"]
    B = {}
    B["dz3"] = (d_rgb * f["rgb"] * (1 - f["rgb"])).astype(np.float32)
    dc2, B["hW2"], B["hb2"] = O.linear_bwd(f["c2"], hW2, f["rgb"], d_rgb, "sigmoid")
    B["dz2"] = dc2 * (f["c2"] > 0)
    dc1, B["hW1"], B["hb1"] = O.linear_bwd(f["c1"], hW1, f["c2"], dc2, "relu")
    B["dz1"] = dc1 * (f["c1"] > 0)
    dxh, B["hW0"], B["hb0"] = O.linear_bwd(f["xh"], hW0, f["c1"], dc1, "relu")
    dh = np.zeros_like(f["h"])
    dh[:, 0] = O.density_bwd(f["h"], d_sigma)
    dh[:, 1:16] = dxh[:, 16:31]
    B["dh"] = dh
    demb = np.zeros(P["emb"].shape, np.float64)  # d xh[:, 31:63] summed per camera (nerfacto_field.py:526-538)
    np.add.at(demb, np.asarray(S["cam"][r0:r0 + k])[m["ray_id"]], dxh[:, 31:63].astype(np.float64))
    B["emb"] = demb
    da1, B["bW1"], B["bb1"] = O.linear_bwd(f["a1"], bW1, f["h"], dh, "none")
    B["dza1"] = da1 * (f["a1"] > 0)
    dfeat, B["bW0"], B["bb0"] = O.linear_bwd(f["feat"], bW0, f["a1"], da1, "relu")
    prim = P["prim"] if block is None else P["block_prims"][block]
    B["table"] = O.hash_bwd(prim, P["bias"], m["x"], np.ascontiguousarray(m["anchors"][:, 0]), dfeat, P["prim"].shape[1], LOG2_T)
    return m, f, comp, B


@pytest.mark.parametrize("stage", ["focal", "global"])
def test_bench_step_forward_and_backward_against_oracle_on_a_run_of_rays(world, oracle, stage):
    """One forward_backward of the bench's batch (focal: global + block table, 787 k samples; global: 524 k samples, decoder
    trains).  (1) INSIDE the full batch, rays [r0, r0 + 20): per-sample sigma / rgb / compositing weights and the pixel colours
    equal the oracle's (rgb / sigma within 1e-3 abs resp. 1e-4 rel: the north-star bar is 1e-3).  (2) the backward: the same
    rays as their own batch, driven by their rows of the full batch's d loss / d rgb -> the table gradient against the oracle's
    f32 scatter (f16 x128 contributions: 2e-2 rel / 3e-3 of the largest entry) [+ decoder / embedding gradients, global stage].
    (3) the full batch's table gradient is the sum of its parts: two half-batches' pools add up to it within f16 rounding."""
    import torch

    from gfnerf_amd import ops

    S = _stage(world, stage)
    hp = S["hp"]
    block = 0 if stage == "focal" else None
    o, d, cam, noise = _t(S["o"]), _t(S["d"]), _t(S["cam"]), _t(S["noise"])
    comp, ctx = hp.forward_train(o, d, cam, noise, stage=stage, block=block)
    s, f = ctx["samples"], ctx["field"]
    n = int(s["pts_offsets"][-1].item())
    po = s["pts_offsets"].cpu().numpy().astype(np.int64)
    loss, g_rgb = _g_rgb(S, comp["rgb"])
    hp.backward(ctx, g_rgb)
    full_pool = hp.table_grad.float().cpu().numpy().reshape(-1) / 128.0
    assert np.isfinite(float(loss.item())) and (full_pool != 0).sum() > 1_000_000
    P = hp.export_params()
    r0, k = 2100, 20
    m, fo, co, Bo = _oracle_rays(world, S, P, r0, k, block, g_rgb[r0:r0 + k].cpu().numpy())
    a, b = int(po[r0]), int(po[r0 + k])
    assert np.array_equal(po[r0:r0 + k + 1] - a, m["pts_offsets"]) and b - a > 80 * k
    # (1) forward inside the full batch
    alive = co["weights"] > 0  # behind the termination point (T < 1e-4) nothing is evaluated that the image can see
    np.testing.assert_allclose(f["sigma"][a:b].cpu().numpy()[alive], fo["sigma"][alive], rtol=2e-4)
    np.testing.assert_allclose(f["rgb"][a:b].cpu().numpy()[alive], fo["rgb"][alive], atol=1e-4)
    np.testing.assert_allclose(comp["weights"][a:b].cpu().numpy(), co["weights"], atol=1e-5)
    np.testing.assert_allclose(comp["rgb"][r0:r0 + k].cpu().numpy(), co["rgb"], atol=1e-4)
    np.testing.assert_allclose(comp["acc"][r0:r0 + k].cpu().numpy(), np.asarray(co["acc"]).reshape(-1), atol=1e-4)
    # (2) backward of the same rays as their own batch
    sub = slice(r0, r0 + k)
    comp_s, ctx_s = hp.forward_train(o[sub].contiguous(), d[sub].contiguous(), cam[sub].contiguous(), noise[r0:r0 + 1024 + k + 10].contiguous(), stage=stage, block=block)
    assert torch.equal(ctx_s["samples"]["ts"][: b - a], s["ts"][a:b])
    hp.backward(ctx_s, g_rgb[sub].contiguous())
    got = hp.table_grad.float().cpu().numpy() / 128.0
    want = Bo["table"]
    scale = np.abs(want).max()
    assert (want != 0).sum() > 10000
    # every contribution is rounded to f16 at x128 (Hash3DAnchored_cuda.cu:150): half a quantum of 2^-24 / 128 each; the coarse levels
    # collect a few hundred contributions of these 20 rays per slot
    quantum = 2.0 ** -24 / 128
    np.testing.assert_allclose(got, want, rtol=2e-2, atol=3e-3 * scale + 48 * quantum)
    if stage == "global":
        # decoder gradients: with the weight gradient's operands as hi + lo' planes ("x3": the arithmetic, 2e-5 of the largest entry)
        # and as the product keeps them ("rne": one f16 plane each, rounded to nearest -- over these 20 rays' 2,500 samples the rounding
        # has little to average over: 2^-12 per term, 3e-4 of the largest entry; over the whole batch: the planes test below, 1e-4)
        names = ["bW0", "bb0", "bW1", "bb1", "hW0", "hb0", "hW1", "hb1", "hW2", "hb2"]
        assert hp.cfg.wgrad_planes == "rne"
        for mode, tol in (("rne", 3e-4), ("x3", 2e-5)):
            if mode != hp.cfg.wgrad_planes:
                hp.cfg.wgrad_planes, hp._plane_bufs = mode, None
                comp_s, ctx_s = hp.forward_train(o[sub].contiguous(), d[sub].contiguous(), cam[sub].contiguous(), noise[r0:r0 + 1024 + k + 10].contiguous(), stage=stage, block=block)
                hp.backward(ctx_s, g_rgb[sub].contiguous())
            for i, (gW, gB) in enumerate(hp.layer_grads):
                for gg, wn in ((gW, names[2 * i]), (gB, names[2 * i + 1])):
                    w = np.asarray(Bo[wn])
                    np.testing.assert_allclose(gg.cpu().numpy(), w, rtol=1e-4, atol=tol * np.abs(w).max() + 1e-12, err_msg=f"{wn} ({mode})")
            np.testing.assert_allclose(hp.emb_grad.cpu().numpy(), Bo["emb"], rtol=1e-4, atol=2e-5 * np.abs(Bo["emb"]).max())
        hp.cfg.wgrad_planes, hp._plane_bufs = "rne", None
    # (3) linearity over rays: pool(rays 0..2047) + pool(rays 2048..4095) = pool(all), each rounded to f16 once
    halves = np.zeros_like(full_pool)
    for h0, h1 in ((0, R // 2), (R // 2, R)):
        hs = slice(h0, h1)
        _, ctx_h = hp.forward_train(o[hs].contiguous(), d[hs].contiguous(), cam[hs].contiguous(), noise[h0:h0 + 1024 + (h1 - h0) + 10].contiguous(), stage=stage, block=block)
        assert int(ctx_h["samples"]["pts_offsets"][-1].item()) == int(po[h1] - po[h0])
        hp.backward(ctx_h, g_rgb[hs].contiguous())
        halves += hp.table_grad.float().cpu().numpy().reshape(-1) / 128.0
    mass = np.abs(full_pool).sum()
    assert np.abs(halves - full_pool).sum() <= (2e-3 if stage == "focal" else 2e-2) * mass, np.abs(halves - full_pool).sum() / mass
    assert abs(halves.sum() - full_pool.sum()) <= 1e-3 * mass


@pytest.mark.parametrize("mode,forward", [("x3", "fused"), ("rne", "fused"), ("rne", "lm"), ("mixed", "lm"), ("rne-plain", "lm")])
def test_bench_global_step_planes_scale_and_weight_gradients_at_full_size(world, oracle, mode, forward):
    """The global stage at 4096 x 128 (524 k samples).  gfn_grad_max / gfn_grad_scale_of: the batch scale is the power of two
    that brings the largest incoming gradient into [8, 16).  The activation planes the forward keeps and the dz planes the
    backward writes (tile-major; `mode` = HotPathConfig.wgrad_planes: "x3" hi + lo' for both operands, "rne" ONE plane per operand
    rounded to nearest, "mixed" dz one plane / activations split; `forward`: the fused kernel or lookup + decoder kernels) read back
    as the oracle's activations and pre-activation gradients on rays [900, 916).  `wgrad_x3_layers`: dW / db of all five layers
    equal a float64 contraction dz^T a / scale over ALL samples of the very planes it read (2e-6 of sum |dz||a|: the kernel's own
    arithmetic), and -- the one-plane modes' accuracy -- the float64 contraction of the x3 planes (22-bit
    operands, taken as exact) to 1e-4 of each layer's largest |dW|: the achieved figure is printed.  "rne-plain": the one-plane
    mode WITHOUT its per-sample dither (HotPathConfig.plane_dither = False: gfn_launch_opts.plain_rounding), reported for comparison and held to 3e-4 only."""
    import torch

    from gfnerf_amd import ops

    S = _stage(world, "global")
    hp = S["hp"]
    prev = (hp.cfg.wgrad_planes, hp.cfg.level_major_global, hp.phase_override, hp.cfg.plane_dither)
    plain = mode == "rne-plain"
    mode = "rne" if plain else mode
    hp.cfg.wgrad_planes, hp.cfg.level_major_global, hp.phase_override = mode, forward == "lm", False  # (one phase: every plane row written)
    hp.cfg.plane_dither = not plain
    hp._plane_bufs = None  # the planes are allocated for one mode
    try:
        _planes_and_weight_gradients(world, S, hp, mode, forward, torch, ops, bar=3e-4 if plain else 1e-4)
    finally:
        hp.cfg.wgrad_planes, hp.cfg.level_major_global, hp.phase_override, hp.cfg.plane_dither = prev
        hp._plane_bufs = None


def _wgrad_exact(world, S, hp, torch, ops):
    """The float64 contraction dz^T a of the x3 planes (22-bit operands) over all samples, per layer: what the one-plane modes are held to.  Made by
    the "x3" case when it runs first, and HERE when a one-plane case runs without it (-k rne, another order, another worker): the accuracy gate never
    passes for want of its reference."""
    exact = world["cache"].setdefault("wgrad_exact", {})
    if len(exact) == 5:
        return exact
    prev = (hp.cfg.wgrad_planes, hp.cfg.level_major_global)
    hp.cfg.wgrad_planes, hp.cfg.level_major_global, hp._plane_bufs = "x3", False, None
    try:
        o, d, cam, noise = _t(S["o"]), _t(S["d"]), _t(S["cam"]), _t(S["noise"])
        comp, ctx = hp.forward_train(o, d, cam, noise, stage="global")
        s = ctx["samples"]
        n = int(s["pts_offsets"][-1].item())
        _, g_rgb = _g_rgb(S, comp["rgb"])
        hp.backward(ctx, g_rgb)
        scale = ops.grad_scale(float(hp._gmax.item()))
        fp, bp = hp._planes(s["ts"].shape[0])
        pl = lambda pair, ld, name, table: (ops.plane_view(pair[0], ld, name, table)[:n].double() +  # noqa: E731
                                            ops.plane_view(pair[1], ld, name, table)[:n].double() / 2048.0)
        cm = torch.from_numpy(hp._xp_col_map.cpu().numpy()).cuda()
        for li, dzk, ak, mv in ((4, "dz3", "c2", 3), (3, "dz2", "c1", 128), (2, "dz1", "xp", 128), (1, "dh", "a1", 16), (0, "dza1", "feat", 128)):
            dz = pl(bp, ops.FIELD_X3_BWD_PLANE_LD, dzk, ops.BWD_PLANE)[:, :mv] / scale
            want = dz.T @ pl(fp, ops.FIELD_X3_FWD_PLANE_LD, ak, ops.FWD_PLANE)
            if ak == "xp":
                w2 = torch.zeros((mv, 63), dtype=torch.float64, device="cuda")
                w2[:, cm[cm >= 0].long()] = want[:, cm >= 0]
                want = w2
            exact[li] = (want, dz.sum(0))
    finally:
        hp.cfg.wgrad_planes, hp.cfg.level_major_global = prev
        hp._plane_bufs = None
    return exact


def _planes_and_weight_gradients(world, S, hp, mode, forward, torch, ops, bar=1e-4):
    if mode != "x3":
        _wgrad_exact(world, S, hp, torch, ops)
    o, d, cam, noise = _t(S["o"]), _t(S["d"]), _t(S["cam"]), _t(S["noise"])
    comp, ctx = hp.forward_train(o, d, cam, noise, stage="global")
    s, f = ctx["samples"], ctx["field"]
    n = int(s["pts_offsets"][-1].item())
    assert abs(n / R - 128) < 2 and hp.last_forward == forward, hp.last_forward
    loss, g_rgb = _g_rgb(S, comp["rgb"])
    d_sigma, d_rgb = ops.composite_bwd(s["pts_offsets"], f["sigma"], s["dists"], f["rgb"], g_rgb, None, early_T=hp.cfg.early_T)
    for gW, gB in hp.layer_grads:
        gW.zero_(); gB.zero_()
    hp.backward(ctx, g_rgb)
    # the batch scale
    gs = (d_sigma[:n] * torch.exp(torch.clamp(f["h0"][:n] + 1.0, -15.0, 15.0))).abs().max()
    gmax_want = float(torch.maximum(gs, d_rgb[:n].abs().max()).item())
    assert abs(float(hp._gmax.item()) - gmax_want) <= 1e-6 * gmax_want
    scale = ops.grad_scale(float(hp._gmax.item()))
    assert 8.0 <= gmax_want * scale < 16.0 and np.log2(scale) == round(np.log2(scale))
    fp, bp = hp._planes(s["ts"].shape[0])
    assert (fp[1] is None) == (mode == "rne") and (bp[1] is None) == (mode != "x3")

    def plane(pair, ld, name, table):
        hi = ops.plane_view(pair[0], ld, name, table)[:n].double()
        return hi if pair[1] is None else hi + ops.plane_view(pair[1], ld, name, table)[:n].double() / 2048.0

    A = {k: plane(fp, ops.FIELD_X3_FWD_PLANE_LD, k, ops.FWD_PLANE) for k in ("feat", "a1", "xp", "c1", "c2")}
    bld = ops.FIELD_X3_BWD_PLANE_LD if bp[1] is not None else ops.FIELD_X3_BWD_PLANE_LD_RNE
    DZ = {k: plane(bp, bld, k, ops.BWD_PLANE) / scale for k in ("dz3", "dh", "dz2", "dz1", "dza1")}
    if bp[1] is None:  # the one-plane form keeps dza1's lo' half (same buffer)
        DZ["dza1"] = DZ["dza1"] + ops.plane_view(bp[0], bld, "dza1_lo", ops.BWD_PLANE)[:n].double() / 2048.0 / scale
    # ---- the planes against the oracle on a run of rays (one-plane modes: the values rounded to f16, 2^-11 relative)
    P = hp.export_params()
    r0, k = 900, 16
    po = s["pts_offsets"].cpu().numpy().astype(np.int64)
    m, fo, co, Bo = _oracle_rays(world, S, P, r0, k, None, g_rgb[r0:r0 + k].cpu().numpy())
    a, b = int(po[r0]), int(po[r0 + k])
    assert np.array_equal(po[r0:r0 + k + 1] - a, m["pts_offsets"])
    alive = co["weights"] > 0
    cm = hp._xp_col_map.cpu().numpy()  # xp plane column -> column of the head's 63-wide input
    rt_a = 1e-4 if fp[1] is not None else 1.2e-3  # one plane: rounding 2^-11 + the sample's dither factor 1 +- 2^-11
    for name, want in (("feat", fo["feat"]), ("a1", fo["a1"]), ("c1", fo["c1"]), ("c2", fo["c2"])):
        got = A[name][a:b].cpu().numpy()
        np.testing.assert_allclose(got[alive], want[alive], rtol=rt_a, atol=2e-5 * np.abs(want).max(), err_msg=name)
    xp = A["xp"][a:b].cpu().numpy()
    for col in range(64):
        if cm[col] >= 0:
            np.testing.assert_allclose(xp[alive, col], fo["xh"][alive, cm[col]], rtol=1e-3, atol=1e-3 * np.abs(fo["xh"]).max())
    for name in ("dz3", "dz2", "dz1", "dh", "dza1"):
        want = np.asarray(Bo[name], np.float64)
        got = DZ[name][a:b].cpu().numpy()[:, : want.shape[1]]
        np.testing.assert_allclose(got, want, rtol=5e-3, atol=2e-3 * np.abs(want).max(), err_msg=name)
    # ---- weight gradients = dz^T a over all 524 k samples, in float64 on the device
    cmt = torch.from_numpy(cm).cuda()
    exact = world["cache"].setdefault("wgrad_exact", {})  # the x3 run's contraction: 22-bit operands, "exact" for the one-plane modes
    report = []
    for li, dzk, ak, mv in ((4, "dz3", "c2", 3), (3, "dz2", "c1", 128), (2, "dz1", "xp", 128), (1, "dh", "a1", 16), (0, "dza1", "feat", 128)):
        dz, act = DZ[dzk][:, :mv], A[ak]
        want = dz.T @ act
        mag = float((dz.abs().T @ act.abs()).max())
        gW, gB = hp.layer_grads[li]
        if ak == "xp":
            w2 = torch.zeros((mv, 63), dtype=torch.float64, device="cuda")
            valid = cmt >= 0
            w2[:, cmt[valid].long()] = want[:, valid]
            want = w2
        assert float((gW.double() - want).abs().max()) < 2e-6 * mag, (li, float((gW.double() - want).abs().max()), mag)
        np.testing.assert_allclose(gB.double().cpu().numpy(), dz.sum(0).cpu().numpy(), rtol=1e-4, atol=2e-6 * float(dz.abs().sum(0).max()))
        if mode == "x3":
            exact[li] = (want.clone(), dz.sum(0).clone())
        elif li in exact:
            wW, wB = exact[li]
            eW = float((gW.double() - wW).abs().max()) / float(wW.abs().max())
            eB = float((gB.double() - wB).abs().max()) / float(wB.abs().max())
            cos = float((gW.double() * wW).sum() / (gW.double().norm() * wW.norm()))
            report.append((li, eW, eB, cos))
    if mode != "x3":
        print(f"\n[wgrad {mode}/{forward}] max |dW - exact| / max |dW| per layer (layer, dW, db, cosine): " +
              ", ".join(f"({li}, {eW:.2e}, {eB:.2e}, {cos:.8f})" for li, eW, eB, cos in report))
        assert len(report) == 5, "the one-plane accuracy gate has no x3 reference to compare with"  # (made by _wgrad_exact when the x3 case did not run first)
        for li, eW, eB, cos in report:
            assert eW <= bar and eB <= bar and cos >= 0.99999, (mode, li, eW, eB, cos)


@pytest.mark.parametrize("mode", ["rne", "x3"])
def test_bench_global_step_decoder_gradient_repeats_its_bits(world, mode):
    """the decoder's weight gradient at the bench's size (524 k samples, 512 per-workgroup partial vectors) is a fixed-order sum --
    wgrad_x3_layers_kernel's slabs are dealt statically, wgrad_reduce_kernel adds the partial vectors in one order, no atomics (round 5 ended with f32
    atomics from four workgroups per element: one run in three differed in an entry's last bits, gfnerf/mlp.py's cuBLAS GEMMs do not) -- so two
    backwards from the same forward and the same d loss / d rgb give the same bits in every layer's dW and db, ten times over."""
    import torch

    S = _stage(world, "global")
    hp = S["hp"]
    prev = (hp.cfg.wgrad_planes, hp.phase_override)
    hp.cfg.wgrad_planes, hp.phase_override, hp._plane_bufs = mode, False, None
    try:
        o, d, cam, noise = _t(S["o"]), _t(S["d"]), _t(S["cam"]), _t(S["noise"])
        comp, ctx = hp.forward_train(o, d, cam, noise, stage="global")
        _, g_rgb = _g_rgb(S, comp["rgb"])
        first = None
        for rep in range(10):
            hp.mlp_grad.zero_()
            hp.backward(ctx, g_rgb)
            got = hp.mlp_grad.clone()
            if first is None:
                first = got
                assert float(first.abs().max()) > 0
            assert torch.equal(got, first), (mode, rep, float((got - first).abs().max()))
    finally:
        hp.cfg.wgrad_planes, hp.phase_override = prev
        hp._plane_bufs = None


def test_bench_scene_octree_statistics_bit_exact_on_a_run_of_rays(world, oracle):
    """K5 / K6 (UpdateOctNodes, PersSampler_cuda.cu:518-665) on the bench's tree from the global stage's real weights and alphas:
    256 consecutive rays of the batch as their own update, node statistics / visit counts / invalidated leaves bit for bit."""
    import torch

    from gfnerf_amd import ops

    S, sc = _stage(world, "global"), world["sc"]
    hp = S["hp"]
    r0, k = 1500, 256
    o, d, cam, noise = _t(S["o"][r0:r0 + k]), _t(S["d"][r0:r0 + k]), _t(S["cam"][r0:r0 + k]), _t(S["noise"][r0:r0 + 1024 + k + 10])
    comp, ctx = hp.forward_train(o, d, cam, noise, stage="global")
    s = ctx["samples"]
    n = int(s["pts_offsets"][-1].item())
    n_nodes = sc["n_nodes"]
    nodes_ref = sc["nodes"].copy().view(oracle.TREE_NODE_DTYPE)
    rng = np.random.default_rng(3)
    ws = np.full(n_nodes, 1000, np.int64); as_ = np.full(n_nodes, 1000, np.int64); vc = np.zeros(n_nodes, np.int64)
    low = rng.random(n_nodes) < 0.3  # some nodes one visit away from invalidation
    ws[low] = 0; as_[low] = 0
    nodes_t, ws_t, as_t, vc_t = _t(sc["nodes"].copy()), _t(ws), _t(as_), _t(vc)
    w, al = comp["weights"], comp["alphas"]
    oracle.update_nodes(s["pts_offsets"].cpu().numpy().astype(np.int64), s["anchors"][:n, 1].cpu().numpy().astype(np.int64).copy(),
                        w[:n].cpu().numpy(), al[:n].cpu().numpy(), ws, as_, vc, nodes_ref)
    ops.update_nodes(s["pts_offsets"], s["anchors"], w, al, ws_t, as_t, vc_t, nodes_t)
    assert np.array_equal(ws_t.cpu().numpy(), ws) and np.array_equal(as_t.cpu().numpy(), as_) and np.array_equal(vc_t.cpu().numpy(), vc)
    assert np.array_equal(nodes_t.cpu().numpy(), nodes_ref.view(np.uint8).reshape(-1))
    assert (nodes_ref["trans_idx"] != sc["nodes"].view(oracle.TREE_NODE_DTYPE)["trans_idx"]).sum() > 0 and vc.sum() > 0


def test_bench_step_phased_forward_is_the_one_phase_forward_at_full_size(world):
    """The termination-aware forward the bench's focal step runs (K planned on the device from the batch: phase A = the first K samples
    of every ray, phase B = the rest of the rays still transparent) against the one-phase forward over all 787 k samples at the same
    early_T, on the bench's batch: the rendered colours agree to 1e-6 (north-star bar 1e-3), sigma / rgb are identical in front of every
    termination point and zero behind, fewer samples are looked up, and the table gradient of one backward agrees within the f16 bar."""
    import torch

    from gfnerf_amd import hotpath, ops

    S, sc = _stage(world, "focal"), world["sc"]
    base = S["hp"]
    o, d, cam, noise = _t(S["o"]), _t(S["d"]), _t(S["cam"]), _t(S["noise"])

    def engine(two_phase):
        cfg = hotpath.HotPathConfig(log2_T=LOG2_T, two_phase=two_phase)
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=1600, cfg=cfg, seed=0, n_blocks=1)
        hp.blocks[0].copy_(base.blocks[0])
        hp.refresh_shadows()
        return hp

    one, two = engine(False), engine(True)
    c1, x1 = one.forward_train(o, d, cam, noise, stage="focal", block=0, want_depth=False)
    two.forward_train(o, d, cam, noise, stage="focal", block=0, want_depth=False)  # K = 1024: plans K from this batch
    K = int(two._phase.k.item())
    assert 64 <= K < 192, K  # the split pays on this batch (bench: K ~ 130)
    c2, x2 = two.forward_train(o, d, cam, noise, stage="focal", block=0, want_depth=False)
    assert one.last_forward == "lm" and two.last_forward == "lm2"
    n = int(x1["samples"]["pts_offsets"][-1].item())
    nA, nB_rays, nB, K_used = (int(v) for v in two._phase.stats_last.tolist())
    assert K_used == K and nA < 0.75 * n and nB_rays <= 8, (nA, n, nB_rays)
    assert float((c2["rgb"] - c1["rgb"]).abs().max()) < 1e-6 and float((c2["acc"] - c1["acc"]).abs().max()) < 1e-6
    w1, w2 = c1["weights"][:n], c2["weights"][:n]
    alive = w1 > 0
    assert torch.equal(w1, w2) and 0.5 < float(alive.float().mean()) < 0.65
    f1, f2 = x1["field"], x2["field"]
    assert torch.equal(f1["sigma"][:n][alive], f2["sigma"][:n][alive]) and torch.equal(f1["rgb"][:n][alive], f2["rgb"][:n][alive])
    listed = torch.zeros(n, dtype=torch.bool, device="cuda")
    listed[two._phase.list_a[:nA].long()] = True
    if nB:
        listed[two._phase.list_b[:nB].long()] = True
    assert bool((f2["sigma"][:n][~listed] == 0).all()) and bool(alive[~listed].sum() == 0)
    # one backward each from the same d loss / d rgb
    g = torch.from_numpy(np.random.default_rng(5).standard_normal((R, 3)).astype(np.float32) * 1e-3).cuda()
    one.backward(x1, g); two.backward(x2, g)
    p1, p2 = one.table_grad.float() / 128.0, two.table_grad.float() / 128.0
    scale = float(p1.abs().max())
    assert scale > 0 and int((p1 != 0).sum()) > 1_000_000
    assert float((p1 - p2).abs().max()) <= 2e-3 * scale  # f16 x128 contributions; the exact LDS accumulation sums the same records


def test_bench_scene_full_frame_tile_render(world, oracle):
    """BASELINE configs[4] at the size bench.py's `render` sub-record times (VERDICT r03 #2b): a 1920 x 1080 frame of the 1,600-camera scene,
    T = 2^21, global + focal table, through render.TileRenderer exactly as bench.render_bench builds it -- choose_tile's 960 x 540 pixel
    blocks, all 4 tiles in ONE hipGraph replay, samplers (traversal + counting march: the samples stay in the march's records) on the producer
    branch through the ring of two record buffers, gfn_fused_render_rec (patches of 8 x 4 neighbouring pixels per wave) on the consumer branch.
      * the frame is finite, colours in [0, 1], accumulation in [0, 1];
      * tiles are seamless: the 8 rows around the tile edge y = 540 and the 8 columns around x = 960 equal one direct HotPath.render of those
        rays (the wave-per-ray fused forward on packed samples: another kernel, another launch shape, another form of the samples) to 2e-6;
      * a run of 32 pixels across the corner (row 540, columns 944..975: two tiles, the first row of the second tile row) against the CPU
        oracle pipeline: rgb within 1e-3 abs (north_star's bar), accumulation 1e-4;
      * a second camera through the same captured graph: the same run against a direct render;
      * the frame rendered three more times: the same bits (round 6);
      * the renderer's default form (colour_x2: head layer 2 on one f16 activation operand): accumulation and depth the same bits, rgb within 2e-5 of the x3
        frame on this random-init scene and within 1e-3 of the oracle on the same run (a trained scene: tests/test_gpu_pipeline.py::test_colour_x2_...).
    Reference: Model.get_outputs_for_camera_ray_bundle, nerfstudio/models/base_model.py:163-186."""
    import torch

    from gfnerf_amd import ops, scene
    from gfnerf_amd import render as rmod

    S, sc = _stage(world, "focal"), world["sc"]
    hp = S["hp"]
    H, W = 1080, 1920
    c2w = _t(sc["c2w"])
    K = sc["intri"]
    zoom = W / 1152.0
    intr = _t(np.stack([K[:, 0, 0] * zoom, K[:, 1, 1] * zoom, np.full(K.shape[0], W / 2), np.full(K.shape[0], H / 2)], -1).astype(np.float32))
    fin = S["fineness"]
    tr = rmod.TileRenderer(hp, c2w, intr, block=0, fineness=fin, use_graph=True, colour_x2=False)  # (the x3 form: compared with other kernels to 2e-6 below)
    cam0 = 37
    img = tr.render_camera(cam0, H, W)
    tw, th = tr.tw, tr.T // tr.tw
    assert (tw, th, tr.P, tr.SLOTS) == (960, 540, 4, 2) and tr.patch_tw == tw and tr.graph is not None
    rgb, acc, dep = img["rgb"], img["accumulation"], img["depth"]
    assert rgb.shape == (H, W, 3) and acc.shape == (H, W)
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(acc).all()) and bool(torch.isfinite(dep).all())
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 and float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    assert float(acc.mean()) > 0.5  # the frame looks at the scene
    # a frame is the same bits every time it is rendered (round 6: an instantiation of the inference kernel whose hashes the compiler had fused into
    # v_bitop3_b32 got a handful of a frame's 2 M pixels wrong, differently from run to run -- csrc/fused_fwd.hip, hash_corners_split)
    first = {k: img[k].clone() for k in ("rgb", "accumulation", "depth")}
    for _ in range(3):
        again = tr.render_camera(cam0, H, W)
        for k, v in first.items():
            assert torch.equal(again[k], v), k

    rays = ops.generate_rays(c2w, intr, cam0=cam0, first_pixel=0, width=W, n=H * W, want_extras=False)
    O, D = rays["origins"].view(H, W, 3), rays["directions"].view(H, W, 3)

    def direct(o, d, cam):
        n = o.shape[0]
        noise = torch.full((1024 + n + 10,), float(fin), device="cuda")
        return hp.render(o.contiguous(), d.contiguous(), torch.full((n,), cam, dtype=torch.int32, device="cuda"), noise, block=0)

    for name, sl in ((f"rows around y = {th}", (slice(th - 4, th + 4), slice(0, W))), (f"columns around x = {tw}", (slice(0, H), slice(tw - 4, tw + 4)))):
        ref = direct(O[sl].reshape(-1, 3), D[sl].reshape(-1, 3), cam0)
        got = rgb[sl].reshape(-1, 3)
        assert float((got - ref["rgb"]).abs().max()) <= 2e-6, name
        assert float((acc[sl].reshape(-1) - ref["accumulation"][:, 0]).abs().max()) <= 2e-6, name

    # the oracle on a run of 32 pixels across the tile corner
    y, x0 = th, tw - 16
    o_np, d_np = O[y, x0:x0 + 32].cpu().numpy(), D[y, x0:x0 + 32].cpu().numpy()
    P = hp.export_params()
    ref = world["opipe"].render(P, o_np, d_np, np.full(32, cam0, np.int32), scene.make_noise(32, False, fin), LOG2_T, block=0, early_T=hp.cfg.early_T,
                                scale_factor=hp.cfg.scale_factor)
    assert ref["acc"].min() > 0.5
    np.testing.assert_allclose(rgb[y, x0:x0 + 32].cpu().numpy(), ref["rgb"], atol=1e-3)
    np.testing.assert_allclose(acc[y, x0:x0 + 32].cpu().numpy(), ref["acc"], atol=1e-4)

    # another camera through the same graph
    cam1 = 74
    img1 = tr.render_camera(cam1, H, W)
    rays1 = ops.generate_rays(c2w, intr, cam0=cam1, first_pixel=0, width=W, n=H * W, want_extras=False)
    O1, D1 = rays1["origins"].view(H, W, 3), rays1["directions"].view(H, W, 3)
    ref1 = direct(O1[y, x0:x0 + 32], D1[y, x0:x0 + 32], cam1)
    assert float((img1["rgb"][y, x0:x0 + 32] - ref1["rgb"]).abs().max()) <= 2e-6
    assert float((img1["rgb"] - rgb).abs().mean()) > 1e-3  # and it is another picture
    del tr
    torch.cuda.empty_cache()

    # the renderer's DEFAULT form (HotPathConfig.render_colour_x2, ABI 108): the colour head's layer 2 on one f16 activation operand.  Nothing the density
    # depends on moves -- accumulation and depth are the x3 frame's bits --, rgb stays within a third of north_star's bar of the x3 frame (here, on the
    # random-init scene: within 2e-5) and within the bar of the oracle on the same 32 pixels; the same bits when rendered again
    assert hp.cfg.render_colour_x2
    tr2 = rmod.TileRenderer(hp, c2w, intr, block=0, fineness=fin, use_graph=True)
    assert tr2.colour_x2
    img2 = tr2.render_camera(cam0, H, W)
    assert torch.equal(img2["accumulation"], first["accumulation"]) and torch.equal(img2["depth"], first["depth"])
    d2 = (img2["rgb"] - first["rgb"]).abs()
    assert 0.0 < float(d2.max()) <= 2e-5, float(d2.max())
    np.testing.assert_allclose(img2["rgb"][y, x0:x0 + 32].cpu().numpy(), ref["rgb"], atol=1e-3)
    keep = img2["rgb"].clone()
    assert torch.equal(tr2.render_camera(cam0, H, W)["rgb"], keep)
    del tr2
    torch.cuda.empty_cache()

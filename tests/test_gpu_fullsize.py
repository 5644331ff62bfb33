"""The hot path at BASELINE.json's FULL sizes -- 4096 rays x 192 samples (786,432 points), T = 2^21 slots per level -- where
the CPU oracle cannot run the whole batch in seconds: size-independent properties (partition of unity, conservation of
the gradient, monotone offsets and depths, bounded compositing) plus the oracle on a SUBSET of the very same full-size
run (random points / a contiguous run of rays), bit for bit.  All through the C ABI (ops.* -> libgfnerf_amd.so)."""
import numpy as np
import pytest

from helpers import rays_for, small_scene

pytestmark = pytest.mark.gpu

R_FULL, S_FULL, LOG2_T = 4096, 192, 21
N_FULL = R_FULL * S_FULL


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


def _t(a, dev):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _ray_like_points(rng, n_vol):
    """786,432 points laid out like a packed batch: 4096 segments through the unit cube, 192 ordered points each, the
    warp volume changing a few times along a ray (anchors in runs, like PersSampler's leaf hops)."""
    a = rng.random((R_FULL, 1, 3), dtype=np.float32)
    b = rng.random((R_FULL, 1, 3), dtype=np.float32)
    s = np.sort(rng.random((R_FULL, S_FULL, 1), dtype=np.float32), axis=1)
    pts = (a + (b - a) * s).reshape(-1, 3).astype(np.float32)
    vol = np.repeat(rng.integers(0, n_vol, (R_FULL, 8)), S_FULL // 8, axis=1).reshape(-1)
    return pts, vol.astype(np.int32)


def test_full_size_lookup_against_oracle_subset_and_partition_of_unity(oracle, ops, dev):
    """Hash3DAnchored forward (Hash3DAnchored_cuda.cu:11-79) at N = 786,432, T = 2^21, two tables with their own primes:
    the f32-master lookup, the f16-shadow lookup and the level-major lookup agree bit for bit over the whole batch; 4,000
    points of that batch are checked against the oracle (which reads the same 2^21-slot table); a constant table comes
    back constant (the eight trilinear weights sum to one)."""
    import torch

    from gfnerf_amd import scene

    rng = np.random.default_rng(2021)
    V = 211
    pool = scene.make_feat_pool(LOG2_T, seed=2)
    prim, prim_b = scene.make_primes(V, seed=3), scene.make_primes(V, seed=4)
    bias = np.zeros((16 * V, 3), np.float32)
    pts, vol = _ray_like_points(rng, V)
    pool_t, prim_t, bias_t, pts_t, vol_t = (_t(x, dev) for x in (pool, prim, bias, pts, vol))
    full = ops.hash_fwd(pool_t, prim_t, bias_t, pts_t, vol_t, V, LOG2_T)
    assert full.shape == (N_FULL, 32)
    t16 = ops.table_to_f16(pool_t)
    assert torch.equal(ops.hash_fwd_f16(t16, prim_t, None, pts_t, vol_t, V, LOG2_T), full)
    assert torch.equal(ops.hash_fwd_lm(t16, prim_t, None, pts_t, vol_t, V, LOG2_T), full)
    # the oracle on a subset of the same batch (first / last points and 4,000 random ones)
    idx = np.unique(np.concatenate([[0, 1, N_FULL - 2, N_FULL - 1], rng.integers(0, N_FULL, 4000)]))
    want = oracle.hash_fwd(pool, prim, bias, pts[idx], vol[idx], V, LOG2_T)
    got = full[torch.from_numpy(idx).to(dev)].cpu().numpy()
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"
    # global + block table (own primes): level-major, planar, accumulated == direct dual lookup
    tb = (torch.randn((16 << LOG2_T, 2), device=dev) * 1e-3).to(torch.float16)
    pb_t = _t(prim_b, dev)
    dual = ops.hash_fwd_f16(t16, prim_t, None, pts_t, vol_t, V, LOG2_T, table2_f16=tb, prim_pool2=pb_t)
    planar = ops.hash_fwd_lm(t16, prim_t, None, pts_t, vol_t, V, LOG2_T, planar=True)
    ops.hash_fwd_lm(tb, pb_t, None, pts_t, vol_t, V, LOG2_T, planar=True, out=planar, accumulate=True)
    assert torch.equal(planar.permute(1, 0, 2).reshape(N_FULL, 32), dual)
    del dual, planar, tb, full
    # partition of unity: sum of the trilinear weights is 1 up to f32 rounding -> a constant table is reproduced (1 f16 ulp)
    const = torch.full((16 << LOG2_T, 2), 0.25, dtype=torch.float16, device=dev)
    out = ops.hash_fwd_lm(const, prim_t, None, pts_t, vol_t, V, LOG2_T)
    assert float((out - 0.25).abs().max()) <= 2.0 ** -12


def test_full_size_gradient_is_conserved_and_matches_oracle_subset(oracle, ops, dev):
    """Hash3DAnchoredFunction::backward (:81-155,198-239) at T = 2^21: what the scatter puts into the pool is what came in
    (every sample spreads g over weights that sum to one; channel k lives at odd/even elements), nothing lands beyond the
    17T halves the level-overlap quirk can reach, and the gradient of a 3,000-point subset equals the oracle's."""
    import torch

    from gfnerf_amd import scene

    rng = np.random.default_rng(77)
    V = 211
    T = 1 << LOG2_T
    prim = scene.make_primes(V, seed=3)
    bias = np.zeros((16 * V, 3), np.float32)
    pts, vol = _ray_like_points(rng, V)
    g = (rng.standard_normal((N_FULL, 32)) * 1e-3).astype(np.float32)
    g[::7] = 0  # the skip-if-zero branch (:149)
    prim_t, bias_t = _t(prim, dev), _t(bias, dev)
    pool = ops.hash_bwd(prim_t, bias_t, _t(pts, dev), _t(vol, dev), _t(g, dev), V, LOG2_T)
    flat = pool.reshape(-1).double()
    assert bool((flat[17 * T:] == 0).all())
    # conservation per channel: contributions are quantised to f16 one by one (:150), relative 2^-11 each, random sign
    g16 = (g.astype(np.float64) * 128).astype(np.float16).astype(np.float64) / 128  # :219 grad_in = (grad * 128).half()
    for ch in range(2):
        want = g16[:, ch::2].sum()
        got = float(flat[ch::2].sum())
        # ~86 M contributions of rms 2e-4, each rounded by <= 2^-11 relative with random sign: sigma ~ 5e-4 on a sum of O(1)
        assert abs(got - want) <= 1e-2, (ch, got, want)
    # the oracle on a subset at the full table size
    idx = rng.integers(0, N_FULL, 3000)
    want = oracle.hash_bwd(prim, bias, pts[idx], vol[idx], g[idx], V, LOG2_T)
    got = ops.hash_bwd(prim_t, bias_t, _t(pts[idx], dev), _t(vol[idx], dev), _t(g[idx], dev), V, LOG2_T).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-9)
    # f16 pool (the reference's, x128): same mass up to the f16 accumulation
    pool16 = ops.hash_bwd(prim_t, bias_t, _t(pts, dev), _t(vol, dev), _t(g, dev), V, LOG2_T, grad_f16=True)
    mass32, mass16 = float(flat.abs().sum()), float(pool16.reshape(-1).double().abs().sum()) / 128
    assert abs(mass16 - mass32) <= 2e-2 * mass32


def test_full_size_sampler_properties_and_oracle_on_a_run_of_rays(oracle, ops, dev):
    """PersSampler GetSamples (PersSampler_cuda.cu:321-477) on 4096 rays with ~192 samples each: CSR offsets are the
    exclusive prefix sum of the per-ray counts, depths grow along every ray, anchors name existing warp transforms, and
    48 consecutive rays of the batch equal the oracle's trace + march bit for bit (the march reads the noise window
    rays_noise[ray + k], :204,270, so the subset run gets the window shifted by its first ray)."""
    from gfnerf_amd import scene

    sc = small_scene()
    o, d_raw, _ = rays_for(sc, R_FULL, seed=5)
    fineness = 1.0
    noise = scene.make_noise(R_FULL, True, fineness, seed=3)
    cn = ops.tree_compact(_t(sc["nodes"], dev))
    d_t = ops.normalize_dirs(_t(d_raw, dev))
    g = ops.sample_rays(_t(o, dev), d_t, _t(noise, dev), cn, _t(sc["transes"], dev), 1.0 / 256, True, want_world=True, table_space=False)
    po = g["pts_offsets"].cpu().numpy().astype(np.int64)
    N = int(po[-1])
    assert int(g["overflow"].item()) == 0
    cnt = np.diff(po)
    assert po[0] == 0 and (cnt >= 0).all() and cnt.max() <= 1024
    assert N >= R_FULL * 150, f"mean samples per ray {N / R_FULL:.1f}: raise the sample budget of this test"
    ts, dists = g["ts"].cpu().numpy()[:N], g["dists"].cpu().numpy()[:N]
    ray_id = g["ray_id"].cpu().numpy()[:N]
    assert np.array_equal(ray_id, np.repeat(np.arange(R_FULL), cnt))
    same_ray = ray_id[1:] == ray_id[:-1]
    assert (np.diff(ts)[same_ray] > 0).all() and (dists > 0).all()
    anc = g["anchors"].cpu().numpy()[:N]
    n_trans, n_nodes = sc["transes"].size // 576, sc["nodes"].size // 128
    assert anc[:, 0].min() >= 0 and anc[:, 0].max() < n_trans and anc[:, 1].min() >= 0 and anc[:, 1].max() < n_nodes
    # the oracle on rays [r0, r0 + 48)
    r0, k = 1777, 48
    nodes = sc["nodes"].view(oracle.TREE_NODE_DTYPE)
    d = oracle.normalize_dirs(d_raw[r0:r0 + k])
    assert np.array_equal(d_t.cpu().numpy()[r0:r0 + k], d)
    off, hi, nf = oracle.trace(o[r0:r0 + k], d, nodes)
    m = oracle.march(o[r0:r0 + k], d, noise[r0:r0 + 1024 + k + 10], 1.0 / 256, True, off, hi, nf, nodes, sc["transes"].view(oracle.TRANS_INFO_DTYPE))
    a, b = int(po[r0]), int(po[r0 + k])
    assert np.array_equal(po[r0:r0 + k + 1] - a, m["pts_offsets"])
    assert np.array_equal(anc[a:b].astype(np.int64), m["anchors"])
    for key in ("ts", "dists", "warped_pts", "world_pts"):
        assert np.array_equal(g[key].cpu().numpy()[a:b], m[key]), key
    assert np.array_equal(g["first_oct_dis"].cpu().numpy()[r0:r0 + k], m["first_oct_dis"])


@pytest.fixture(scope="module")
def full_engine():
    import torch

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import hotpath, ops, scene

    sc = small_scene()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    o, d, cam = rays_for(sc, R_FULL, seed=5)
    rng = np.random.default_rng(9)
    batch = dict(o=t(o), d=t(d), cam=t(cam), gt=t(rng.random((R_FULL, 3)).astype(np.float32)),
                 noise=t(scene.make_noise(R_FULL, True, 1.0, seed=3)), s3=ops.s3im_index(R_FULL, 10, device="cuda"))

    def make(**kw):
        cfg = hotpath.HotPathConfig(log2_T=LOG2_T, **kw)
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=1, n_blocks=1)
        hp.blocks[0].uniform_(-1e-3, 1e-3, generator=torch.Generator(device="cuda").manual_seed(5))
        hp.refresh_shadows()
        return hp

    return make, batch, sc


def test_full_size_focal_step_record_streams_against_atomics(full_engine):
    """One focal-stage forward + backward at 4096 rays / T = 2^21 (all 1088 slot ranges of the record streams in use).  The
    f32 gradient pool (exact sum of the f16-quantised contributions) is the yardstick: the reference's f16 pool (x128) as
    accumulated by scattered atomics and as accumulated in LDS from the record streams equals it up to f16 rounding: the
    record form sums exactly in fixed point and rounds once; with lds_accumulate_exact=False it adds every contribution on
    its own in f16 like the reference's atomicAdd(half2) does (Hash3DAnchored_cuda.cu:151) and is the least accurate of the
    three; the atomic form merges runs in f32 first.  Every record was counted."""
    make, B, _ = full_engine
    pools, losses = {}, {}
    for name, kw in (("exact", dict(grad_f16=False)), ("atomic", dict(lds_accumulate=False)), ("records", dict(lds_accumulate=True)),
                     ("records_f16", dict(lds_accumulate=True, lds_accumulate_exact=False))):
        hp = make(**kw)
        if name.startswith("records"):
            import torch
            hp.record_counts_seen = torch.zeros_like(hp._record_buffers()[1])
        loss, aux = hp.forward_backward(B["o"], B["d"], B["cam"], B["gt"], B["noise"], stage="focal", block=0, s3im_index=B["s3"])
        pools[name] = hp.table_grad.double().cpu().numpy() * (128.0 if name == "exact" else 1.0)
        losses[name] = float(loss.item())
        if name.startswith("records"):
            from gfnerf_amd import _lib
            n_ranges = _lib.lib().gfn_grad_record_ranges(LOG2_T)
            counts = hp.record_counts_seen.cpu().numpy().astype(np.int64)[: n_ranges * 256]
            cap = hp._records[0].numel() // (n_ranges * 256)
            assert n_ranges == 1088 and counts.shape == (n_ranges * 256,) and counts.max() <= cap and counts.sum() > 5_000_000
            assert (counts.reshape(n_ranges, 256).sum(1) > 0).all()  # every range of every level received records
        del hp
    assert np.isfinite(list(losses.values())).all()
    assert max(losses.values()) - min(losses.values()) <= 1e-6 * losses["exact"]  # same forward; the loss sums by atomics
    ex = pools["exact"]
    scale, mass = np.abs(ex).max(), np.abs(ex).sum()
    assert scale > 0 and (ex != 0).sum() > 5_000_000
    for name, l1_tol, max_tol in (("atomic", 3e-4, 3e-2), ("records", 3e-4, 3e-2), ("records_f16", 3e-3, 0.2)):
        err = np.abs(pools[name] - ex)
        assert err.sum() <= l1_tol * mass, (name, err.sum() / mass)
        assert err.max() <= max_tol * scale, (name, err.max() / scale)
        assert abs(pools[name].sum() - ex.sum()) <= 5e-4 * mass, name  # no systematic loss of gradient mass
        assert ((pools[name] != 0) != (ex != 0)).mean() < 1e-4, name


def test_full_size_render_is_bounded_and_matches_oracle_on_a_run_of_rays(full_engine):
    """Forward of 4096 rays x ~190 samples through global + block table, decoder and compositing: accumulation and colour
    stay in [0, 1], transmittance ordering gives depth inside the sampled range, and rays [600, 632) equal the CPU
    pipeline oracle within the north-star tolerance (1e-3 abs)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pipeline as opipe

    make, B, sc = full_engine
    hp = make(early_T=0.0)
    out = hp.render(B["o"], B["d"], B["cam"], B["noise"], block=0)
    rgb, acc = out["rgb"].cpu().numpy(), out["accumulation"].cpu().numpy().reshape(-1)
    assert np.isfinite(rgb).all() and rgb.min() >= 0 and rgb.max() <= 1 + 1e-6
    assert acc.min() >= 0 and acc.max() <= 1 + 1e-5
    r0, k = 600, 32
    P = hp.export_params()
    noise = B["noise"].cpu().numpy()
    ref = opipe.render(P, B["o"].cpu().numpy()[r0:r0 + k], B["d"].cpu().numpy()[r0:r0 + k], B["cam"].cpu().numpy()[r0:r0 + k],
                       noise[r0:r0 + 1024 + k + 10], LOG2_T, block=0, early_T=0.0, scale_factor=hp.cfg.scale_factor)
    np.testing.assert_allclose(rgb[r0:r0 + k], ref["rgb"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(acc[r0:r0 + k], np.asarray(ref["acc"]).reshape(-1), atol=1e-3, rtol=0)
    assert int(ref["samples"]["pts_offsets"][-1]) > 100 * k

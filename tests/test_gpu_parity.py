"""GPU parity tests: every HIP kernel, called through the C ABI (gfnerf_amd.ops -> ctypes ->
libgfnerf_amd.so), against the CPU oracle on the same seeded inputs, and against the golden
vectors captured from the reference.  Run with `-m gpu` on an MI355X.

Bars (SURVEY 8(d) parity gates): octree leaf lists, sample counts, anchors and hash
features BIT-EXACT; f32 sampler outputs bit-exact (same op order, no FMA contraction);
MLP / compositing within 1e-5 relative; end-to-end rgb / sigma within 1e-3 abs.
"""
import numpy as np
import pytest

from helpers import rays_for, small_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch

    import gfnerf_amd
    from gfnerf_amd import _lib

    _lib.require_device()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    import gfnerf_amd
    from gfnerf_amd import ops as _ops

    return _ops


@pytest.fixture
def set_march_lanes(monkeypatch):
    """The samplers' lanes per ray for one test: every ops.sample_rays / ops.sample_rays_record call the test makes without launch options of its own
    carries ops.launch_opts(march_lanes=lanes) (the policy is an argument of the call since ABI 107: nothing process-wide is set or put back)."""
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import ops as _ops

    def _set(lanes):
        for name in ("sample_rays", "sample_rays_record"):
            fn = getattr(_ops, name)

            def call(*a, _fn=fn, **k):
                k.setdefault("opts", _ops.launch_opts(march_lanes=int(lanes)))
                return _fn(*a, **k)

            monkeypatch.setattr(_ops, name, call)

    return _set


def _t(a, dev):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------------------ hash
@pytest.mark.parametrize("log2_T,n_vol,n_pts,i64", [(14, 7, 4097, True), (19, 50, 20000, False), (4, 1, 64, False)])
def test_hash_forward_bit_exact(oracle, ops, dev, log2_T, n_vol, n_pts, i64):
    from gfnerf_amd import scene

    rng = np.random.default_rng(log2_T)
    pool = scene.make_feat_pool(log2_T, seed=2)
    prim = scene.make_primes(n_vol, seed=3)
    bias = np.zeros((16 * n_vol, 3), np.float32)
    pts = rng.random((n_pts, 3), dtype=np.float32)
    pts[:8] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0.999999, 0, 1], [1, 0, 0], [0, 1, 0], [0.25, 0.75, 0.125], [1e-7, 1e-7, 1e-7]]
    anchors = rng.integers(0, n_vol, n_pts)
    want = oracle.hash_fwd(pool, prim, bias, pts, anchors, n_vol, log2_T)
    got = ops.hash_fwd(_t(pool, dev), _t(prim, dev), _t(bias, dev), _t(pts, dev),
                       _t(anchors.astype(np.int64 if i64 else np.int32), dev), n_vol, log2_T).cpu().numpy()
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


def test_hash_forward_random_bias_and_empty(oracle, ops, dev):
    """rand_bias branch (Hash3DAnchored.cpp:57-59) + n = 0."""
    import torch

    from gfnerf_amd import scene

    rng = np.random.default_rng(5)
    log2_T, n_vol, n = 12, 3, 3000
    pool = scene.make_feat_pool(log2_T, seed=9)
    prim = scene.make_primes(n_vol, seed=4)
    bias = (rng.random((16 * n_vol, 3), dtype=np.float32) * 1000 + 100).astype(np.float32)
    pts = rng.random((n, 3), dtype=np.float32)
    anchors = rng.integers(0, n_vol, n).astype(np.int32)
    want = oracle.hash_fwd(pool, prim, bias, pts, anchors, n_vol, log2_T)
    args = (_t(pool, dev), _t(prim, dev), _t(bias, dev))
    got = ops.hash_fwd(*args, _t(pts, dev), _t(anchors, dev), n_vol, log2_T).cpu().numpy()
    assert np.array_equal(got, want)
    empty = ops.hash_fwd(*args, torch.empty((0, 3), device=dev), torch.empty(0, dtype=torch.int32, device=dev), n_vol, log2_T)
    assert empty.shape == (0, 32)


def test_hash_forward_respects_device_count(oracle, ops, dev):
    """n_dev < n_max: rows beyond the live count are left untouched."""
    import torch

    from gfnerf_amd import scene

    log2_T, n_vol, n = 10, 2, 1000
    rng = np.random.default_rng(6)
    pool, prim = scene.make_feat_pool(log2_T), scene.make_primes(n_vol)
    bias = np.zeros((16 * n_vol, 3), np.float32)
    pts = rng.random((n, 3), dtype=np.float32)
    anchors = rng.integers(0, n_vol, n).astype(np.int32)
    out = torch.full((n, 32), -7.0, device=dev)
    n_dev = torch.tensor([600], dtype=torch.int32, device=dev)
    ops.hash_fwd(_t(pool, dev), _t(prim, dev), _t(bias, dev), _t(pts, dev), _t(anchors, dev), n_vol, log2_T, out=out, n_dev=n_dev)
    want = oracle.hash_fwd(pool, prim, bias, pts[:600], anchors[:600], n_vol, log2_T)
    got = out.cpu().numpy()
    assert np.array_equal(got[:600], want) and np.all(got[600:] == -7.0)


def test_hash_backward(oracle, ops, dev):
    from gfnerf_amd import scene

    rng = np.random.default_rng(8)
    log2_T, n_vol, n = 12, 5, 6000
    prim = scene.make_primes(n_vol, seed=3)
    bias = np.zeros((16 * n_vol, 3), np.float32)
    pts = rng.random((n, 3), dtype=np.float32)
    anchors = rng.integers(0, n_vol, n).astype(np.int32)
    g = (rng.standard_normal((n, 32)) * 1e-3).astype(np.float32)
    g[:50] = 0  # exercises the skip-if-zero branch (Hash3DAnchored_cuda.cu:149)
    want = oracle.hash_bwd(prim, bias, pts, anchors, g, n_vol, log2_T)
    got = ops.hash_bwd(_t(prim, dev), _t(bias, dev), _t(pts, dev), _t(anchors, dev), _t(g, dev), n_vol, log2_T).cpu().numpy()
    assert (want != 0).sum() > 1000
    # only slots the level-overlap quirk can reach are touched: elements [0, 17T)
    T = 1 << log2_T
    assert np.all(got.reshape(-1)[17 * T:] == 0)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-9)
    # linearity in grad_out (a size-independent property): bwd(2g) == 2 bwd(g) up to f16 rounding of g*128
    got2 = ops.hash_bwd(_t(prim, dev), _t(bias, dev), _t(pts, dev), _t(anchors, dev), _t(2 * g, dev), n_vol, log2_T).cpu().numpy()
    np.testing.assert_allclose(got2, 2 * got, rtol=2e-3, atol=1e-7)


def test_hash_forward_f16_shadow_and_fused_dual(oracle, ops, dev):
    """f16 shadow tables give the same bits as the f32 master path; the fused dual lookup equals feat(A)+feat(B)."""
    from gfnerf_amd import scene

    rng = np.random.default_rng(21)
    log2_T, n_vol, n = 15, 11, 9000
    pa, pb = scene.make_feat_pool(log2_T, seed=2), scene.make_feat_pool(log2_T, seed=5, scale=3e-3)
    prim = scene.make_primes(n_vol, seed=3)
    bias = np.zeros((16 * n_vol, 3), np.float32)
    pts = rng.random((n, 3), dtype=np.float32)
    anchors3 = np.stack([np.sort(rng.integers(0, n_vol, n)), rng.integers(0, 99, n), rng.integers(0, 9, n)], 1).astype(np.int32)
    wa = oracle.hash_fwd(pa, prim, bias, pts, anchors3[:, 0], n_vol, log2_T)
    wb = oracle.hash_fwd(pb, prim, bias, pts, anchors3[:, 0], n_vol, log2_T)
    a16, b16 = ops.table_to_f16(_t(pa, dev)), ops.table_to_f16(_t(pb, dev))
    assert np.array_equal(a16.cpu().numpy(), pa.astype(np.float16))
    common = (_t(prim, dev), None, _t(pts, dev), _t(anchors3, dev), n_vol, log2_T)  # bias None == zero pool; packed [n,3] anchors
    got = ops.hash_fwd_f16(a16, *common).cpu().numpy()
    assert np.array_equal(got, wa)
    dual = ops.hash_fwd_f16(a16, *common, table2_f16=b16).cpu().numpy()
    assert np.array_equal(dual, wa + wb)
    # the residual table hashed with its OWN primes (what add_table does in the reference)
    prim2 = scene.make_primes(n_vol, seed=77)
    wb2 = oracle.hash_fwd(pb, prim2, bias, pts, anchors3[:, 0], n_vol, log2_T)
    dual2 = ops.hash_fwd_f16(a16, *common, table2_f16=b16, prim_pool2=_t(prim2, dev)).cpu().numpy()
    assert np.array_equal(dual2, wa + wb2) and not np.array_equal(dual2, dual)
    # unfused two-call form of the boundary (accumulate) agrees bit for bit
    two = ops.hash_fwd(_t(pa, dev), _t(prim, dev), _t(bias, dev), _t(pts, dev), _t(anchors3, dev), n_vol, log2_T)
    ops.hash_fwd(_t(pb, dev), _t(prim, dev), _t(bias, dev), _t(pts, dev), _t(anchors3, dev), n_vol, log2_T, out=two, accumulate=True)
    assert np.array_equal(two.cpu().numpy(), dual)


def test_hash_backward_f16_pool_and_runs(oracle, ops, dev):
    """Packed-f16 gradient pool (the reference's own accumulation, x128-scaled) and the run-length path
    (consecutive samples in one cell), against the oracle's exact sum of the same quantised contributions."""
    from gfnerf_amd import scene

    rng = np.random.default_rng(9)
    log2_T, n_vol, n = 13, 3, 8192
    prim = scene.make_primes(n_vol, seed=3)
    # a slow march through table space: long runs on the coarse levels, none on the fine ones
    t = np.linspace(0, 1, n, dtype=np.float32)[:, None]
    pts = (np.array([[0.1, 0.2, 0.3]], np.float32) + t * np.array([[0.7, 0.5, 0.6]], np.float32)).astype(np.float32)
    anchors = (np.arange(n) // 3000).astype(np.int32)
    g = (rng.standard_normal((n, 32)) * 1e-3).astype(np.float32)
    want = oracle.hash_bwd(prim, None if False else np.zeros((16 * n_vol, 3), np.float32), pts, anchors, g, n_vol, log2_T)
    args = (_t(prim, dev), None, _t(pts, dev), _t(anchors, dev), _t(g, dev), n_vol, log2_T)
    got32 = ops.hash_bwd(*args).cpu().numpy()
    np.testing.assert_allclose(got32, want, rtol=2e-5, atol=1e-9)
    got16 = ops.hash_bwd(*args, grad_f16=True)
    assert got16.dtype.is_floating_point and got16.element_size() == 2
    got16 = got16.float().cpu().numpy() / 128.0
    scale = np.abs(want).max()
    # f16 running sums: relative error ~2^-11 per add; coarse-level slots take tens of adds
    np.testing.assert_allclose(got16, want, rtol=2e-2, atol=2e-3 * scale)
    assert np.abs(got16 - want).max() < 4e-3 * scale
    assert np.all(got16[want == 0] == 0)  # untouched slots stay exactly zero (tiny sums may underflow f16 to zero)


def test_adam_table_f16_grad_and_shadow(oracle, ops, dev):
    import torch

    rng = np.random.default_rng(3)
    n = 5000
    p0 = rng.standard_normal(n).astype(np.float32) * 1e-2
    g = (rng.standard_normal(n) * 1e-3).astype(np.float32)
    g16 = (g * 128).astype(np.float16)
    p_ref, m_ref, v_ref = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    oracle.adam_step(p_ref, (g16.astype(np.float32) / 128).astype(np.float32), m_ref, v_ref, 1)
    p = _t(p0.copy(), dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    shadow = torch.zeros(n, dtype=torch.float16, device=dev)
    ops.adam_step_table(p, _t(g16, dev), m, v, 1, shadow_f16=shadow, n=4000)
    got = p.cpu().numpy()
    np.testing.assert_allclose(got[:4000], p_ref[:4000], rtol=1e-6, atol=1e-8)
    assert np.array_equal(got[4000:], p0[4000:])  # elements beyond n untouched
    assert np.array_equal(shadow.cpu().numpy()[:4000], got[:4000].astype(np.float16))


# --------------------------------------------------------------------------- sampler
def _sample_both(oracle, ops, dev, n_rays=700, train=True, fineness=6.0, seed=0):
    from gfnerf_amd import scene

    sc = small_scene()
    o, d_raw, cam = rays_for(sc, n_rays, seed)
    d_raw = d_raw * np.linspace(0.5, 2.0, n_rays, dtype=np.float32)[:, None]  # un-normalised input
    nodes = sc["nodes"].view(oracle.TREE_NODE_DTYPE)
    tr = sc["transes"].view(oracle.TRANS_INFO_DTYPE)
    noise = scene.make_noise(n_rays, train, fineness, seed=1)
    d = oracle.normalize_dirs(d_raw)
    off, hi, nf = oracle.trace(o, d, nodes)
    m = oracle.march(o, d, noise, 1.0 / 256, True, off, hi, nf, nodes, tr)
    nodes_t, tr_t = _t(sc["nodes"], dev), _t(sc["transes"], dev)
    o_t = _t(o, dev)
    d_t = ops.normalize_dirs(_t(d_raw, dev))
    g_off, g_hi, g_nf, g_ovf = ops.trace(o_t, d_t, nodes_t)
    g = ops.march(o_t, d_t, _t(noise, dev), 1.0 / 256, True, g_off, g_hi, g_nf, nodes_t, tr_t)
    return sc, dict(o=o, d=d, off=off, hi=hi, nf=nf, m=m, cam=cam), dict(d=d_t, off=g_off, hi=g_hi, nf=g_nf, ovf=g_ovf, m=g, nodes=nodes_t)


def test_trace_bit_exact(oracle, ops, dev):
    sc, ref, gpu = _sample_both(oracle, ops, dev)
    assert np.array_equal(gpu["d"].cpu().numpy(), ref["d"])
    off = gpu["off"].cpu().numpy()
    assert np.array_equal(off.astype(np.int64), ref["off"])
    H = int(off[-1])
    assert H > 1000
    assert np.array_equal(gpu["hi"].cpu().numpy()[:H].astype(np.int64), ref["hi"])
    assert np.array_equal(gpu["nf"].cpu().numpy()[:H], ref["nf"])
    assert int(gpu["ovf"].item()) == 0


def test_trace_max_hits_cap_and_axis_parallel_rays(oracle, ops, dev):
    """max_oct_intersect_per_ray cap (PersSampler_cuda.cu:89) and the |d_i|<1e-6 slab case (:31-38)."""
    sc = small_scene()
    nodes = sc["nodes"].view(oracle.TREE_NODE_DTYPE)
    o = np.array([[0.3, 0.2, 11.0], [5.0, -3.0, 12.0], [1.0, 1.0, 300.0], [0.1, 0.1, 10.0]], np.float32)
    d = np.array([[0, 0, -1.0], [1.0, 0, 0], [0, 0, 1.0], [1e-7, -1e-7, -1.0]], np.float32)
    d = oracle.normalize_dirs(d)
    for max_hits in (3, 1024):
        off, hi, nf = oracle.trace(o, d, nodes, max_hits=max_hits)
        g_off, g_hi, g_nf, _ = ops.trace(_t(o, dev), _t(d, dev), _t(sc["nodes"], dev), max_hits=max_hits)
        H = int(off[-1])
        assert np.array_equal(g_off.cpu().numpy().astype(np.int64), off)
        assert np.array_equal(g_hi.cpu().numpy()[:H].astype(np.int64), hi)
        assert np.array_equal(g_nf.cpu().numpy()[:H], nf)
        assert np.diff(off).max() <= max_hits


@pytest.mark.parametrize("train,fineness", [(True, 6.0), (False, 16.0), (False, 0.5)])
def test_march_bit_exact(oracle, ops, dev, train, fineness):
    sc, ref, gpu = _sample_both(oracle, ops, dev, n_rays=300, train=train, fineness=fineness)
    m, g = ref["m"], gpu["m"]
    po = g["pts_offsets"].cpu().numpy()
    assert np.array_equal(po.astype(np.int64), m["pts_offsets"])
    N = int(po[-1])
    assert N > 3000
    if fineness == 0.5:
        assert np.diff(po).max() == 1024  # the MAX_SAMPLE_PER_RAY cap is exercised
    assert np.array_equal(g["anchors"].cpu().numpy()[:N].astype(np.int64), m["anchors"])
    for k in ("ts", "dists", "warped_pts", "world_pts", "dirs"):
        a, b = g[k].cpu().numpy()[:N], m[k]
        assert np.array_equal(a, b), f"{k}: max abs diff {np.abs(a - b).max()}"
    assert np.array_equal(g["first_oct_dis"].cpu().numpy(), m["first_oct_dis"])
    rid = g["ray_id"].cpu().numpy()[:N]
    assert np.array_equal(rid, np.repeat(np.arange(len(po) - 1), np.diff(po)))
    assert int(g["overflow"].item()) == 0


@pytest.mark.parametrize("lanes,with_rec", [(16, True), (16, False), (4, True), (4, False)])
@pytest.mark.parametrize("train,fineness,n_rays", [(True, 6.0, 333), (False, 0.5, 70), (False, 16.0, 4)])
def test_fast_sampler_bit_exact(oracle, ops, dev, set_march_lanes, train, fineness, n_rays, lanes, with_rec):
    """gfn_sample_rays (compact nodes, one-pass dense trace, a group of lanes per ray with DPP reductions) produces the same bits as
    the oracle -- and therefore as the gfn_trace + gfn_march boundary functions -- in each of its forms: a 16-lane row per ray (what
    batches below 32 Ki rays get) or a quad of lanes (render tiles), gfn_launch_opts.march_lanes forcing either on these rays; the counting march
    recording (t, dist, leaf, warped point) for the sample-parallel fill, or, without the record buffer, a second march writing the samples."""
    import torch

    from gfnerf_amd import scene

    set_march_lanes(lanes)
    sc, ref, gpu = _sample_both(oracle, ops, dev, n_rays=n_rays, train=train, fineness=fineness)
    m = ref["m"]
    noise = scene.make_noise(n_rays, train, fineness, seed=1)
    cn = ops.tree_compact(_t(sc["nodes"], dev))
    ws = None
    if not with_rec:  # a workspace without the record buffer: count pass, scan, fill pass (both marches)
        ws = dict(hit_counts=torch.empty(n_rays, dtype=torch.int32, device=dev), hit_idx=torch.empty(n_rays * ops.MAX_OCT_PER_RAY, dtype=torch.int32, device=dev),
                  hit_nf=torch.empty((n_rays * ops.MAX_OCT_PER_RAY, 2), dtype=torch.float32, device=dev))
    for table_space in (False, True):
        g = ops.sample_rays(_t(ref["o"], dev), _t(ref["d"], dev), _t(noise, dev), cn, _t(sc["transes"], dev), 1.0 / 256, True,
                            want_world=True, want_dirs=True, table_space=table_space, workspace=ws)
        assert ("march_rec" in g["workspace"]) == with_rec
        po = g["pts_offsets"].cpu().numpy()
        assert np.array_equal(po.astype(np.int64), m["pts_offsets"])
        N = int(po[-1])
        assert np.array_equal(g["workspace"]["hit_counts"].cpu().numpy()[:n_rays].astype(np.int64), np.diff(ref["off"]))
        assert np.array_equal(g["anchors"].cpu().numpy()[:N].astype(np.int64), m["anchors"])
        want_w = m["warped_pts"] if not table_space else ((m["warped_pts"] + np.float32(1.5)) * (np.float32(1) / np.float32(3))).astype(np.float32)
        for k, want in (("ts", m["ts"]), ("dists", m["dists"]), ("warped_pts", want_w), ("world_pts", m["world_pts"]), ("dirs", m["dirs"])):
            a = g[k].cpu().numpy()[:N]
            assert np.array_equal(a, want), f"{k} (table_space={table_space}): max abs diff {np.abs(a - want).max()}"
        assert np.array_equal(g["first_oct_dis"].cpu().numpy(), m["first_oct_dis"])
        assert np.array_equal(g["ray_id"].cpu().numpy()[:N], np.repeat(np.arange(n_rays), np.diff(po)))
        assert int(g["overflow"].item()) == 0


@pytest.mark.parametrize("lanes", [16, 4])
def test_sample_record_holds_the_packed_samples(oracle, ops, dev, set_march_lanes, lanes):
    """gfn_sample_rays_record (traversal + counting march only; what the inference kernel reads, gfn_fused_render_rec): slot k of ray r of
    march_rec is sample pts_offsets[r] + k of gfn_sample_rays, field by field and bit for bit -- {t, dist, leaf, transform, warped point
    (before the table-space map), block} -- and counts / first_oct_dis are the packed form's; slots beyond a ray's count are never read."""
    import torch

    from gfnerf_amd import scene

    set_march_lanes(lanes)
    n_rays = 160
    sc, ref, _ = _sample_both(oracle, ops, dev, n_rays=n_rays, train=True, fineness=6.0)
    noise = _t(scene.make_noise(n_rays, True, 6.0, seed=1), dev)
    cn, tr = ops.tree_compact(_t(sc["nodes"], dev)), _t(sc["transes"], dev)
    o, d = _t(ref["o"], dev), _t(ref["d"], dev)
    packed = ops.sample_rays(o, d, noise, cn, tr, 1.0 / 256, True, table_space=False)
    rec = ops.sample_rays_record(o, d, noise, cn, tr, 1.0 / 256, True)
    po = packed["pts_offsets"].cpu().numpy().astype(np.int64)
    assert np.array_equal(rec["counts"].cpu().numpy(), np.diff(po)) and int(po[-1]) > 3000
    assert torch.equal(rec["first_oct_dis"], packed["first_oct_dis"]) and int(rec["overflow"].item()) == 0
    R = rec["march_rec"].view(n_rays, ops.MAX_SAMPLE_PER_RAY, ops.MARCH_REC_FLOATS).cpu().numpy()
    Ri = R.view(np.int32)
    ts, dists, wp, an = (packed[k].cpu().numpy() for k in ("ts", "dists", "warped_pts", "anchors"))
    for r in range(n_rays):
        a, b = po[r], po[r + 1]
        m = R[r, :b - a]
        assert np.array_equal(m[:, 0], ts[a:b]) and np.array_equal(m[:, 1], dists[a:b]) and np.array_equal(m[:, 4:7], wp[a:b]), r
        mi = Ri[r, :b - a]
        assert np.array_equal(mi[:, 2], an[a:b, 1]) and np.array_equal(mi[:, 3], an[a:b, 0]) and np.array_equal(mi[:, 7], an[a:b, 2]), r
    # the same dict written again in place
    again = ops.sample_rays_record(o, d, noise, cn, tr, 1.0 / 256, True, workspace=rec["workspace"], out=rec)
    assert again is rec and np.array_equal(rec["counts"].cpu().numpy(), np.diff(po))
    empty = torch.empty((0, 3), device=dev)
    assert ops.sample_rays_record(empty, empty, noise, cn, tr, 1.0 / 256, True)["counts"].numel() == 0


def test_traversal_stack_sized_by_tree_depth(oracle, ops, dev):
    """gfn_sample_rays with tree_depth = the blob's real depth (ops.tree_depth: parent links) gives the bits of the default
    (the reference's 24-level bound): 7 entries per level + 2 is what a depth-first traversal can stack.  A bound below the real
    depth is reported through the overflow flag, not by a fault."""
    import torch

    from gfnerf_amd import scene

    sc = small_scene()
    nodes_np = sc["nodes"].view(oracle.TREE_NODE_DTYPE)
    depth_np = 0
    for u in range(nodes_np.shape[0]):  # by hand: walk every node up to the root
        k, v = 0, int(nodes_np["parent"][u])
        while v >= 0:
            k, v = k + 1, int(nodes_np["parent"][v])
        depth_np = max(depth_np, k)
    nodes = _t(sc["nodes"], dev)
    depth = ops.tree_depth(nodes.view(-1, ops.TREE_NODE_BYTES))
    assert depth == depth_np and 2 <= depth <= 24
    n_rays = 700
    o, d, _ = scene.make_rays(sc["c2w"], n_rays, seed=3)
    d = oracle.normalize_dirs(d)
    noise = _t(scene.make_noise(n_rays, True, 6.0, seed=2), dev)
    cn = ops.tree_compact(nodes)
    args = (_t(o, dev), _t(d, dev), noise, cn, _t(sc["transes"], dev), 1.0 / 256, True)
    ref = ops.sample_rays(*args)
    got = ops.sample_rays(*args, tree_depth=depth)
    assert int(ref["overflow"].item()) == 0 and int(got["overflow"].item()) == 0
    N = int(ref["pts_offsets"][-1].item())
    assert N > n_rays and torch.equal(ref["pts_offsets"], got["pts_offsets"])
    for k in ("ts", "dists", "warped_pts", "anchors", "ray_id"):
        assert torch.equal(ref[k][:N], got[k][:N]), k
    assert torch.equal(ref["workspace"]["hit_counts"][:n_rays], got["workspace"]["hit_counts"][:n_rays])
    low = ops.sample_rays(*args, tree_depth=1)
    assert int(low["overflow"].item()) & 2  # stacks of 9 entries cannot hold this tree's traversal: said so
    with pytest.raises(Exception):
        ops.sample_rays(*args, tree_depth=25)


def test_fast_sampler_missing_and_empty(oracle, ops, dev):
    import torch

    from gfnerf_amd import scene

    sc = small_scene()
    cn = ops.tree_compact(_t(sc["nodes"], dev))
    o = np.array([[1000.0, 1000.0, 1000.0], [0.0, 0.0, 12.0], [900.0, 0, 0]], np.float32)
    d = oracle.normalize_dirs(np.array([[1.0, 1.0, 1.0], [0.1, 0.2, -1.0], [1.0, 0, 0]], np.float32))
    noise = scene.make_noise(3, False, 8.0)
    nodes = sc["nodes"].view(oracle.TREE_NODE_DTYPE)
    off, hi, nf = oracle.trace(o, d, nodes)
    m = oracle.march(o, d, noise, 1 / 256, True, off, hi, nf, nodes, sc["transes"].view(oracle.TRANS_INFO_DTYPE))
    g = ops.sample_rays(_t(o, dev), _t(d, dev), _t(noise, dev), cn, _t(sc["transes"], dev), 1 / 256, True, table_space=False)
    assert np.array_equal(g["pts_offsets"].cpu().numpy().astype(np.int64), m["pts_offsets"])
    assert np.array_equal(g["first_oct_dis"].cpu().numpy(), m["first_oct_dis"])
    e = torch.empty((0, 3), device=dev)
    g0 = ops.sample_rays(e, e, _t(noise, dev), cn, _t(sc["transes"], dev), 1 / 256, True)
    assert g0["pts_offsets"].cpu().tolist() == [0]


def test_dense_export_matches_reference_layout(oracle, ops, dev):
    """GetSamples() return list (PersSampler_cuda.cu:437-476): dense [R,1024,.], zero padded, int64 anchors."""
    sc, ref, gpu = _sample_both(oracle, ops, dev, n_rays=64, train=False, fineness=8.0)
    m = ref["m"]
    R = 64
    dense = ops.samples_to_dense(gpu["m"], R)
    world, warped, dirs, dists, ts, anchors, se, fod = [x.cpu().numpy() for x in dense]
    assert world.shape == (R, 1024, 3) and anchors.shape == (R, 1024, 3) and anchors.dtype == np.int64
    assert se.shape == (R, 2) and fod.shape == (R, 1)
    po = m["pts_offsets"]
    for r in range(R):
        n = po[r + 1] - po[r]
        sl = slice(po[r], po[r + 1])
        assert np.array_equal(warped[r, :n], m["warped_pts"][sl]) and np.all(warped[r, n:] == 0)
        assert np.array_equal(world[r, :n], m["world_pts"][sl]) and np.array_equal(dirs[r, :n], m["dirs"][sl])
        assert np.array_equal(dists[r, :n], m["dists"][sl]) and np.all(dists[r, n:] == 0)
        assert np.array_equal(ts[r, :n], m["ts"][sl]) and np.all(ts[r, n:] == 0)
        assert np.array_equal(anchors[r, :n], m["anchors"][sl]) and np.all(anchors[r, n:] == 0)  # zeros, not -1 (:440)
        assert se[r, 0] == po[r] and se[r, 1] == po[r + 1]
    assert np.array_equal(fod[:, 0], m["first_oct_dis"])


def test_sampler_empty_and_missing_rays(oracle, ops, dev):
    """Rays that miss every valid leaf (first_oct_dis = 1e9, zero samples) and an empty batch."""
    import torch

    from gfnerf_amd import scene

    sc = small_scene()
    nodes = sc["nodes"].view(oracle.TREE_NODE_DTYPE)
    tr = sc["transes"].view(oracle.TREE_NODE_DTYPE if False else oracle.TRANS_INFO_DTYPE)
    o = np.array([[1000.0, 1000.0, 1000.0], [0.0, 0.0, 12.0]], np.float32)
    d = oracle.normalize_dirs(np.array([[1.0, 1.0, 1.0], [0.1, 0.2, -1.0]], np.float32))
    noise = scene.make_noise(2, False, 8.0)
    off, hi, nf = oracle.trace(o, d, nodes)
    m = oracle.march(o, d, noise, 1 / 256, True, off, hi, nf, nodes, tr)
    assert off[1] == 0 and m["first_oct_dis"][0] == np.float32(1e9)
    g_off, g_hi, g_nf, _ = ops.trace(_t(o, dev), _t(d, dev), _t(sc["nodes"], dev))
    g = ops.march(_t(o, dev), _t(d, dev), _t(noise, dev), 1 / 256, True, g_off, g_hi, g_nf, _t(sc["nodes"], dev), _t(sc["transes"], dev))
    assert np.array_equal(g["pts_offsets"].cpu().numpy().astype(np.int64), m["pts_offsets"])
    assert np.array_equal(g["first_oct_dis"].cpu().numpy(), m["first_oct_dis"])
    e = torch.empty((0, 3), device=dev)
    e_off, e_hi, e_nf, _ = ops.trace(e, e, _t(sc["nodes"], dev))
    assert e_off.cpu().tolist() == [0]


def test_update_oct_nodes_bit_exact(oracle, ops, dev):
    """K5/K6 + stat arithmetic (PersSampler_cuda.cu:518-665), two consecutive updates."""
    import torch

    sc, ref, gpu = _sample_both(oracle, ops, dev, n_rays=400, train=True, fineness=6.0)
    m = ref["m"]
    N = m["ts"].shape[0]
    rng = np.random.default_rng(3)
    n_nodes = sc["n_nodes"]
    nodes_ref = sc["nodes"].copy().view(oracle.TREE_NODE_DTYPE)  # byte copy (keeps struct padding)
    ws = np.full(n_nodes, 1000, np.int64); as_ = np.full(n_nodes, 1000, np.int64); vc = np.zeros(n_nodes, np.int64)
    nodes_t = _t(sc["nodes"].copy(), dev)
    ws_t, as_t, vc_t = _t(ws, dev), _t(as_, dev), _t(vc, dev)
    for it in range(2):
        # mostly-empty space: tiny weights push stats negative fast (decrement -1 per visit would take 1000
        # iterations, so start some nodes near zero to see invalidation)
        if it == 0:
            low = rng.random(n_nodes) < 0.3
            ws[low] = 0; as_[low] = 0
            ws_t, as_t = _t(ws, dev), _t(as_, dev)
        w = (rng.random(N) ** 8 * 0.2).astype(np.float32)
        a = (rng.random(N) ** 8 * 0.4).astype(np.float32)
        oracle.update_nodes(m["pts_offsets"], m["anchors"][:, 1].copy(), w, a, ws, as_, vc, nodes_ref)
        ops.update_nodes(gpu["m"]["pts_offsets"], gpu["m"]["anchors"], _t(np.pad(w, (0, gpu["m"]["ts"].shape[0] - N)), dev),
                         _t(np.pad(a, (0, gpu["m"]["ts"].shape[0] - N)), dev), ws_t, as_t, vc_t, nodes_t)
        assert np.array_equal(ws_t.cpu().numpy(), ws) and np.array_equal(as_t.cpu().numpy(), as_)
        assert np.array_equal(vc_t.cpu().numpy(), vc)
        assert np.array_equal(nodes_t.cpu().numpy(), nodes_ref.view(np.uint8).reshape(-1))
    assert (nodes_ref["trans_idx"] != sc["nodes"].view(oracle.TREE_NODE_DTYPE)["trans_idx"]).sum() > 0


# ------------------------------------------------------------------------------ MLP
def test_mlp_matches_reference_golden(ops, dev, golden):
    """a8: MLPNetwork fwd + all grads vs vectors captured from gfnerf/mlp.py (257 rows: ragged last tile)."""
    for tag, acts in (("base", ["relu", "none"]), ("head", ["relu", "relu", "sigmoid"])):
        hs = [_t(golden[f"mlp_{tag}_x"], dev)]
        Ws = [_t(golden[f"mlp_{tag}_W{i}"], dev) for i in range(len(acts))]
        bs = [_t(golden[f"mlp_{tag}_b{i}"], dev) for i in range(len(acts))]
        for i, act in enumerate(acts):
            hs.append(ops.linear_fwd(hs[-1], Ws[i], bs[i], act))
        np.testing.assert_allclose(hs[-1].cpu().numpy(), golden[f"mlp_{tag}_y"], rtol=1e-5, atol=1e-6)
        g = _t(golden[f"mlp_{tag}_gy"], dev)
        for i in reversed(range(len(acts))):
            g, dW, db = ops.linear_bwd(hs[i], Ws[i], hs[i + 1], g, acts[i])
            np.testing.assert_allclose(dW.cpu().numpy(), golden[f"mlp_{tag}_gW{i}"], rtol=2e-4, atol=2e-5)
            np.testing.assert_allclose(db.cpu().numpy(), golden[f"mlp_{tag}_gb{i}"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(g.cpu().numpy(), golden[f"mlp_{tag}_gx"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("n,in_dim,out_dim,act", [(1000, 32, 128, "relu"), (33, 128, 16, "none"), (4099, 63, 128, "relu"),
                                                   (777, 128, 128, "relu"), (500, 128, 3, "sigmoid"), (1, 5, 7, "none")])
def test_linear_vs_oracle(oracle, ops, dev, n, in_dim, out_dim, act):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, in_dim)).astype(np.float32)
    W = (rng.standard_normal((out_dim, in_dim)) / np.sqrt(in_dim)).astype(np.float32)
    b = rng.standard_normal(out_dim).astype(np.float32)
    dy = rng.standard_normal((n, out_dim)).astype(np.float32)
    y_ref = oracle.linear_fwd(x, W, b, act)
    dx_ref, dW_ref, db_ref = oracle.linear_bwd(x, W, y_ref, dy, act)
    y = ops.linear_fwd(_t(x, dev), _t(W, dev), _t(b, dev), act)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref, rtol=2e-5, atol=2e-6)
    dx, dW, db = ops.linear_bwd(_t(x, dev), _t(W, dev), y, _t(dy, dev), act)
    np.testing.assert_allclose(dx.cpu().numpy(), dx_ref, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dW.cpu().numpy(), dW_ref, rtol=2e-4, atol=1e-4 * max(1, n ** 0.5) * 1e-1)
    np.testing.assert_allclose(db.cpu().numpy(), db_ref, rtol=2e-4, atol=1e-4 * max(1, n ** 0.5) * 1e-1)


def test_sh4_and_head_pack(oracle, ops, dev):
    import torch

    rng = np.random.default_rng(2)
    R, N = 50, 900
    d = rng.standard_normal((R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sh = ops.sh4(_t(d, dev))
    want = oracle.sh4(((d + 1.0) / 2.0).astype(np.float32), convention=0, round_f16=True)
    got = sh.cpu().numpy()
    # identical polynomial, f16-rounded: allow one f16 ulp where f32 evaluation order differs
    assert np.abs(got - want).max() <= 2 ** -10 * np.abs(want).max()
    assert (got == want).mean() > 0.98
    h = rng.standard_normal((N, 16)).astype(np.float32)
    ray_id = np.sort(rng.integers(0, R, N)).astype(np.int32)
    cam = rng.integers(0, 9, R).astype(np.int32)
    emb = rng.standard_normal((9, 32)).astype(np.float32)
    sigma, xh = ops.head_pack(_t(h, dev), sh, _t(emb, dev), _t(cam, dev), _t(ray_id, dev))
    xh = xh.cpu().numpy()
    assert np.array_equal(xh[:, :16], got[ray_id]) and np.array_equal(xh[:, 16:31], h[:, 1:16])
    assert np.array_equal(xh[:, 31:63], emb[cam[ray_id]]) and np.all(xh[:, 63] == 0)
    np.testing.assert_allclose(sigma.cpu().numpy(), oracle.density_fwd(h), rtol=2e-6)
    # backward
    counts = np.bincount(ray_id, minlength=R)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    dxh = rng.standard_normal((N, 64)).astype(np.float32)
    dsig = rng.standard_normal(N).astype(np.float32)
    demb = torch.zeros((9, 32), device=dev)
    dh = ops.head_unpack_bwd(_t(h, dev), _t(dsig, dev), _t(dxh, dev), _t(cam, dev), _t(offs, dev), demb=demb).cpu().numpy()
    np.testing.assert_allclose(dh[:, 0], oracle.density_bwd(h, dsig), rtol=2e-6)
    assert np.array_equal(dh[:, 1:], dxh[:, 16:31])
    want_demb = np.zeros((9, 32), np.float64)
    np.add.at(want_demb, cam[ray_id], dxh[:, 31:63].astype(np.float64))
    np.testing.assert_allclose(demb.cpu().numpy(), want_demb, rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------------- compositing
def _packed_golden(golden):
    counts = golden["cmp_counts"]
    S = golden["cmp_sigma"].shape[1]
    mask = np.arange(S)[None, :] < counts[:, None]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    f = lambda k: golden[k][..., 0][mask]  # noqa: E731
    return offs, mask, f("cmp_sigma"), f("cmp_delta"), f("cmp_t"), golden["cmp_rgb"][mask]


def test_composite_matches_reference_golden(ops, dev, golden):
    """a12-a13 vs get_weights_f2nerf + renderers captured from the reference (incl. inf density, empty rays)."""
    offs, mask, sigma, delta, t, rgb = _packed_golden(golden)
    out = ops.composite_fwd(_t(offs, dev), _t(sigma, dev), _t(delta, dev), _t(t, dev), _t(np.nan_to_num(rgb), dev))
    np.testing.assert_allclose(out["weights"].cpu().numpy(), golden["cmp_weights"][..., 0][mask], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(out["alphas"].cpu().numpy(), golden["cmp_alphas"][..., 0][mask], rtol=2e-5, atol=1e-6)
    ref_rgb = golden["cmp_rgb_eval"]  # NaN colour removed by nan_to_num in eval mode; clamp is a no-op below 1
    np.testing.assert_allclose(np.clip(out["rgb"].cpu().numpy(), 0, 1), ref_rgb, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["acc"].cpu().numpy(), golden["cmp_acc"][:, 0], rtol=2e-5, atol=2e-6)
    depth = ops.depth_finalize(out["depth_raw"], out["t_minmax"], 1.0).cpu().numpy()
    np.testing.assert_allclose(depth, golden["cmp_depth"][:, 0], rtol=2e-5, atol=2e-6)


def test_composite_vs_oracle_long_rays_and_early_termination(oracle, ops, dev):
    """Rays of up to 1024 samples (16 wave-chunks), early termination at T<1e-4 within 1e-3 of the full sum."""
    rng = np.random.default_rng(4)
    counts = np.array([0, 1, 63, 64, 65, 200, 1024, 700, 129, 5] * 20)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    N = int(offs[-1])
    sigma = np.exp(rng.standard_normal(N) * 2).astype(np.float32)
    dt = (rng.random(N) * 0.1 + 0.01).astype(np.float32)
    t = np.concatenate([np.cumsum(rng.random(c) * 0.1 + 0.05) + 0.2 for c in counts]).astype(np.float32)
    rgb = rng.random((N, 3)).astype(np.float32)
    full = oracle.composite_fwd(offs, sigma, dt, t, rgb)
    for early in (0.0, 1e-4):
        ref = oracle.composite_fwd(offs, sigma, dt, t, rgb, early_T=early)
        out = ops.composite_fwd(_t(offs, dev), _t(sigma, dev), _t(dt, dev), _t(t, dev), _t(rgb, dev), early_T=early)
        np.testing.assert_allclose(out["rgb"].cpu().numpy(), ref["rgb"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(out["acc"].cpu().numpy(), ref["acc"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(out["depth_raw"].cpu().numpy(), ref["depth_raw"], rtol=1e-4, atol=1e-5)
        w = out["weights"].cpu().numpy()
        bad = ~np.isclose(w, ref["weights"], rtol=1e-4, atol=1e-6)
        assert bad.mean() < 1e-4  # a sample exactly at the threshold may fall either side
        # the north-star tolerance: early-terminated colours stay within 1e-3 abs of the untruncated ones
        assert np.abs(out["rgb"].cpu().numpy() - full["rgb"]).max() < 1e-3
        d = ops.depth_finalize(out["depth_raw"], out["t_minmax"], 10.0).cpu().numpy()
        np.testing.assert_allclose(d, oracle.depth_finalize(ref["depth_raw"], offs, t, 10.0), rtol=1e-4, atol=1e-5)
    g_rgb = rng.standard_normal((len(counts), 3)).astype(np.float32)
    g_acc = rng.standard_normal(len(counts)).astype(np.float32)
    ds_ref, dr_ref = oracle.composite_bwd(offs, sigma, dt, rgb, g_rgb, g_acc)
    ds, dr = ops.composite_bwd(_t(offs, dev), _t(sigma, dev), _t(dt, dev), _t(rgb, dev), _t(g_rgb, dev), _t(g_acc, dev))
    np.testing.assert_allclose(dr.cpu().numpy(), dr_ref, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ds.cpu().numpy(), ds_ref, rtol=2e-3, atol=2e-6)


def test_composite_backward_matches_reference_golden(ops, dev, golden):
    offs, mask, sigma, delta, t, rgb = _packed_golden(golden)
    counts = golden["cmp_counts"]
    ds, dr = ops.composite_bwd(_t(offs, dev), _t(sigma, dev), _t(delta, dev), _t(np.nan_to_num(rgb), dev),
                               _t(golden["cmp_g_rgb"], dev), _t(golden["cmp_g_acc"][:, 0].copy(), dev))
    ref_ds = golden["cmp_d_sigma"][..., 0][mask]
    ref_dr = golden["cmp_d_rgb"][mask]
    ray_of = np.repeat(np.arange(len(counts)), counts)
    finite = np.isfinite(ref_ds) & np.isfinite(sigma) & (sigma < 1e20)
    keep = ~np.isin(ray_of, np.unique(ray_of[~finite]))
    np.testing.assert_allclose(ds.cpu().numpy()[keep], ref_ds[keep], rtol=5e-4, atol=2e-6)
    np.testing.assert_allclose(dr.cpu().numpy()[keep], ref_dr[keep], rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
def test_fill_zero_replays(ops, dev):
    """gfn_fill_zero (and every zeroing inside the library: csrc/gfn_common.h, gfn_zero_async) is a kernel, not a memset: captured into a
    hipGraph, a hipMemsetAsync node zeroes its whole buffer on the FIRST launch of the instantiated graph only (this image's ROCm runtime; measured
    in round 5 on 4 KB, 1 MB and 71 MB buffers, first or in the middle of the graph: a quarter to all of the buffer left as it was from the second
    replay on).  Here: buffers of the sizes the training step zeroes (the 71 MB gradient pool, a 4-byte maximum, ragged ends), first and mid-graph,
    four replays each, all zero every time; and the eager call with unaligned ends touches nothing outside its range."""
    import torch

    big = torch.ones(71303168 // 2, dtype=torch.float16, device=dev)
    small = torch.ones(1, dtype=torch.float32, device=dev)
    ragged = torch.ones(4099, dtype=torch.uint8, device=dev)
    x = torch.zeros(16, device=dev)
    for order in ("first", "mid"):
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            if order == "mid":
                x.add_(1.0)
            ops.fill_zero(big)
            x.add_(1.0)
            ops.fill_zero(small)
            ops.fill_zero(ragged[3:4098])
            x.add_(1.0)
        for rep in range(4):
            big.fill_(1.0); small.fill_(1.0); ragged.fill_(1)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert int((big != 0).sum()) == 0 and float(small) == 0.0, (order, rep)
            assert ragged[:3].tolist() == [1, 1, 1] and int(ragged[4098]) == 1 and int(ragged[3:4098].sum()) == 0, (order, rep)
    for lo, hi in ((0, 0), (1, 2), (5, 37), (15, 4099), (16, 4096), (17, 33)):
        ragged.fill_(1)
        ops.fill_zero(ragged[lo:hi])
        want = torch.ones(4099, dtype=torch.uint8, device=dev)
        want[lo:hi] = 0
        assert torch.equal(ragged, want), (lo, hi)


def test_charbonnier_and_adam(oracle, ops, dev, golden):
    loss, g = ops.charbonnier(_t(golden["charb_pred"], dev), _t(golden["charb_gt"], dev))
    np.testing.assert_allclose(loss.item(), golden["charb_loss"], rtol=1e-5)
    np.testing.assert_allclose(g.cpu().numpy(), golden["charb_gpred"], rtol=1e-4, atol=1e-7)
    import torch

    p = _t(golden["adam_p0"].copy(), dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for s in range(3):
        ops.adam_step(p, _t(golden["adam_g"][s], dev), m, v, s + 1)
    np.testing.assert_allclose(p.cpu().numpy(), golden["adam_p3"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("m_feat,k_feat,m_valid,k_valid", [(128, 128, 128, 128), (8, 128, 3, 128), (128, 64, 128, 63),
                                                           (16, 128, 16, 128), (128, 32, 128, 32)])
def test_wgrad_x3_matches_f64(ops, dev, m_feat, k_feat, m_valid, k_valid, tiled):
    """gfn_wgrad_x3 (split-f16 planes, transposed LDS reads) against a float64 dz^T a, ragged n, device count, scale, col_map;
    planes as rows [n][width] and in the tile-major layout [n/32][width/8][32][8] the fused decoder kernels write."""
    import torch

    rng = np.random.default_rng(5)
    n_cap, n = 5024, 4321  # whole 32-sample tiles allocated; n not a multiple of the 64-sample slab
    dz = (rng.standard_normal((n_cap, m_feat)) * np.exp(rng.uniform(-6, 0, (n_cap, 1)))).astype(np.float32)
    a = np.maximum(rng.standard_normal((n_cap, k_feat)), 0).astype(np.float32) * 3.0
    dz[:, m_valid:] = 0
    scale = 64.0
    t = lambda x: torch.from_numpy(x).to(dev)  # noqa: E731
    dzp = ops.split_f16(t(dz * scale)); ap = ops.split_f16(t(a))
    if tiled:
        tile = lambda x: x.view(n_cap // 32, 32, x.shape[1] // 8, 8).permute(0, 2, 1, 3).contiguous().view(-1)  # noqa: E731
        dzp, ap = (tile(dzp[0]), tile(dzp[1]), m_feat), (tile(ap[0]), tile(ap[1]), k_feat)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    col_map = None
    ldw = k_valid
    if k_valid == 63:  # permuted / dropped columns, like the head's first layer
        perm = rng.permutation(63).astype(np.int32)
        perm[5] = -1
        col_map = t(np.concatenate([perm, [-1]]).astype(np.int32))
    dW = torch.zeros((m_valid, ldw), dtype=torch.float32, device=dev)
    db = torch.zeros(m_valid, dtype=torch.float32, device=dev)
    assert ops.grad_scale(0.2) == scale  # gmax in [1/8, 1/4) -> scale 64
    ops.wgrad_x3(dzp, ap, dW, db=db, n_dev=n_dev, gmax=torch.tensor([0.2], device=dev), m_valid=m_valid, k_valid=k_valid,
                 col_map=col_map, tiled=tiled)
    want = dz[:n, :m_valid].astype(np.float64).T @ a[:n, :k_valid].astype(np.float64)
    if col_map is not None:
        w2 = np.zeros((m_valid, ldw))
        cm = col_map.cpu().numpy()
        for k in range(k_valid):
            if cm[k] >= 0:
                w2[:, cm[k]] = want[:, k]
        want = w2
    got = dW.cpu().numpy()
    mag = (np.abs(dz[:n, :m_valid]).astype(np.float64).T @ np.abs(a[:n, :k_valid]).astype(np.float64)).max()
    assert np.abs(got - want).max() < 2e-6 * mag, (np.abs(got - want).max(), mag)
    np.testing.assert_allclose(db.cpu().numpy(), dz[:n, :m_valid].astype(np.float64).sum(0), rtol=1e-4, atol=2e-6 * np.abs(dz[:n]).sum(0).max())


@pytest.mark.parametrize("form", ["x3", "rne", "mixed"])
@pytest.mark.parametrize("with_parts", [True, False])
def test_wgrad_x3_layers_matches_f64(ops, dev, with_parts, form):
    """gfn_wgrad_x3_layers: the five decoder layers in one launch (slabs by LDS-DMA, bias sums from the MFMA fragments), partial
    vectors + gfn_wgrad_reduce or atomics, against float64 dz^T a; ragged n, device count, scale, col_map, gaps in the flat vector.
    form: both operands as hi + lo' planes ("x3"), both as ONE plane rounded to nearest ("rne": 64-sample slabs, one MFMA per
    fragment pair) or dz one plane and the activations split ("mixed") -- the contraction of the values the planes hold."""
    import torch

    rng = np.random.default_rng(11)
    n_cap, n = 6048, 5555
    scale = 64.0
    t = lambda x: torch.from_numpy(x).to(dev)  # noqa: E731
    tile = lambda x: x.view(n_cap // 32, 32, x.shape[1] // 8, 8).permute(0, 2, 1, 3).contiguous().view(-1)  # noqa: E731
    shapes = [(8, 128, 3, 128), (128, 128, 128, 128), (128, 64, 128, 63), (16, 128, 16, 128), (128, 32, 128, 32)]
    sizes = [mv * (63 if kv == 63 else kv) + mv for _, _, mv, kv in shapes]
    grad = torch.zeros(sum(sizes) + 40, dtype=torch.float32, device=dev)  # 40 elements no layer owns
    grad[-40:] = 7.0
    layers, wants, off = [], [], 0
    for (m_feat, k_feat, m_valid, k_valid) in shapes:
        dz = (rng.standard_normal((n_cap, m_feat)) * np.exp(rng.uniform(-6, 0, (n_cap, 1)))).astype(np.float32)
        a = np.maximum(rng.standard_normal((n_cap, k_feat)), 0).astype(np.float32) * 3.0
        dz[:, m_valid:] = 0
        if form != "x3":  # one plane: the operand IS its f16 rounding
            dz = ((dz * scale).astype(np.float16).astype(np.float32) / scale).astype(np.float32)
        if form == "rne":
            a = a.astype(np.float16).astype(np.float32)
        dzp, ap = ops.split_f16(t(dz * scale)), ops.split_f16(t(a))
        if form != "x3":
            assert float(dzp[1].abs().max()) == 0.0
            dzp = (dzp[0], None)
        if form == "rne":
            ap = (ap[0], None)
        ldw = k_valid
        col_map = None
        want = dz[:n, :m_valid].astype(np.float64).T @ a[:n, :k_valid].astype(np.float64)
        if k_valid == 63:
            perm = rng.permutation(63).astype(np.int32)
            perm[7] = -1
            col_map = t(np.concatenate([perm, [-1]]).astype(np.int32))
            w2 = np.zeros((m_valid, ldw))
            for k in range(k_valid):
                if perm[k] >= 0:
                    w2[:, perm[k]] = want[:, k]
            want = w2
        dW = grad[off:off + m_valid * ldw].view(m_valid, ldw); off += m_valid * ldw
        db = grad[off:off + m_valid]; off += m_valid
        tl = lambda x: None if x is None else tile(x)  # noqa: E731
        layers.append(((tile(dzp[0]), tl(dzp[1]), m_feat), (tile(ap[0]), tl(ap[1]), k_feat), dW, db, m_valid, col_map))
        mag = (np.abs(dz[:n, :m_valid]).astype(np.float64).T @ np.abs(a[:n, :k_valid]).astype(np.float64)).max()
        wants.append((want, dz[:n, :m_valid].astype(np.float64).sum(0), mag, np.abs(dz[:n]).sum(0).max()))
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
    parts = ops.wgrad_parts(grad.numel(), dev) if with_parts else None
    for rep in range(2):  # the partial vectors are reused from step to step without being cleared
        grad[:-40] = 0
        ops.wgrad_x3_layers(layers, grad, n_cap, n_dev=n_dev, gmax=torch.tensor([0.2], device=dev), parts=parts)
        for (dz_, a_, dW, db, m_valid, _), (want, wb, mag, bmag) in zip(layers, wants):
            assert np.abs(dW.cpu().numpy() - want).max() < 2e-6 * mag
            np.testing.assert_allclose(db.cpu().numpy(), wb, rtol=1e-4, atol=2e-6 * bmag)
        assert float(grad[-40:].min()) == 7.0 and float(grad[-40:].max()) == 7.0


def test_copy_list(ops, dev):
    """gfn_copy_list: several small device copies in one launch; mixed dtypes and sizes, an empty one, argument checks."""
    import torch

    from gfnerf_amd import _lib

    g = torch.Generator(device="cpu").manual_seed(3)
    srcs = [torch.randint(0, 1 << 30, (4096,), generator=g, dtype=torch.int32).to(dev), torch.rand((4096, 3), generator=g).to(dev),
            torch.rand((5, 3), generator=g).to(dev), torch.zeros(0, dtype=torch.float32, device=dev), torch.rand((70001,), generator=g).to(dev)]
    dsts = [torch.full_like(s, 7) for s in srcs]
    ops.copy_list(list(zip(dsts, srcs)))
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)
    with pytest.raises(_lib.GfnError):
        ops.copy_list([(dsts[1], srcs[0])])  # dtype / size mismatch
    with pytest.raises(_lib.GfnError):
        ops.copy_list([(torch.zeros(3, dtype=torch.float16, device=dev), torch.zeros(3, dtype=torch.float16, device=dev))])  # 6 bytes
    with pytest.raises(_lib.GfnError):
        ops.copy_list([(dsts[0], srcs[0])] * 9)  # more than 8


def test_generate_rays_matches_golden_and_oracle(oracle, ops, dev, golden):
    """gfn_generate_rays against the vectors captured from the reference's Cameras.generate_rays, explicit and raster mode."""
    import torch

    g = golden
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    c2w, intr = t(g["rays_c2w"]), t(g["rays_intr"])
    r = ops.generate_rays(c2w, intr, cam_idx=t(g["rays_cam_idx"]), coords=t(g["rays_coords"]))
    assert np.array_equal(r["origins"].cpu().numpy(), g["rays_origins"]) and np.array_equal(r["lookat"].cpu().numpy(), g["rays_lookat"])
    np.testing.assert_allclose(r["directions"].cpu().numpy(), g["rays_directions"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(r["dir_norm"].cpu().numpy(), g["rays_dir_norm"][:, 0], rtol=2e-7)
    o = oracle.generate_rays(g["rays_c2w"], g["rays_intr"], g["rays_cam_idx"], g["rays_coords"])
    np.testing.assert_allclose(r["pixel_area"].cpu().numpy(), o["pixel_area"], rtol=1e-3)
    np.testing.assert_allclose(r["pixel_area"].cpu().numpy(), g["rays_pixel_area"][:, 0], rtol=2e-3)
    # raster mode: row 431 of camera 2's 1152-wide image
    W = g["rays_row_directions"].shape[0]
    row = ops.generate_rays(c2w, intr, cam0=2, first_pixel=431 * W, width=W, n=W)
    np.testing.assert_allclose(row["directions"].cpu().numpy(), g["rays_row_directions"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(row["pixel_area"].cpu().numpy(), g["rays_row_pixel_area"][:, 0], rtol=2e-3)


def test_s3im_matches_golden(ops, dev, golden):
    """gfn_s3im (loss + gradient, accumulating) against the reference's S3IM + autograd."""
    import torch

    g = golden
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    loss, grad = ops.s3im(t(g["s3im_image"]), t(g["s3im_pred"]), t(g["s3im_index"]), patch_h=32)
    np.testing.assert_allclose(loss.item(), float(g["s3im_loss"]), rtol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), g["s3im_gpred"], rtol=2e-3, atol=2e-6)
    # accumulation semantics: loss += mult * s3im, g_pred += mult * grad
    base_loss = torch.full((1,), 0.25, device=dev); base_g = torch.ones_like(grad)
    ops.s3im(t(g["s3im_image"]), t(g["s3im_pred"]), t(g["s3im_index"]), patch_h=32, mult=0.5, loss=base_loss, g_pred=base_g)
    np.testing.assert_allclose(base_loss.item(), 0.25 + 0.5 * float(g["s3im_loss"]), rtol=2e-5)
    np.testing.assert_allclose(base_g.cpu().numpy(), 1.0 + 0.5 * g["s3im_gpred"], rtol=1e-5, atol=1e-6)
    idx = ops.s3im_index(256, 10, device=dev)
    assert idx.shape == (2560,) and idx.dtype == torch.int32 and torch.equal(idx[:256].cpu(), torch.arange(256, dtype=torch.int32))
    assert all(sorted(idx[256 * k: 256 * (k + 1)].cpu().tolist()) == list(range(256)) for k in range(1, 10))


@pytest.mark.parametrize("n,log2_T,halves_from,split_log2", [(5000, 14, 0, 1), (5000, 14, 0, 2), (777, 12, 8, 3), (300000, 16, 6, 0),
                                                             (300000, 16, 6, 2), (1, 12, 0, 2)])
def test_level_major_lookup_is_bit_identical(ops, dev, n, log2_T, halves_from, split_log2):
    """gfn_hash_fwd_lm (all workgroups sweep level by level, one L2-sized address part -- half, quarter, eighth -- at a time)
    == gfn_hash_fwd_f16, bit for bit: one table, two tables with their own primes, row-major and level-major output, device-side
    live count, and a sample count that needs more than one round of workgroup runs."""
    import functools
    import torch

    rng = np.random.default_rng(n)
    V = 37
    T = 1 << log2_T
    tabs = [torch.from_numpy((rng.standard_normal((16 * T, 2)) * 0.1).astype(np.float16)).to(dev) for _ in range(2)]
    prims = [torch.from_numpy(rng.integers(2**28, 2**30, (16, V, 3)).astype(np.int32)).to(dev) for _ in range(2)]
    bias = torch.from_numpy(rng.random((16 * V, 3)).astype(np.float32)).to(dev)
    pts = torch.from_numpy((rng.random((n, 3)) * 1.2 - 0.1).astype(np.float32)).to(dev)
    anc = torch.from_numpy(np.sort(rng.integers(0, V, n)).astype(np.int32)).to(dev)
    lm = functools.partial(ops.hash_fwd_lm, split_log2=split_log2)
    for b in (None, bias):
        ref1 = ops.hash_fwd_f16(tabs[0], prims[0], b, pts, anc, V, log2_T)
        ref2 = ops.hash_fwd_f16(tabs[0], prims[0], b, pts, anc, V, log2_T, table2_f16=tabs[1], prim_pool2=prims[1])
        got1 = lm(tabs[0], prims[0], b, pts, anc, V, log2_T, halves_from=halves_from)
        got2 = lm(tabs[0], prims[0], b, pts, anc, V, log2_T, table2_f16=tabs[1], prim_pool2=prims[1], halves_from=halves_from)
        assert torch.equal(ref1, got1) and torch.equal(ref2, got2)
        pl = lm(tabs[0], prims[0], b, pts, anc, V, log2_T, table2_f16=tabs[1], prim_pool2=prims[1], halves_from=halves_from, planar=True)
        assert pl.shape == (16, n, 2) and torch.equal(pl.permute(1, 0, 2).reshape(n, 32), ref2)
    # live count on the device smaller than the buffers: rows beyond it are not written
    if n > 10:
        live = torch.tensor([n - 7], dtype=torch.int32, device=dev)
        out = torch.full((n, 32), -5.0, device=dev)
        lm(tabs[0], prims[0], None, pts, anc, V, log2_T, out=out, n_dev=live, halves_from=halves_from)
        ref = ops.hash_fwd_f16(tabs[0], prims[0], None, pts, anc, V, log2_T)
        assert torch.equal(out[: n - 7], ref[: n - 7]) and bool((out[n - 7:] == -5.0).all())
    # shared primes (second table without its own pool)
    ref = ops.hash_fwd_f16(tabs[0], prims[0], None, pts, anc, V, log2_T, table2_f16=tabs[1])
    got = lm(tabs[0], prims[0], None, pts, anc, V, log2_T, table2_f16=tabs[1], halves_from=halves_from)
    assert torch.equal(ref, got)


def test_grad_records_reduce_exact_and_f16_paths(ops, dev):
    """gfn_grad_records_reduce on hand-made record streams (the layout include/gfnerf_amd.h documents): small values take the
    exact fixed-point path -- the pool is f16(exact sum), whatever the order --; a range that could overflow the 32-bit
    fields (sum of |component| >= 128) is done again in coarser fixed-point units instead of wrapping (a NaN among the addends: on the
    f16 path, which carries it); f16_accumulate forces the reference's f16 adds; the counters are consumed."""
    import torch

    from gfnerf_amd import _lib

    log2_T = 12
    T = 1 << log2_T
    n_ranges = _lib.lib().gfn_grad_record_ranges(log2_T)
    assert n_ranges == 3  # 8.5 T = 34816 f16 pairs in ranges of 16 Ki
    pool_slots = 8 * T + T // 2
    rng = np.random.default_rng(5)
    cap = 64
    rec, cnt = ops.grad_records_buffers(log2_T, n_ranges * 256 * cap, dev)
    rec_np = np.zeros((n_ranges, 256, cap, 2), np.uint32)
    cnt_np = np.zeros((n_ranges, 256), np.int32)
    want = np.zeros((pool_slots, 2), np.float64)

    def fill(r, wg, slots, vals):
        n = len(slots)
        rec_np[r, wg, :n, 0] = slots
        rec_np[r, wg, :n, 1] = vals.astype(np.float16).view(np.uint16).astype(np.uint32) @ np.array([1, 65536], np.uint32)
        cnt_np[r, wg] = n
        np.add.at(want, slots, vals.astype(np.float16).astype(np.float64))

    # range 0: thousands of small contributions, many per slot (f16 adds would swamp them); range 1: big values -> f16 path;
    # range 2 (partial: slots up to 34816): a few records in its last slots
    for wg in range(256):
        fill(0, wg, rng.integers(0, 64, cap), (rng.standard_normal((cap, 2)) * 1e-3 + 2e-3))
        fill(1, wg, 16384 + rng.integers(0, 8, cap), np.full((cap, 2), 0.5) * np.array([1.0, -1.0]))
    fill(2, 7, np.array([pool_slots - 1, pool_slots - 2, pool_slots - 1]), np.array([[1e-4, 2e-4], [3e-5, 0.0], [-5e-5, 6.1e-5]]))
    upload = lambda: (rec.copy_(torch.from_numpy(rec_np.reshape(-1, 2).view(np.int64).reshape(-1).copy()).to(dev)),  # noqa: E731
                      cnt.copy_(torch.from_numpy(cnt_np.reshape(-1)).to(dev)))
    got = {}
    for mode in ("auto", "f16"):
        upload()
        pool = torch.zeros((16 * T, 2), dtype=torch.float16, device=dev)
        seen = torch.zeros_like(cnt)
        ops.grad_records_reduce((rec, cnt), pool, log2_T, counts_seen=seen, f16_accumulate=mode == "f16")
        assert int(cnt.abs().max().item()) == 0 and np.array_equal(seen.cpu().numpy(), cnt_np.reshape(-1))
        got[mode] = pool.float().cpu().numpy()[:pool_slots].astype(np.float64)
        assert float(pool[pool_slots:].abs().max()) == 0.0
    exact16 = want.astype(np.float16).astype(np.float64)
    # exact path (ranges 0 and 2): bit-identical to rounding the exact sum once
    assert np.array_equal(got["auto"][:16384], exact16[:16384]) and np.array_equal(got["auto"][32768:], exact16[32768:])
    # range 1 holds 256 * 64 * 0.5 / 8 slots = +-1024 per slot: beyond a 32-bit field of 2^-24 units -> done again in coarser units
    # chosen to fit (round 5; the f16 image before): still the exact sum of these addends (multiples of 0.5), rounded to f16 once
    big = got["auto"][16384:16392]
    assert (big[:, 0] > 500).all() and (big[:, 1] < -500).all() and np.array_equal(big, exact16[16384:16392])
    assert np.abs(got["f16"][16384:16392] - want[16384:16392]).max() <= 0.6 * 1024  # (the reference's f16 adds lose most of it)
    # the reference's accumulation everywhere: small sums lose what the running f16 value swallows, but stay close
    assert np.abs(got["f16"][:64] - want[:64]).max() <= 2e-2 * np.abs(want[:64]).max()
    assert np.abs(got["f16"][:64] - exact16[:64]).max() > 0  # (and is indeed not the exact sum)
    # range by range (ranges=(begin, end)): the same pool, only the named ranges' counters consumed per call
    upload()
    pool = torch.zeros((16 * T, 2), dtype=torch.float16, device=dev)
    ops.grad_records_reduce((rec, cnt), pool, log2_T, ranges=(1, 3))
    left = cnt.cpu().numpy().reshape(n_ranges, 256)
    assert np.array_equal(left[0], cnt_np[0]) and not left[1:].any() and float(pool[:16384].abs().max()) == 0.0
    ops.grad_records_reduce((rec, cnt), pool, log2_T, ranges=(0, 1))
    ops.grad_records_reduce((rec, cnt), pool, log2_T, ranges=(2, 2))  # empty: nothing to do
    assert int(cnt.abs().max().item()) == 0
    assert np.array_equal(pool.float().cpu().numpy()[:pool_slots].astype(np.float64), got["auto"])
    with pytest.raises(Exception):
        ops.grad_records_reduce((rec, cnt), pool, log2_T, ranges=(2, 1))
    # a NaN among a range's addends reaches the pool (as atomicAdd(half2) would carry it), the other ranges are untouched by it
    fill(2, 9, np.array([pool_slots - 5]), np.array([[np.nan, 1.0]]))
    upload()
    pool = torch.zeros((16 * T, 2), dtype=torch.float16, device=dev)
    ops.grad_records_reduce((rec, cnt), pool, log2_T)
    out = pool.float().cpu().numpy()
    assert np.isnan(out[pool_slots - 5, 0]) and out[pool_slots - 5, 1] == 1.0 and np.isfinite(out[:32768]).all()
    assert np.array_equal(out[:16384].astype(np.float64), exact16[:16384])


@pytest.mark.parametrize("lanes", [16, 4])
def test_sample_record_paged_form_holds_the_same_samples(oracle, ops, dev, set_march_lanes, lanes):
    """gfn_sample_rays_record with a page pool (round 5): pages of 32 record slots handed to the rays as they march instead of 1024 slots
    for every ray.  Slot k of ray r -- at (page_table[r * 32 + k / 32] * 32 + k % 32) -- holds the dense form's record bit for bit; every
    page is handed out once; a pool that is too small sets bit 16 of the overflow word and writes nothing outside the pool."""
    import torch

    from gfnerf_amd import scene

    set_march_lanes(lanes)
    n_rays = 160
    sc, ref, _ = _sample_both(oracle, ops, dev, n_rays=n_rays, train=True, fineness=6.0)
    noise = _t(scene.make_noise(n_rays, True, 6.0, seed=1), dev)
    cn, tr = ops.tree_compact(_t(sc["nodes"], dev)), _t(sc["transes"], dev)
    o, d = _t(ref["o"], dev), _t(ref["d"], dev)
    dense = ops.sample_rays_record(o, d, noise, cn, tr, 1.0 / 256, True)
    cnt = dense["counts"].cpu().numpy()
    need = int(np.ceil(cnt / 32).sum())  # pages the rays need
    mean = float(cnt.mean())
    paged = ops.sample_rays_record(o, d, noise, cn, tr, 1.0 / 256, True, pool_slots_per_ray=int(mean) + 40)
    assert paged["n_pages"] >= need and int(paged["overflow"].item()) == 0 and int(paged["page_counter"].item()) == need
    assert torch.equal(paged["counts"], dense["counts"]) and torch.equal(paged["first_oct_dis"], dense["first_oct_dis"])
    D = dense["march_rec"].view(n_rays, ops.MAX_SAMPLE_PER_RAY, ops.MARCH_REC_FLOATS).cpu().numpy().view(np.uint32)
    P = paged["march_rec"].view(-1, ops.REC_PAGE, ops.MARCH_REC_FLOATS).cpu().numpy().view(np.uint32)
    pt = paged["page_table"].view(n_rays, ops.MAX_SAMPLE_PER_RAY // ops.REC_PAGE).cpu().numpy()
    used = []
    for r in range(n_rays):
        for p in range((cnt[r] + 31) // 32):
            k0, k1 = 32 * p, min(32 * p + 32, cnt[r])
            assert 0 <= pt[r, p] < paged["n_pages"]
            used.append(int(pt[r, p]))
            assert np.array_equal(P[pt[r, p], : k1 - k0], D[r, k0:k1]), (r, p)
    assert sorted(used) == list(range(need))  # every page handed out exactly once
    # written again in place (the tile pipeline's ring): the same result, the counter starts from zero again
    again = ops.sample_rays_record(o, d, noise, cn, tr, 1.0 / 256, True, out=paged)
    assert again is paged and int(paged["page_counter"].item()) == need and int(paged["overflow"].item()) == 0
    # a pool of half the pages: flagged, and nothing written outside it
    small = ops.sample_rays_record(o, d, noise, cn, tr, 1.0 / 256, True, pool_slots_per_ray=max(int(mean) // 2, 1))
    assert small["n_pages"] < need and int(small["overflow"].item()) & ops.REC_POOL_OVERFLOW
    assert torch.equal(small["counts"], dense["counts"])
    pts = small["page_table"].view(n_rays, -1).cpu().numpy()
    for r in range(n_rays):
        assert (pts[r, : (cnt[r] + 31) // 32] < small["n_pages"]).all() and (pts[r, : (cnt[r] + 31) // 32] >= 0).all()

"""numpy/ctypes front-end of the CPU oracle (oracle/gfn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  Every function
is a thin wrapper; the algorithm (with reference file:line citations) lives in
gfn_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("GFN_ORACLE_LIB") or os.path.join(_HERE, "libgfn_oracle.so")  # (GFN_ORACLE_LIB: the sanitizer build, tests/test_sanitizers.py)

TREE_NODE_DTYPE = np.dtype(
    {
        "names": ["center", "side_len", "parent", "childs", "is_leaf_node", "trans_idx", "block_idx"],
        "formats": [("<f4", 3), "<f4", "<i8", ("<i8", 8), "u1", "<i8", "<i8"],
        "offsets": [0, 12, 16, 24, 88, 96, 104],
        "itemsize": 128,
    }
)
TRANS_INFO_DTYPE = np.dtype(
    {
        "names": ["w2xz", "weight", "center", "side_len", "dis_summary"],
        "formats": [("<f4", (12, 2, 4)), ("<f4", (3, 12)), ("<f4", 3), "<f4", "<f4"],
        "offsets": [0, 384, 528, 540, 544],
        "itemsize": 576,
    }
)


def build(force=False):
    src = os.path.join(_HERE, "gfn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gfn_o_f16_to_f32.restype = C.c_float
        _lib.gfn_o_f16_to_f32.argtypes = [C.c_uint16]
        _lib.gfn_o_f32_to_f16.restype = C.c_uint16
        _lib.gfn_o_f32_to_f16.argtypes = [C.c_float]
        _lib.gfn_o_charbonnier.restype = C.c_double
        _lib.gfn_o_free.argtypes = [C.c_void_p]
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def level_scales():
    out = np.zeros(16, np.float32)
    lib().gfn_o_level_scales(_p(out))
    return out


def search_order():
    out = np.zeros(64, np.uint8)
    lib().gfn_o_search_order(_p(out))
    return out.reshape(8, 8)


def normalize_dirs(d):
    d = _f32(d)
    out = np.empty_like(d)
    lib().gfn_o_normalize_dirs(_p(d), _p(out), C.c_int64(d.shape[0]))
    return out


def hash_fwd(feat_pool, prim_pool, bias_pool, pts, anchors, n_volumes, log2_T, threads=0):
    feat_pool, bias_pool, pts = _f32(feat_pool), _f32(bias_pool), _f32(pts)
    prim_pool, anchors = _i32(prim_pool), _i64(anchors)
    n = pts.shape[0]
    assert feat_pool.size == 16 * (1 << log2_T) * 2
    assert prim_pool.size == 16 * n_volumes * 3 and bias_pool.size == 16 * n_volumes * 3
    assert n == 0 or (anchors.min() >= 0 and anchors.max() < n_volumes)
    out = np.zeros((n, 32), np.float32)
    lib().gfn_o_hash_fwd(_p(feat_pool), _p(prim_pool), _p(bias_pool), _p(pts), _p(anchors), C.c_int64(n),
                         C.c_int64(n_volumes), C.c_int(log2_T), _p(out), C.c_int(threads))
    return out


def hash_bwd(prim_pool, bias_pool, pts, anchors, grad_out, n_volumes, log2_T, threads=0):
    """threads: any number gives the same bits (the double sums of f16 addends are exact, gfn_oracle.c)."""
    bias_pool, pts, grad_out = _f32(bias_pool), _f32(pts), _f32(grad_out)
    prim_pool, anchors = _i32(prim_pool), _i64(anchors)
    n = pts.shape[0]
    grad_pool = np.zeros((16 * (1 << log2_T), 2), np.float32)
    lib().gfn_o_hash_bwd(_p(prim_pool), _p(bias_pool), _p(pts), _p(anchors), _p(grad_out), C.c_int64(n),
                         C.c_int64(n_volumes), C.c_int(log2_T), _p(grad_pool), C.c_int(threads))
    return grad_pool


def trace(rays_o, rays_d, nodes, near=0.01, far=1e8, max_hits=1024, threads=0):
    """-> (offsets i64[R+1], hit_idx i64[H], hit_near_far f32[H,2]); rays_d must be unit."""
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    assert nodes.dtype == TREE_NODE_DTYPE and nodes.flags["C_CONTIGUOUS"]
    R = rays_o.shape[0]
    offsets = np.zeros(R + 1, np.int64)
    args = (_p(rays_o), _p(rays_d), C.c_int64(R), _p(nodes), C.c_float(near), C.c_float(far), C.c_int64(max_hits))
    lib().gfn_o_trace(*args, _p(offsets), None, None, C.c_int(threads))
    H = int(offsets[-1])
    hit_idx = np.zeros(max(H, 1), np.int64)
    hit_nf = np.zeros((max(H, 1), 2), np.float32)
    lib().gfn_o_trace(*args, _p(offsets), _p(hit_idx), _p(hit_nf), C.c_int(threads))
    return offsets, hit_idx[:H], hit_nf[:H]


def march(rays_o, rays_d, noise, sample_l, scale_by_dis, oct_offsets, hit_idx, hit_nf, nodes, transes, threads=0):
    """-> dict of PACKED sample arrays (world_pts, warped_pts, dirs, dists, ts, anchors[N,3]) + pts_offsets + first_oct_dis."""
    rays_o, rays_d, noise = _f32(rays_o), _f32(rays_d), _f32(noise)
    oct_offsets, hit_idx, hit_nf = _i64(oct_offsets), _i64(hit_idx), _f32(hit_nf)
    if hit_idx.size == 0:
        hit_idx = np.zeros(1, np.int64)
        hit_nf = np.zeros((1, 2), np.float32)
    assert transes.dtype == TRANS_INFO_DTYPE and nodes.dtype == TREE_NODE_DTYPE
    R = rays_o.shape[0]
    assert noise.size >= 1024 + R
    pts_offsets = np.zeros(R + 1, np.int64)
    head = (_p(rays_o), _p(rays_d), _p(noise), C.c_int64(R), C.c_float(sample_l), C.c_int(int(scale_by_dis)),
            _p(oct_offsets), _p(hit_idx), _p(hit_nf), _p(nodes), _p(transes), _p(pts_offsets))
    lib().gfn_o_march(*head, None, None, None, None, None, None, None, C.c_int(threads))
    N = int(pts_offsets[-1])
    M = max(N, 1)
    out = dict(
        world_pts=np.zeros((M, 3), np.float32), warped_pts=np.zeros((M, 3), np.float32),
        dirs=np.zeros((M, 3), np.float32), dists=np.zeros(M, np.float32), ts=np.zeros(M, np.float32),
        anchors=np.zeros((M, 3), np.int64),
    )
    first = np.zeros(R, np.float32)
    lib().gfn_o_march(*head, _p(out["world_pts"]), _p(out["warped_pts"]), _p(out["dirs"]), _p(out["dists"]),
                      _p(out["ts"]), _p(out["anchors"]), _p(first), C.c_int(threads))
    out = {k: v[:N] for k, v in out.items()}
    out["pts_offsets"] = pts_offsets
    out["first_oct_dis"] = first
    return out


def warp_point(trans, xyz):
    out = np.zeros(3, np.float32)
    lib().gfn_o_warp_point(_p(trans), _p(_f32(xyz)), _p(out))
    return out


def warp_jac(trans, xyz):
    out = np.zeros(9, np.float32)
    lib().gfn_o_warp_jac(_p(trans), _p(_f32(xyz)), _p(out))
    return out.reshape(3, 3)


def sh4(dirs, convention=0, round_f16=True):
    dirs = _f32(dirs)
    out = np.zeros((dirs.shape[0], 16), np.float32)
    lib().gfn_o_sh4(_p(dirs), C.c_int64(dirs.shape[0]), C.c_int(convention), C.c_int(int(round_f16)), _p(out))
    return out


ACT = {"none": 0, "relu": 1, "sigmoid": 2}


def linear_fwd(x, W, b, act="none", threads=0):
    x, W = _f32(x), _f32(W)
    b = None if b is None else _f32(b)
    n, k = x.shape
    o = W.shape[0]
    assert W.shape[1] == k
    y = np.zeros((n, o), np.float32)
    lib().gfn_o_linear_fwd(_p(x), _p(W), _p(b), C.c_int64(n), C.c_int(k), C.c_int(o), C.c_int(ACT[act]), _p(y),
                           C.c_int(threads))
    return y


def linear_bwd(x, W, y, dy, act="none", need_dx=True, threads=1):
    """threads = 1: the serial accumulation the parity tests are pinned with; 0 = all cores (per-thread double partial sums)."""
    x, W, y, dy = _f32(x), _f32(W), _f32(y), _f32(dy)
    n, k = x.shape
    o = W.shape[0]
    dx = np.zeros((n, k), np.float32) if need_dx else None
    dW = np.zeros((o, k), np.float32)
    db = np.zeros(o, np.float32)
    lib().gfn_o_linear_bwd(_p(x), _p(W), _p(y), _p(dy), C.c_int64(n), C.c_int(k), C.c_int(o), C.c_int(ACT[act]),
                           _p(dx), _p(dW), _p(db), C.c_int(threads))
    return dx, dW, db


def density_fwd(h):
    h = _f32(h)
    sigma = np.zeros(h.shape[0], np.float32)
    lib().gfn_o_density_fwd(_p(h), C.c_int64(h.shape[0]), C.c_int(h.shape[1]), _p(sigma))
    return sigma


def density_bwd(h, dsigma):
    h, dsigma = _f32(h), _f32(dsigma)
    out = np.zeros(h.shape[0], np.float32)
    lib().gfn_o_density_bwd(_p(h), _p(dsigma), C.c_int64(h.shape[0]), C.c_int(h.shape[1]), _p(out))
    return out


def composite_fwd(offsets, sigma, dt, t, rgb, early_T=0.0, threads=0):
    offsets, sigma, dt, t, rgb = _i64(offsets), _f32(sigma), _f32(dt), _f32(t), _f32(rgb)
    R = offsets.shape[0] - 1
    N = max(sigma.shape[0], 1)
    w, a, T = (np.zeros(N, np.float32) for _ in range(3))
    o_rgb = np.zeros((R, 3), np.float32)
    o_acc = np.zeros(R, np.float32)
    o_dep = np.zeros(R, np.float32)
    lib().gfn_o_composite_fwd(_p(offsets), C.c_int64(R), _p(sigma), _p(dt), _p(t), _p(rgb), C.c_float(early_T),
                              _p(w), _p(a), _p(T), _p(o_rgb), _p(o_acc), _p(o_dep), C.c_int(threads))
    n = sigma.shape[0]
    return dict(weights=w[:n], alphas=a[:n], trans=T[:n], rgb=o_rgb, acc=o_acc, depth_raw=o_dep)


def depth_finalize(depth_raw, offsets, t, scale_factor):
    d = _f32(depth_raw).copy()
    offsets, t = _i64(offsets), _f32(t)
    lib().gfn_o_depth_finalize(_p(d), _p(offsets), C.c_int64(offsets.shape[0] - 1), _p(t), C.c_float(scale_factor))
    return d


def composite_bwd(offsets, sigma, dt, rgb, g_rgb, g_acc=None, threads=0):
    offsets, sigma, dt, rgb, g_rgb = _i64(offsets), _f32(sigma), _f32(dt), _f32(rgb), _f32(g_rgb)
    g_acc = None if g_acc is None else _f32(g_acc)
    N = max(sigma.shape[0], 1)
    d_sigma = np.zeros(N, np.float32)
    d_rgb = np.zeros((N, 3), np.float32)
    lib().gfn_o_composite_bwd(_p(offsets), C.c_int64(offsets.shape[0] - 1), _p(sigma), _p(dt), _p(rgb), _p(g_rgb),
                              _p(g_acc), _p(d_sigma), _p(d_rgb), C.c_int(threads))
    n = sigma.shape[0]
    return d_sigma[:n], d_rgb[:n]


def update_nodes(pts_offsets, oct_idx, weights, alphas, weight_stats, alpha_stats, visit_cnt, nodes):
    """In-place on weight_stats/alpha_stats/visit_cnt (i64) and nodes."""
    pts_offsets, oct_idx, weights, alphas = _i64(pts_offsets), _i64(oct_idx), _f32(weights), _f32(alphas)
    for a in (weight_stats, alpha_stats, visit_cnt):
        assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    n_nodes = nodes.shape[0]
    lib().gfn_o_update_nodes(_p(pts_offsets), C.c_int64(pts_offsets.shape[0] - 1), _p(oct_idx), _p(weights),
                             _p(alphas), C.c_int64(n_nodes), _p(weight_stats), _p(alpha_stats), _p(visit_cnt),
                             _p(nodes))


def proc_octree(nodes, weight_stats, alpha_stats, visit_cnt, compact=True, subdivide=False, brute_force=False):
    """PersOctree::ProcOctree -> (nodes, weight_stats, alpha_stats) of the new tree (inputs untouched)."""
    nodes = np.ascontiguousarray(nodes.view(TREE_NODE_DTYPE).reshape(-1))
    weight_stats, alpha_stats, visit_cnt = _i64(weight_stats), _i64(alpha_stats), _i64(visit_cnt)
    n = nodes.shape[0]
    assert weight_stats.shape[0] == n and alpha_stats.shape[0] == n and visit_cnt.shape[0] == n
    out_nodes, out_w, out_a, n_out = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
    lib().gfn_o_proc_octree(_p(nodes), C.c_int64(n), _p(weight_stats), _p(alpha_stats), _p(visit_cnt), C.c_int(int(compact)),
                            C.c_int(int(subdivide)), C.c_int(int(brute_force)), C.byref(out_nodes), C.byref(n_out),
                            C.byref(out_w), C.byref(out_a))
    m = n_out.value
    # copy as raw bytes: a structured copy would drop the padding bytes, which are part of the checkpoint blob
    res_nodes = np.frombuffer(C.string_at(out_nodes.value, m * 128), dtype=np.uint8).copy().view(TREE_NODE_DTYPE)
    res_w = np.frombuffer(C.string_at(out_w.value, m * 8), dtype=np.int64).copy()
    res_a = np.frombuffer(C.string_at(out_a.value, m * 8), dtype=np.int64).copy()
    for ptr in (out_nodes, out_w, out_a):
        lib().gfn_o_free(ptr)
    return res_nodes, res_w, res_a


def mark_invisible(nodes, intri, w2c, bounds):
    """K7 MarkInvisibleNodes, in place on nodes."""
    intri, w2c, bounds = _f32(intri), _f32(w2c), _f32(bounds)
    assert nodes.dtype == TREE_NODE_DTYPE and nodes.flags["C_CONTIGUOUS"]
    lib().gfn_o_mark_invisible(_p(nodes), C.c_int64(nodes.shape[0]), _p(intri), _p(w2c), _p(bounds), C.c_int64(intri.shape[0]))


def visible_cams(c2w, intri, bounds, center, side):
    """GetVisiCams (PersSampler.cpp:45-88), every one of the 128 x res_h pixel rays of every camera against the node box:
    camera c is visible iff any ray has min(far_box, far_c) > max(near_box, near_c).  f32 arithmetic in the order of the
    reference's expressions; returns a bool array [n_cams]."""
    c2w, K, bounds = _f32(c2w), _f32(intri).reshape(-1, 3, 3)[0], _f32(bounds)
    f = np.float32
    half_w, half_h, cx, cy, fx, fy = K[0, 2], K[1, 2], K[0, 2], K[1, 2], K[0, 0], K[1, 1]
    res_w = 128
    res_h = int(np.round(f(res_w) / half_w * half_h))
    i = (f(.5) + (half_h * f(2) - f(1)) * np.arange(res_h, dtype=f) / f(max(res_h - 1, 1))).astype(f)   # linspace(.5, 2h-.5)
    j = (f(.5) + (half_w * f(2) - f(1)) * np.arange(res_w, dtype=f) / f(max(res_w - 1, 1))).astype(f)
    ii, jj = np.meshgrid(i, j, indexing="ij")
    ccx, ccy, ccz = ((jj - cx) / fx).reshape(-1), (-(ii - cy) / fy).reshape(-1), f(-1)
    center = _f32(center)
    lo, hi = center - f(side) * f(.5), center + f(side) * f(.5)
    out = np.zeros(c2w.shape[0], bool)
    with np.errstate(divide="ignore", invalid="ignore"):
        for c in range(c2w.shape[0]):
            m = c2w[c]
            d = np.stack([m[k, 0] * ccx + m[k, 1] * ccy + m[k, 2] * ccz for k in range(3)], -1)   # [n_pix, 3]
            o = m[:3, 3]
            a = np.nan_to_num((lo - o) / d, nan=0.0, posinf=1e6, neginf=-1e6).astype(f)
            b = np.nan_to_num((hi - o) / d, nan=0.0, posinf=1e6, neginf=-1e6).astype(f)
            far = np.minimum(np.maximum(a, b).min(-1), bounds[c, 1])
            near = np.maximum(np.minimum(a, b).max(-1), bounds[c, 0])
            out[c] = bool((far > near).any())
    return out


def generate_rays(c2w, intr, cam_idx, coords):
    """Cameras.generate_rays (perspective, no distortion) -> dict(origins, directions, lookat, pixel_area, dir_norm)."""
    c2w, intr, coords = _f32(c2w), _f32(intr), _f32(coords)
    cam_idx = np.ascontiguousarray(cam_idx, np.int32)
    n = cam_idx.shape[0]
    o = {k: np.zeros((n, 3), np.float32) for k in ("origins", "directions", "lookat")}
    o["pixel_area"] = np.zeros(n, np.float32); o["dir_norm"] = np.zeros(n, np.float32)
    lib().gfn_o_generate_rays(_p(c2w), _p(intr), _p(cam_idx), _p(coords), C.c_int64(n), _p(o["origins"]), _p(o["directions"]),
                              _p(o["lookat"]), _p(o["pixel_area"]), _p(o["dir_norm"]))
    return o


def gf_lr_multiplier(step, lr_init, lr_final=None, max_steps=100000, warmup_steps=0, lr_pre_warmup=1e-8, ramp="cosine",
                     n_split_dataset=1, steps_per_split_dataset=1000, steps_perssampler_init=10000):
    """GFNerfExponentialDecayScheduler.get_scheduler's lambda (engine/schedulers.py:145-182), statement by statement."""
    if lr_final is None:
        lr_final = lr_init
    if step < warmup_steps:
        if ramp == "cosine":
            lr = lr_pre_warmup + (1 - lr_pre_warmup) * np.sin(0.5 * np.pi * np.clip(step / warmup_steps, 0, 1))
        else:
            lr = lr_pre_warmup + (lr_init - lr_pre_warmup) * step / warmup_steps
    else:
        init_stage = steps_perssampler_init > 0 and step < steps_perssampler_init
        if init_stage:
            relative_step = step
        else:
            cur_split = ((step - steps_perssampler_init) // steps_per_split_dataset) % n_split_dataset
            cur_circles = (step - steps_perssampler_init) // (steps_per_split_dataset * n_split_dataset)
            relative_step = (step - steps_perssampler_init - cur_circles * steps_per_split_dataset * n_split_dataset
                             - cur_split * steps_per_split_dataset + cur_circles * steps_per_split_dataset)
        t = np.clip((relative_step - warmup_steps) / (max_steps - warmup_steps), 0, 1)
        lr = np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return lr / lr_init


def s3im(a, b, index, patch_h=32):
    """S3IM loss (kernel 4, stride 4) -> (loss, d loss / d b)."""
    a, b = _f32(a), _f32(b)
    index = np.ascontiguousarray(index, np.int32)
    d = np.zeros_like(b)
    lib().gfn_o_s3im.restype = C.c_double
    loss = lib().gfn_o_s3im(_p(a), _p(b), _p(index), C.c_int64(index.size), C.c_int(patch_h), _p(d), C.c_int64(a.shape[0]))
    return float(loss), d


def charbonnier(pred, gt, eps=1e-6):
    pred, gt = _f32(pred), _f32(gt)
    d = np.zeros_like(pred)
    loss = lib().gfn_o_charbonnier(_p(pred), _p(gt), C.c_int64(pred.size), C.c_float(eps), _p(d))
    return float(loss), d


def adam_step(p, g, m, v, step, lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-15):
    """In-place on p, m, v (f32 contiguous)."""
    g = _f32(g)
    lib().gfn_o_adam_step(_p(p), _p(g), _p(m), _p(v), C.c_int64(p.size), C.c_int(step), C.c_float(lr),
                          C.c_float(beta1), C.c_float(beta2), C.c_float(eps))


def f32_to_f16_roundtrip(a):
    """Vector helper built on numpy's RNE conversion (cross-checked against the C one in tests)."""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)

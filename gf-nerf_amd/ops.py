"""Torch-tensor front-end of the C ABI (include/gfnerf_amd.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
function below validates its tensors, then passes raw device pointers, sizes and the
stream handle to the hand-written HIP kernels.  There is no eager/PyTorch fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

MAX_SAMPLE_PER_RAY = 1024
MAX_OCT_PER_RAY = 1024
MARCH_REC_FLOATS = 8  # GFN_MARCH_REC_FLOATS: t, dist, leaf, transform, warped x y z, block per recorded sample
REC_PAGE = 32         # GFN_REC_PAGE: sample slots per page of sample_rays_record's paged form
REC_POOL_OVERFLOW = 16  # bit of the sampler's overflow word: the page pool ran out
HIT_CAP_REACHED = 8     # ... some ray's traversal stopped at max_hits leaves with nodes still to visit
TREE_NODE_BYTES = 128
TRANS_INFO_BYTES = 576


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream of the current device as a void*.  (torch.cuda.current_stream() builds a Stream object through three
    layers of Python -- 10 us a call, 0.3 ms per training iteration of the plugin surface; the raw getter is one C call.)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype, name, shape_last=None):
    if t is None:
        return
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.GfnError(f"{name}: expected a CUDA(HIP) tensor")
    if t.dtype != dtype:
        raise _lib.GfnError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.GfnError(f"{name}: tensor must be contiguous")
    if shape_last is not None and (t.dim() == 0 or t.shape[-1] != shape_last):
        raise _lib.GfnError(f"{name}: last dimension must be {shape_last}, got {tuple(t.shape)}")


def launch_opts(decoder_cus=0, plain_rounding=False, record_merge=_lib.MERGE_OFF, march_lanes=0, merge_gate=None, colour_x2=False):
    """gfn_launch_opts for ONE call (include/gfnerf_amd.h): the library keeps no launch policy of its own -- whoever launches says how.
    decoder_cus: cap on the persistent kernels' workgroups (0: every CU); plain_rounding: one-plane producers without their dither;
    record_merge: _lib.MERGE_OFF / MERGE_ON, or merge_gate = (front count i32 [1], sample count i32 [1], frac): MERGE_DEVICE, the device picks
    the form from the step's own counts; march_lanes: 0 (by batch size), 4 or 16; colour_x2: the inference kernel's colour head layer 2 with a one-f16
    activation operand (fused_render on march records, tile renderer's form; density path unchanged).  The returned struct keeps the gate's tensors alive."""
    o = _lib.LaunchOpts()
    o.colour_x2 = int(bool(colour_x2))
    o.decoder_cus = int(decoder_cus) if 0 < int(decoder_cus) < 256 else 0
    o.plain_rounding = int(bool(plain_rounding))
    o.march_lanes = int(march_lanes)
    if merge_gate is not None:
        front, n_dev, frac = merge_gate
        _chk(front, torch.int32, "merge gate front count"); _chk(n_dev, torch.int32, "merge gate sample count")
        o.record_merge, o.merge_front, o.merge_n, o.merge_frac = _lib.MERGE_DEVICE, front.data_ptr(), n_dev.data_ptr(), float(frac)
        o._keep = (front, n_dev)
    else:
        o.record_merge = int(record_merge)
    return o


def _opt(opts):
    return None if opts is None else C.byref(opts)


def level_scales():
    out = np.zeros(16, np.float32)
    _lib.lib().gfn_hash_level_scales_host(out.ctypes.data_as(C.c_void_p))
    return out


# --------------------------------------------------------------------------- hash
def _anchor_stride(anchors, n):
    if anchors.dim() == 2 and anchors.shape == (n, 3):
        return 3  # packed sampler anchors: column 0 is trans_idx
    if anchors.numel() != n:
        raise _lib.GfnError("anchors and pts disagree on n")
    return 1


def hash_fwd(feat_pool, prim_pool, bias_pool, pts, anchors, n_volumes, log2_T, out=None, n_dev=None, accumulate=False):
    """Hash3DAnchored forward: pts [n,3] f32, anchors [n] i32|i64 (or packed [n,3] i32) -> [n,32] f32."""
    _chk(feat_pool, torch.float32, "feat_pool"); _chk(prim_pool, torch.int32, "prim_pool")
    _chk(bias_pool, torch.float32, "bias_pool"); _chk(pts, torch.float32, "pts", 3)
    if anchors.dtype not in (torch.int32, torch.int64):
        raise _lib.GfnError("anchors: expected int32 or int64")
    _chk(anchors, anchors.dtype, "anchors")
    n = pts.shape[0]
    if feat_pool.numel() != 16 * (1 << log2_T) * 2:
        raise _lib.GfnError(f"feat_pool has {feat_pool.numel()} elements, expected {16 * (1 << log2_T) * 2}")
    if prim_pool.numel() != 16 * n_volumes * 3 or bias_pool.numel() != 16 * n_volumes * 3:
        raise _lib.GfnError("prim_pool / bias_pool do not match n_volumes")
    stride = _anchor_stride(anchors, n)
    if out is None:
        if accumulate:
            raise _lib.GfnError("accumulate=True needs an existing out tensor")
        out = torch.empty((n, 32), dtype=torch.float32, device=pts.device)
    _chk(out, torch.float32, "out", 32)
    rc = _lib.lib().gfn_hash_fwd(_ptr(feat_pool), _ptr(prim_pool), _ptr(bias_pool), _ptr(pts), _ptr(anchors),
                                 int(anchors.dtype == torch.int64), stride, n, _ptr(n_dev), n_volumes, log2_T,
                                 int(bool(accumulate)), _ptr(out), _stream())
    _lib.check(rc, "gfn_hash_fwd")
    return out


_part_ws = {}


def hash_fwd_lm(table_f16, prim_pool, bias_pool, pts, anchors, n_volumes, log2_T, table2_f16=None, prim_pool2=None, out=None, n_dev=None,
                halves_from=6, planar=False, accumulate=False, split_log2=0, sample_list=None, opts=None):
    """Level-major lookup (csrc/lm_lookup.hip): same features as hash_fwd_f16, gathered one L2-sized address range at a time
    (levels >= halves_from in 2^split_log2 parts; 0 = halves).
    planar: out is float2 [16, n] (level-major) instead of f32 [n, 32].
    sample_list = (list i32 [>= count] of ascending sample indices, count i32 [1]): only those samples are looked up (nothing at all
    when the count is 0)."""
    lst, lst_n = sample_list if sample_list is not None else (None, None)
    _chk(lst, torch.int32, "sample_list"); _chk(lst_n, torch.int32, "sample_list count")
    _chk(table_f16, torch.float16, "table_f16"); _chk(table2_f16, torch.float16, "table2_f16"); _chk(prim_pool, torch.int32, "prim_pool")
    _chk(prim_pool2, torch.int32, "prim_pool2"); _chk(bias_pool, torch.float32, "bias_pool"); _chk(pts, torch.float32, "pts", 3)
    _chk(anchors, torch.int32, "anchors")
    n = pts.shape[0]
    if out is None:
        out = torch.empty((16, n, 2) if planar else (n, 32), dtype=torch.float32, device=pts.device)
    _chk(out, torch.float32, "out", 2 if planar else 32)
    if out.numel() != n * 32:
        raise _lib.GfnError("hash_fwd_lm: out must hold 32 features per sample")
    rc = _lib.lib().gfn_hash_fwd_lm(_ptr(table_f16), _ptr(table2_f16), _ptr(prim_pool), _ptr(prim_pool2), _ptr(bias_pool), _ptr(pts), _ptr(anchors),
                                    _anchor_stride(anchors, n), n, _ptr(n_dev), int(n_volumes), int(log2_T), int(halves_from), int(split_log2), int(bool(planar)), int(bool(accumulate)), _ptr(out),
                                    _ptr(lst), _ptr(lst_n), _opt(opts), _stream())
    _lib.check(rc, "gfn_hash_fwd_lm")
    return out


def table_to_f16(feat_pool, out=None):
    """f16 shadow of a master table (same flat indexing, incl. the level-overlap quirk)."""
    _chk(feat_pool, torch.float32, "feat_pool")
    if out is None:
        out = torch.empty(feat_pool.shape, dtype=torch.float16, device=feat_pool.device)
    _chk(out, torch.float16, "out")
    _lib.check(_lib.lib().gfn_table_to_f16(_ptr(feat_pool), _ptr(out), feat_pool.numel(), _stream()), "gfn_table_to_f16")
    return out


def hash_fwd_f16(table_f16, prim_pool, bias_pool, pts, anchors, n_volumes, log2_T, table2_f16=None, prim_pool2=None, out=None,
                 n_dev=None):
    """Lookup from f16 shadow table(s); table2_f16 fuses the focal stage's residual table (sum of both), hashed with
    prim_pool2 (its own primes) or, when None, with prim_pool."""
    _chk(prim_pool2, torch.int32, "prim_pool2")
    _chk(table_f16, torch.float16, "table_f16"); _chk(table2_f16, torch.float16, "table2_f16")
    _chk(prim_pool, torch.int32, "prim_pool"); _chk(bias_pool, torch.float32, "bias_pool"); _chk(pts, torch.float32, "pts", 3)
    _chk(anchors, anchors.dtype, "anchors")
    n = pts.shape[0]
    for t in (table_f16, table2_f16):
        if t is not None and t.numel() != 16 * (1 << log2_T) * 2:
            raise _lib.GfnError("f16 table has the wrong size")
    if prim_pool.numel() != 16 * n_volumes * 3:
        raise _lib.GfnError("prim_pool does not match n_volumes")
    stride = _anchor_stride(anchors, n)
    if out is None:
        out = torch.empty((n, 32), dtype=torch.float32, device=pts.device)
    _chk(out, torch.float32, "out", 32)
    rc = _lib.lib().gfn_hash_fwd_f16(_ptr(table_f16), _ptr(table2_f16), _ptr(prim_pool), _ptr(prim_pool2), _ptr(bias_pool), _ptr(pts),
                                     _ptr(anchors), int(anchors.dtype == torch.int64), stride, n, _ptr(n_dev), n_volumes,
                                     log2_T, _ptr(out), _stream())
    _lib.check(rc, "gfn_hash_fwd_f16")
    return out


def hash_bwd(prim_pool, bias_pool, pts, anchors, grad_out, n_volumes, log2_T, grad_pool=None, n_dev=None, grad_f16=False):
    """Scatter grad_out [n,32] into grad_pool [16*T,2] (accumulating; allocated zeroed if None).
    grad_f16: pool is float16 and keeps the reference's x128 scale (packed f16 atomics)."""
    _chk(prim_pool, torch.int32, "prim_pool"); _chk(bias_pool, torch.float32, "bias_pool")
    _chk(pts, torch.float32, "pts", 3); _chk(grad_out, torch.float32, "grad_out", 32)
    _chk(anchors, anchors.dtype, "anchors")
    n = pts.shape[0]
    if grad_out.shape[0] != n:
        raise _lib.GfnError("grad_out / pts disagree on n")
    stride = _anchor_stride(anchors, n)
    gdt = torch.float16 if grad_f16 else torch.float32
    if grad_pool is None:
        grad_pool = torch.zeros((16 * (1 << log2_T), 2), dtype=gdt, device=pts.device)
    _chk(grad_pool, gdt, "grad_pool")
    if grad_pool.numel() != 16 * (1 << log2_T) * 2:
        raise _lib.GfnError("grad_pool has the wrong size")
    rc = _lib.lib().gfn_hash_bwd(_ptr(prim_pool), _ptr(bias_pool), _ptr(pts), _ptr(anchors),
                                 int(anchors.dtype == torch.int64), stride, _ptr(grad_out), n, _ptr(n_dev), n_volumes,
                                 log2_T, _ptr(grad_pool), int(bool(grad_f16)), _stream())
    _lib.check(rc, "gfn_hash_bwd")
    return grad_pool


# ------------------------------------------------------------------------ sampler
def normalize_dirs(rays_d, out=None):
    _chk(rays_d, torch.float32, "rays_d", 3); _chk(out, torch.float32, "out", 3)
    out = torch.empty_like(rays_d) if out is None else out
    _lib.check(_lib.lib().gfn_normalize_dirs(_ptr(rays_d), _ptr(out), rays_d.shape[0], _stream()), "gfn_normalize_dirs")
    return out


def trace(rays_o, rays_d, nodes, near=0.01, far=1e8, max_hits=MAX_OCT_PER_RAY, hit_cap=None):
    """K3.  nodes: uint8 blob [n_nodes*128].  -> (oct_offsets i32[R+1], hit_idx i32[cap], hit_nf f32[cap,2], overflow i32[1])"""
    _chk(rays_o, torch.float32, "rays_o", 3); _chk(rays_d, torch.float32, "rays_d", 3)
    _chk(nodes, torch.uint8, "nodes")
    if nodes.numel() % TREE_NODE_BYTES:
        raise _lib.GfnError("nodes blob is not a multiple of 128 bytes")
    R = rays_o.shape[0]
    if hit_cap is None:
        hit_cap = max(R * max_hits, 1)
    dev = rays_o.device
    oct_offsets = torch.empty(R + 1, dtype=torch.int32, device=dev)
    hit_idx = torch.empty(hit_cap, dtype=torch.int32, device=dev)
    hit_nf = torch.empty((hit_cap, 2), dtype=torch.float32, device=dev)
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = _lib.lib().gfn_trace(_ptr(rays_o), _ptr(rays_d), R, _ptr(nodes), nodes.numel() // TREE_NODE_BYTES, near, far,
                              max_hits, _ptr(oct_offsets), _ptr(hit_idx), _ptr(hit_nf), hit_cap, _ptr(overflow),
                              _stream())
    _lib.check(rc, "gfn_trace")
    return oct_offsets, hit_idx, hit_nf, overflow


def march(rays_o, rays_d, noise, sample_l, scale_by_dis, oct_offsets, hit_idx, hit_nf, nodes, transes, pts_cap=None,
          want_world=True, want_dirs=True, table_space=False):
    """K4 -> dict of PACKED tensors (capacity pts_cap; live count = pts_offsets[-1] on device)."""
    _chk(rays_o, torch.float32, "rays_o", 3); _chk(rays_d, torch.float32, "rays_d", 3)
    _chk(noise, torch.float32, "noise"); _chk(oct_offsets, torch.int32, "oct_offsets")
    _chk(hit_idx, torch.int32, "hit_idx"); _chk(hit_nf, torch.float32, "hit_nf", 2)
    _chk(nodes, torch.uint8, "nodes"); _chk(transes, torch.uint8, "transes")
    R = rays_o.shape[0]
    if noise.numel() < MAX_SAMPLE_PER_RAY + R:
        raise _lib.GfnError(f"noise needs at least {MAX_SAMPLE_PER_RAY + R} entries")
    if transes.numel() % TRANS_INFO_BYTES:
        raise _lib.GfnError("transes blob is not a multiple of 576 bytes")
    if pts_cap is None:
        pts_cap = max(R * MAX_SAMPLE_PER_RAY, 1)
    dev = rays_o.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
    out = dict(
        pts_offsets=torch.empty(R + 1, dtype=torch.int32, device=dev),
        world_pts=f(pts_cap, 3) if want_world else None,
        warped_pts=f(pts_cap, 3),
        dirs=f(pts_cap, 3) if want_dirs else None,
        dists=f(pts_cap), ts=f(pts_cap),
        anchors=torch.empty((pts_cap, 3), dtype=torch.int32, device=dev),
        ray_id=torch.empty(pts_cap, dtype=torch.int32, device=dev),
        first_oct_dis=f(R),
        overflow=torch.zeros(1, dtype=torch.int32, device=dev),
    )
    rc = _lib.lib().gfn_march(_ptr(rays_o), _ptr(rays_d), _ptr(noise), R, float(sample_l), int(bool(scale_by_dis)),
                              int(bool(table_space)), _ptr(oct_offsets), _ptr(hit_idx), _ptr(hit_nf), _ptr(nodes), _ptr(transes),
                              _ptr(out["pts_offsets"]), _ptr(out["world_pts"]), _ptr(out["warped_pts"]),
                              _ptr(out["dirs"]), _ptr(out["dists"]), _ptr(out["ts"]), _ptr(out["anchors"]),
                              _ptr(out["ray_id"]), _ptr(out["first_oct_dis"]), pts_cap, _ptr(out["overflow"]),
                              _stream())
    _lib.check(rc, "gfn_march")
    return out


def tree_compact(nodes, out=None):
    """Checkpoint node blob -> one 64-byte line per node for the fast sampler."""
    _chk(nodes, torch.uint8, "nodes")
    n_nodes = nodes.numel() // TREE_NODE_BYTES
    if out is None:
        out = torch.empty(max(n_nodes, 1) * 64, dtype=torch.uint8, device=nodes.device)
    _lib.check(_lib.lib().gfn_tree_compact(_ptr(nodes), n_nodes, _ptr(out), _stream()), "gfn_tree_compact")
    return out


def sample_rays(rays_o, rays_d, noise, compact_nodes, transes, sample_l, scale_by_dis, near=0.01, far=1e8,
                max_hits=MAX_OCT_PER_RAY, pts_cap=None, want_world=False, want_dirs=False, table_space=True, workspace=None, out=None,
                tree_depth=0, opts=None):
    """Fast GetSamples: single-pass dense trace + march with a group of lanes per ray (a 16-lane row below 32 Ki rays, a quad from there on).  Same dict as march().  out: a dict an earlier call
    returned (same R / pts_cap / options): its tensors are written again instead of allocating new ones (static buffers of a
    captured step).  tree_depth: depth of the deepest node (see tree_depth()) or 0 for the reference's bound of 24."""
    _chk(rays_o, torch.float32, "rays_o", 3); _chk(rays_d, torch.float32, "rays_d", 3); _chk(noise, torch.float32, "noise")
    _chk(compact_nodes, torch.uint8, "compact_nodes"); _chk(transes, torch.uint8, "transes")
    R = rays_o.shape[0]
    if noise.numel() < MAX_SAMPLE_PER_RAY + R:
        raise _lib.GfnError(f"noise needs at least {MAX_SAMPLE_PER_RAY + R} entries")
    if pts_cap is None:
        pts_cap = max(R * MAX_SAMPLE_PER_RAY, 1)
    dev = rays_o.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
    if workspace is None or workspace["hit_idx"].numel() < R * max_hits:
        workspace = dict(hit_counts=torch.empty(max(R, 1), dtype=torch.int32, device=dev),
                         hit_idx=torch.empty(max(R * max_hits, 1), dtype=torch.int32, device=dev),
                         hit_nf=f(max(R * max_hits, 1), 2),
                         march_rec=f(max(R, 1) * MAX_SAMPLE_PER_RAY * MARCH_REC_FLOATS))  # the counting march's record of every sample
    if out is not None:
        if out["pts_offsets"].numel() != R + 1 or out["ts"].numel() != pts_cap or (want_world and out["world_pts"] is None):
            raise _lib.GfnError("sample_rays: `out` was made for another batch size / capacity / option set")
        out["overflow"].zero_()
        out["workspace"] = workspace
    else:
        out = dict(
            pts_offsets=torch.empty(R + 1, dtype=torch.int32, device=dev),
            world_pts=f(pts_cap, 3) if want_world else None, warped_pts=f(pts_cap, 3), dirs=f(pts_cap, 3) if want_dirs else None,
            dists=f(pts_cap), ts=f(pts_cap), anchors=torch.empty((pts_cap, 3), dtype=torch.int32, device=dev),
            ray_id=torch.empty(pts_cap, dtype=torch.int32, device=dev), first_oct_dis=f(R),
            overflow=torch.zeros(1, dtype=torch.int32, device=dev), workspace=workspace,
        )
    rc = _lib.lib().gfn_sample_rays(_ptr(rays_o), _ptr(rays_d), _ptr(noise), R, _ptr(compact_nodes), _ptr(transes), near, far,
                                    max_hits, float(sample_l), int(bool(scale_by_dis)), int(bool(table_space)),
                                    _ptr(workspace["hit_counts"]), _ptr(workspace["hit_idx"]), _ptr(workspace["hit_nf"]),
                                    _ptr(out["pts_offsets"]), _ptr(out["world_pts"]), _ptr(out["warped_pts"]), _ptr(out["dirs"]),
                                    _ptr(out["dists"]), _ptr(out["ts"]), _ptr(out["anchors"]), _ptr(out["ray_id"]),
                                    _ptr(out["first_oct_dis"]), pts_cap, _ptr(out["overflow"]), _ptr(workspace.get("march_rec")), int(tree_depth),
                                    _opt(opts), _stream())
    _lib.check(rc, "gfn_sample_rays")
    return out


def sample_rays_record(rays_o, rays_d, noise, compact_nodes, transes, sample_l, scale_by_dis, near=0.01, far=1e8, max_hits=MAX_OCT_PER_RAY,
                       workspace=None, out=None, tree_depth=0, pool_slots_per_ray=0, opts=None):
    """gfn_sample_rays_record: traversal + counting march only -> dict(counts i32 [R], first_oct_dis [R], march_rec f32 [R * 1024 * 8],
    overflow, workspace): the samples stay where the march recorded them (slot k of ray r at (r * 1024 + k) * MARCH_REC_FLOATS), for
    fused_render's record form.  out: the dict of an earlier call with the same R, written again in place.
    pool_slots_per_ray > 0: the PAGED form -- march_rec is a pool of R * pool_slots_per_ray / 32 pages of 32 slots, page_table i32
    [R * 32] says where each ray's pages are: sized by the samples the rays have, not by the 1024 they may have (7 KB instead of 32 KB
    per ray at 224).  When the pool runs out, overflow gets bit REC_POOL_OVERFLOW: sample again with a larger pool."""
    _chk(rays_o, torch.float32, "rays_o", 3); _chk(rays_d, torch.float32, "rays_d", 3); _chk(noise, torch.float32, "noise")
    _chk(compact_nodes, torch.uint8, "compact_nodes"); _chk(transes, torch.uint8, "transes")
    R = rays_o.shape[0]
    if noise.numel() < MAX_SAMPLE_PER_RAY + R:
        raise _lib.GfnError(f"noise needs at least {MAX_SAMPLE_PER_RAY + R} entries")
    dev = rays_o.device
    if workspace is None or workspace["hit_idx"].numel() < R * max_hits:
        workspace = dict(hit_counts=torch.empty(max(R, 1), dtype=torch.int32, device=dev),
                         hit_idx=torch.empty(max(R * max_hits, 1), dtype=torch.int32, device=dev),
                         hit_nf=torch.empty((max(R * max_hits, 1), 2), dtype=torch.float32, device=dev))
    if out is not None:
        if out["counts"].numel() != R:
            raise _lib.GfnError("sample_rays_record: `out` was made for another batch size")
        out["overflow"].zero_()
        out["workspace"] = workspace
    elif pool_slots_per_ray > 0:
        n_pages = max((R * int(pool_slots_per_ray) + REC_PAGE - 1) // REC_PAGE, 1)
        out = dict(counts=torch.empty(R, dtype=torch.int32, device=dev), first_oct_dis=torch.empty(R, dtype=torch.float32, device=dev),
                   march_rec=torch.empty(n_pages * REC_PAGE * MARCH_REC_FLOATS, dtype=torch.float32, device=dev),
                   page_table=torch.empty(max(R, 1) * (MAX_SAMPLE_PER_RAY // REC_PAGE), dtype=torch.int32, device=dev),
                   page_counter=torch.zeros(1, dtype=torch.int32, device=dev), n_pages=n_pages,
                   overflow=torch.zeros(1, dtype=torch.int32, device=dev), workspace=workspace)
    else:
        out = dict(counts=torch.empty(R, dtype=torch.int32, device=dev), first_oct_dis=torch.empty(R, dtype=torch.float32, device=dev),
                   march_rec=torch.empty(max(R, 1) * MAX_SAMPLE_PER_RAY * MARCH_REC_FLOATS, dtype=torch.float32, device=dev),
                   overflow=torch.zeros(1, dtype=torch.int32, device=dev), workspace=workspace)
    rc = _lib.lib().gfn_sample_rays_record(_ptr(rays_o), _ptr(rays_d), _ptr(noise), R, _ptr(compact_nodes), _ptr(transes), near, far, max_hits,
                                           float(sample_l), int(bool(scale_by_dis)), _ptr(workspace["hit_counts"]), _ptr(workspace["hit_idx"]),
                                           _ptr(workspace["hit_nf"]), _ptr(out["counts"]), _ptr(out["first_oct_dis"]), _ptr(out["overflow"]),
                                           _ptr(out["march_rec"]), _ptr(out.get("page_table")), _ptr(out.get("page_counter")), int(out.get("n_pages", 0)),
                                           int(tree_depth), _opt(opts), _stream())
    _lib.check(rc, "gfn_sample_rays_record")
    return out


def tree_depth(nodes):
    """Depth of the deepest node of a node blob (uint8 [n,128] device tensor; root = 0): parent links walked level by level (a tree
    changes a few times per training run; this synchronises)."""
    n = nodes.shape[0]
    parent = nodes.view(-1).view(torch.int64).view(n, 16)[:, 2].contiguous()  # gfn_tree_node.parent (byte 16)
    cur, depth = parent.clone(), 0
    for depth in range(0, 64):
        alive = cur >= 0
        if not bool(alive.any()):
            break
        cur = torch.where(alive, parent[cur.clamp(min=0)], cur)
    return depth


def samples_to_dense(s, n_rays):
    """Packed sampler output -> the reference's dense GetSamples() list (PersSampler_cuda.cu:467-476)."""
    dev = s["pts_offsets"].device
    M = MAX_SAMPLE_PER_RAY
    f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)  # noqa: E731
    d_world, d_warped, d_dirs = f(n_rays, M, 3), f(n_rays, M, 3), f(n_rays, M, 3)
    d_dists, d_ts = f(n_rays, M), f(n_rays, M)
    d_anchors = torch.empty((n_rays, M, 3), dtype=torch.int64, device=dev)
    d_se = torch.empty((n_rays, 2), dtype=torch.int64, device=dev)
    rc = _lib.lib().gfn_samples_to_dense(_ptr(s["pts_offsets"]), n_rays, _ptr(s["world_pts"]), _ptr(s["warped_pts"]),
                                         _ptr(s["dirs"]), _ptr(s["dists"]), _ptr(s["ts"]), _ptr(s["anchors"]),
                                         _ptr(d_world), _ptr(d_warped), _ptr(d_dirs), _ptr(d_dists), _ptr(d_ts),
                                         _ptr(d_anchors), _ptr(d_se), _stream())
    _lib.check(rc, "gfn_samples_to_dense")
    return [d_world, d_warped, d_dirs, d_dists, d_ts, d_anchors, d_se, s["first_oct_dis"].view(n_rays, 1)]


def update_nodes(pts_offsets, anchors, weights, alphas, weight_stats, alpha_stats, visit_cnt, nodes, reduce_max=None):
    """UpdateOctNodes' device part (K5, statistics, K6).  reduce_max: callable taking a list of int64 tensors and replacing each
    with its elementwise maximum over the ranks (gfnerf_amd.dist.allreduce_max_) -- run between K5 and the statistics update, it
    gives every ray-data-parallel replica the tree ONE process would have computed from all ranks' rays."""
    _chk(pts_offsets, torch.int32, "pts_offsets"); _chk(anchors, torch.int32, "anchors", 3)
    _chk(weights, torch.float32, "weights"); _chk(alphas, torch.float32, "alphas")
    for t, nm in ((weight_stats, "weight_stats"), (alpha_stats, "alpha_stats"), (visit_cnt, "visit_cnt")):
        _chk(t, torch.int64, nm)
    _chk(nodes, torch.uint8, "nodes")
    n_nodes = nodes.numel() // TREE_NODE_BYTES
    if weight_stats.numel() != n_nodes or alpha_stats.numel() != n_nodes or visit_cnt.numel() != n_nodes:
        raise _lib.GfnError("stat tensors must have n_nodes entries")
    scratch = torch.empty(3 * n_nodes, dtype=torch.int64, device=nodes.device)
    L = _lib.lib()
    rc = L.gfn_update_nodes_mark(_ptr(pts_offsets), pts_offsets.numel() - 1, _ptr(anchors), _ptr(weights), _ptr(alphas), n_nodes,
                                 _ptr(visit_cnt), _ptr(scratch), _stream())
    _lib.check(rc, "gfn_update_nodes_mark")
    if reduce_max is not None:
        reduce_max([scratch, visit_cnt])
    rc = L.gfn_update_nodes_apply(n_nodes, _ptr(scratch), _ptr(weight_stats), _ptr(alpha_stats), _ptr(nodes), _stream())
    _lib.check(rc, "gfn_update_nodes_apply")


def query_tree_nodes_centers(nodes, anchors):
    """PersSampler::QueryTreeNodeCenters (K12): anchors i64 [n,1] (node indices) -> centers f32 [n,3], zeros where out of range."""
    _chk(nodes, torch.uint8, "nodes"); _chk(anchors, torch.int64, "anchors")
    n = anchors.shape[0]
    if anchors.numel() != n:
        raise _lib.GfnError("anchors must be [n, 1]")
    out = torch.zeros((n, 3), dtype=torch.float32, device=nodes.device)
    rc = _lib.lib().gfn_query_tree_nodes_centers(_ptr(nodes), nodes.numel() // TREE_NODE_BYTES, _ptr(anchors), n, _ptr(out), _stream())
    _lib.check(rc, "gfn_query_tree_nodes_centers")
    return out


def mark_invisible_nodes(nodes, intri, w2c, bounds):
    """K7: trans_idx = -1 for nodes no camera sees (in place on the device blob)."""
    _chk(nodes, torch.uint8, "nodes"); _chk(intri, torch.float32, "intri", 3); _chk(w2c, torch.float32, "w2c", 4)
    _chk(bounds, torch.float32, "bounds", 2)
    n_cams = intri.shape[0]
    if tuple(intri.shape) != (n_cams, 3, 3) or tuple(w2c.shape) != (n_cams, 3, 4) or tuple(bounds.shape) != (n_cams, 2):
        raise _lib.GfnError("expected intri [n,3,3], w2c [n,3,4], bounds [n,2]")
    rc = _lib.lib().gfn_mark_invisible_nodes(_ptr(nodes), nodes.numel() // TREE_NODE_BYTES, _ptr(intri), _ptr(w2c), _ptr(bounds), n_cams,
                                             _stream())
    _lib.check(rc, "gfn_mark_invisible_nodes")


def octree_proc_host(nodes, weight_stats, alpha_stats, visit_cnt, compact=True, subdivide=False, brute_force=False):
    """PersOctree::ProcOctree on host numpy arrays -> (nodes uint8 blob, weight_stats i64, alpha_stats i64) of the new tree."""
    nodes = np.ascontiguousarray(np.asarray(nodes).view(np.uint8).reshape(-1))
    n = nodes.size // TREE_NODE_BYTES
    w, a, v = (np.ascontiguousarray(x, np.int64) for x in (weight_stats, alpha_stats, visit_cnt))
    if not (w.size == n and a.size == n and v.size == n):
        raise _lib.GfnError("stat arrays must have n_nodes entries")
    out_n, out_w, out_a, m = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
    L = _lib.lib()
    vp = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = L.gfn_octree_proc_host(vp(nodes), n, vp(w), vp(a), vp(v), int(compact), int(subdivide), int(brute_force), C.byref(out_n),
                                C.byref(m), C.byref(out_w), C.byref(out_a))
    _lib.check(rc, "gfn_octree_proc_host")
    try:
        k = m.value
        new_nodes = np.ctypeslib.as_array(C.cast(out_n, C.POINTER(C.c_uint8)), shape=(k * TREE_NODE_BYTES,)).copy()
        new_w = np.ctypeslib.as_array(C.cast(out_w, C.POINTER(C.c_int64)), shape=(k,)).copy()
        new_a = np.ctypeslib.as_array(C.cast(out_a, C.POINTER(C.c_int64)), shape=(k,)).copy()
    finally:
        for ptr in (out_n, out_w, out_a):
            L.gfn_free_host(ptr)
    return new_nodes, new_w, new_a


def octree_proc(nodes, weight_stats, alpha_stats, visit_cnt, subdivide=False, brute_force=False):
    """PersOctree::ProcOctree on the device (compaction, optionally subdivision): device tensors in, device tensors out
    -> (nodes u8 blob, weight_stats i64, alpha_stats i64) of the new tree.  The only thing that crosses to the host is the
    new node count + a status word (16 bytes), read after the launch to size the views."""
    _chk(nodes, torch.uint8, "nodes")
    for t, nm in ((weight_stats, "weight_stats"), (alpha_stats, "alpha_stats"), (visit_cnt, "visit_cnt")):
        _chk(t, torch.int64, nm)
    n = nodes.numel() // TREE_NODE_BYTES
    if not (weight_stats.numel() == n and alpha_stats.numel() == n and visit_cnt.numel() == n) or n < 1:
        raise _lib.GfnError("octree_proc: stat tensors must have n_nodes entries")
    dev = nodes.device
    cap = 9 * n if subdivide else n
    L = _lib.lib()
    ws = torch.empty(L.gfn_octree_proc_workspace(n, int(subdivide)), dtype=torch.uint8, device=dev)
    out = torch.empty(cap * TREE_NODE_BYTES, dtype=torch.uint8, device=dev)
    w_out = torch.empty(cap, dtype=torch.int64, device=dev)
    a_out = torch.empty(cap, dtype=torch.int64, device=dev)
    head = torch.zeros(2, dtype=torch.int64, device=dev)  # [new count, status (low 32 bits)]
    rc = L.gfn_octree_proc(_ptr(nodes), n, None, _ptr(weight_stats), _ptr(alpha_stats), _ptr(visit_cnt), int(subdivide), int(brute_force),
                           _ptr(out), _ptr(w_out), _ptr(a_out), cap, _ptr(head), C.c_void_p(head.data_ptr() + 8), _ptr(ws), _stream())
    _lib.check(rc, "gfn_octree_proc")
    m, status = (int(v) for v in head.tolist())
    status &= 0xFFFFFFFF
    if status == 1:
        raise _lib.GfnError("gfn_octree_proc: node blob is not a tree rooted at node 0 (parent / child links disagree)")
    if status != 0 or m < 1:
        raise _lib.GfnError(f"gfn_octree_proc: status {status}, {m} nodes")
    # exact-size tensors of their own (the buffers above hold room for the worst case -- 9 n nodes with subdivision -- and what lies
    # behind the new tree is uninitialised: a view would carry all of it into every checkpoint and keep it resident)
    return out[: m * TREE_NODE_BYTES].clone(), w_out[:m].clone(), a_out[:m].clone()


def generate_rays(c2w, intr, cam_idx=None, coords=None, cam0=0, first_pixel=0, width=0, n=None, out=None, want_extras=True):
    """Cameras.generate_rays (perspective, no distortion).  Explicit (cam_idx i32 [n], coords f32 [n,2] = (y, x)) pairs, or the
    raster run [first_pixel, first_pixel + n) of camera cam0's image of the given width.
    -> dict(origins, directions, lookat [n,3], pixel_area, dir_norm [n]); `out` may pre-allocate origins / directions."""
    _chk(c2w, torch.float32, "c2w", 4); _chk(intr, torch.float32, "intr", 4); _chk(cam_idx, torch.int32, "cam_idx")
    _chk(coords, torch.float32, "coords", 2)
    if coords is not None:
        n = coords.shape[0]
        if cam_idx is None or cam_idx.numel() != n:
            raise _lib.GfnError("explicit coords need one camera index per ray")
    elif n is None or width <= 0:
        raise _lib.GfnError("raster mode needs n and width")
    n_cams = c2w.shape[0]
    if tuple(c2w.shape) != (n_cams, 3, 4) or tuple(intr.shape) != (n_cams, 4) or not (0 <= cam0 < n_cams):
        raise _lib.GfnError("expected c2w [n_cams,3,4], intr [n_cams,4] and a valid cam0")
    dev = c2w.device
    o = out or {}
    res = dict(origins=o.get("origins", None), directions=o.get("directions", None))
    for k in ("origins", "directions"):
        if res[k] is None:
            res[k] = torch.empty((n, 3), dtype=torch.float32, device=dev)
        _chk(res[k], torch.float32, k, 3)
    res["lookat"] = torch.empty((n, 3), dtype=torch.float32, device=dev) if want_extras else None
    res["pixel_area"] = torch.empty(n, dtype=torch.float32, device=dev) if want_extras else None
    res["dir_norm"] = torch.empty(n, dtype=torch.float32, device=dev) if want_extras else None
    rc = _lib.lib().gfn_generate_rays(_ptr(c2w), _ptr(intr), _ptr(cam_idx), int(cam0), _ptr(coords), int(first_pixel), int(width), n,
                                      _ptr(res["origins"]), _ptr(res["directions"]), _ptr(res["lookat"]), _ptr(res["pixel_area"]),
                                      _ptr(res["dir_norm"]), _stream())
    _lib.check(rc, "gfn_generate_rays")
    return res


# ------------------------------------------------------------------ batch assembly
ERRMAP_BLOCK, PIXEL_CHUNK, PIXEL_ATTEMPTS = 4096, 1 << 24, 64


def pixel_uniform(u, n_images, height, width):
    """PixelSampler.sample_method without a mask: u f32 [n,3] in [0,1) -> lin i64 [n] = (c*H + y)*W + x."""
    _chk(u, torch.float32, "u", 3)
    lin = torch.empty(u.shape[0], dtype=torch.int64, device=u.device)
    _lib.check(_lib.lib().gfn_pixel_uniform(_ptr(u), u.shape[0], int(n_images), int(height), int(width), _ptr(lin), _stream()), "gfn_pixel_uniform")
    return lin


def pixel_collate(lin, images, image_idx=None, rel_cam_idx=None, shape=None, want_coords=True):
    """collate_image_dataset_batch for the pixels lin i64 [n] of the cached stack images [n_images,H,W,3] (f32 or u8; None
    with shape=(n_images,H,W) skips the colour gather) -> dict(indices i32 [n,3] (abs cam, y, x), slot, rel_camera_indices
    i32 [n], image f32 [n,3], coords f32 [n,2])."""
    _chk(lin, torch.int64, "lin"); _chk(image_idx, torch.int32, "image_idx"); _chk(rel_cam_idx, torch.int32, "rel_cam_idx")
    if images is not None:
        if images.dtype not in (torch.float32, torch.uint8):
            raise _lib.GfnError("pixel_collate: images must be float32 or uint8")
        _chk(images, images.dtype, "images", 3)
        if images.dim() != 4:
            raise _lib.GfnError("pixel_collate: images must be [n_images,H,W,3]")
        shape = images.shape[:3]
    if shape is None:
        raise _lib.GfnError("pixel_collate: need images or shape")
    n_img, H, W = (int(v) for v in shape)
    for t, nm in ((image_idx, "image_idx"), (rel_cam_idx, "rel_cam_idx")):
        if t is not None and t.numel() != n_img:
            raise _lib.GfnError(f"pixel_collate: {nm} must have one entry per cached image")
    n, dev = lin.numel(), lin.device
    out = dict(indices=torch.empty((n, 3), dtype=torch.int32, device=dev), slot=torch.empty(n, dtype=torch.int32, device=dev),
               rel_camera_indices=torch.empty(n, dtype=torch.int32, device=dev),
               image=torch.empty((n, 3), dtype=torch.float32, device=dev) if images is not None else None,
               coords=torch.empty((n, 2), dtype=torch.float32, device=dev) if want_coords else None)
    rc = _lib.lib().gfn_pixel_collate(_ptr(lin), n, n_img, H, W, _ptr(images), int(images is not None and images.dtype == torch.uint8),
                                      _ptr(image_idx), _ptr(rel_cam_idx), _ptr(out["indices"]), _ptr(out["slot"]),
                                      _ptr(out["rel_camera_indices"]), _ptr(out["image"]), _ptr(out["coords"]), _stream())
    _lib.check(rc, "gfn_pixel_collate")
    return out


def errmap_block_sums(error_map, out=None):
    """f64 partial sums per ERRMAP_BLOCK pixels of error_map f32 [n_images,H,W(,1)]."""
    _chk(error_map, torch.float32, "error_map")
    n = error_map.numel()
    nb = (n + ERRMAP_BLOCK - 1) // ERRMAP_BLOCK
    if out is None:
        out = torch.empty(nb, dtype=torch.float64, device=error_map.device)
    _chk(out, torch.float64, "block_sums")
    if out.numel() != nb:
        raise _lib.GfnError("errmap_block_sums: wrong number of partial sums")
    _lib.check(_lib.lib().gfn_errmap_block_sums(_ptr(error_map), n, _ptr(out), _stream()), "gfn_errmap_block_sums")
    return out


def errmap_update(error_map, block_sums, indices, pred, gt):
    """CacheDataloader._update_error_map with error = |gt - pred|.sum(-1): error_map [n_images,H,W(,1)] in place."""
    _chk(error_map, torch.float32, "error_map"); _chk(block_sums, torch.float64, "block_sums"); _chk(indices, torch.int32, "indices", 3)
    _chk(pred, torch.float32, "pred", 3); _chk(gt, torch.float32, "gt", 3)
    n = indices.shape[0]
    if pred.shape[0] != n or gt.shape[0] != n or error_map.dim() < 3:
        raise _lib.GfnError("errmap_update: one (c, y, x) row per ray and an error map [n_images,H,W]")
    n_img, H, W = error_map.shape[:3]
    if block_sums is not None and block_sums.numel() != (error_map.numel() + ERRMAP_BLOCK - 1) // ERRMAP_BLOCK:
        raise _lib.GfnError("errmap_update: block_sums does not belong to this error map")
    _lib.check(_lib.lib().gfn_errmap_update(_ptr(indices), n, _ptr(pred), _ptr(gt), n_img, H, W, _ptr(error_map), _ptr(block_sums), _stream()),
               "gfn_errmap_update")


def errmap_sample(error_map, block_sums, bits, n_weighted, workspace=None):
    """ErrorPixelSampler.sample_method: bits i64 [n] random -> (lin i64 [n], status i32 [2], workspace)."""
    _chk(error_map, torch.float32, "error_map"); _chk(block_sums, torch.float64, "block_sums"); _chk(bits, torch.int64, "bits")
    n, npix, dev = bits.numel(), error_map.numel(), error_map.device
    nb = (npix + ERRMAP_BLOCK - 1) // ERRMAP_BLOCK
    if block_sums.numel() != nb:
        raise _lib.GfnError("errmap_sample: block_sums does not belong to this error map")
    slots = _lib.lib().gfn_errmap_table_slots(max(int(n_weighted), n - int(n_weighted)))
    ws = workspace or {}
    if ws.get("key") != (n, nb, slots):
        ws = dict(key=(n, nb, slots), prefix=torch.empty(nb + 1, dtype=torch.float64, device=dev),
                  cand=torch.empty(n, dtype=torch.int64, device=dev), pending=torch.empty(max(n, 1), dtype=torch.int32, device=dev),
                  table=torch.empty(slots * 2, dtype=torch.int64, device=dev) if slots * 16 > 128 * 1024 else None)
    lin = torch.empty(n, dtype=torch.int64, device=dev)
    status = torch.empty(2, dtype=torch.int32, device=dev)
    rc = _lib.lib().gfn_errmap_sample(_ptr(error_map), npix, _ptr(block_sums), _ptr(bits), n, int(n_weighted), _ptr(ws["prefix"]), _ptr(ws["cand"]),
                                      _ptr(ws["pending"]), _ptr(ws["table"]), _ptr(lin), _ptr(status), _stream())
    _lib.check(rc, "gfn_errmap_sample")
    return lin, status, ws


# ---------------------------------------------------------------------------- MLP
ACT = {"none": 0, "relu": 1, "sigmoid": 2}


def linear_fwd(x, W, b, act="none", out=None, n_dev=None, in_dim=None):
    """y = act(x[:, :in_dim] W^T + b).  x [n, ldx] f32, W [out,in] f32."""
    _chk(x, torch.float32, "x"); _chk(W, torch.float32, "W"); _chk(b, torch.float32, "b")
    n, ldx = x.shape
    out_dim, w_in = W.shape
    in_dim = w_in if in_dim is None else in_dim
    if in_dim != w_in or in_dim > ldx:
        raise _lib.GfnError(f"linear_fwd: W is {tuple(W.shape)} but x has {ldx} columns")
    if out is None:
        out = torch.empty((n, out_dim), dtype=torch.float32, device=x.device)
    _chk(out, torch.float32, "out")
    rc = _lib.lib().gfn_linear_fwd(_ptr(x), ldx, _ptr(W), _ptr(b), n, _ptr(n_dev), in_dim, out_dim, ACT[act], _ptr(out),
                                   out.shape[1], _stream())
    _lib.check(rc, "gfn_linear_fwd")
    return out


def linear_bwd(x, W, y, dy, act="none", need_dx=True, dW=None, db=None, dx=None, n_dev=None, need_dw=True, in_dim=None):
    """-> (dx or None, dW, db); dW/db accumulate into the given tensors (allocated zeroed if None);
    need_dw=False skips the weight/bias gradient (frozen MLPs of the focal stage)."""
    for t, nm in ((x, "x"), (W, "W"), (y, "y"), (dy, "dy")):
        _chk(t, torch.float32, nm)
    n, ldx = x.shape
    out_dim, w_in = W.shape
    in_dim = w_in if in_dim is None else in_dim
    if in_dim != w_in or in_dim > ldx:
        raise _lib.GfnError(f"linear_bwd: W is {tuple(W.shape)} but x has {ldx} columns")
    if need_dw and dW is None:
        dW = torch.zeros_like(W)
    if need_dw and db is None:
        db = torch.zeros(out_dim, dtype=torch.float32, device=x.device)
    if not need_dw:
        dW = db = None
    _chk(dW, torch.float32, "dW"); _chk(db, torch.float32, "db")
    if need_dx and dx is None:
        dx = torch.empty((n, ldx), dtype=torch.float32, device=x.device)
        if ldx > in_dim:
            dx[:, in_dim:].zero_()
    rc = _lib.lib().gfn_linear_bwd(_ptr(x), ldx, _ptr(W), _ptr(y), y.shape[1], _ptr(dy), dy.shape[1], n, _ptr(n_dev),
                                   in_dim, out_dim, ACT[act], _ptr(dx) if need_dx else None,
                                   dx.shape[1] if need_dx else 0, _ptr(dW), _ptr(db), _stream())
    _lib.check(rc, "gfn_linear_bwd")
    return (dx if need_dx else None), dW, db


def _weight_ptrs(layers):
    """10 device pointers {W,b} x {base0, base1, head0, head1, head2} as a C array (kept alive by the caller)."""
    shapes = [(128, 32), (16, 128), (128, 63), (128, 128), (3, 128)]
    if len(layers) != 5:
        raise _lib.GfnError("expected 5 (W, b) layers")
    ptrs = []
    for (W, b), shp in zip(layers, shapes):
        _chk(W, torch.float32, "W"); _chk(b, torch.float32, "b")
        if tuple(W.shape) != shp or b.numel() != shp[0]:
            raise _lib.GfnError(f"fused field kernels need hidden_dim 128 layers, got {tuple(W.shape)} (expected {shp})")
        ptrs += [W.data_ptr(), b.data_ptr()]
    return (C.c_void_p * 10)(*ptrs)


FIELD_X3_FWD_BYTES, FIELD_X3_BWD_BYTES = 149248, 131072
FIELD_X3_FWD_PLANE_LD, FIELD_X3_BWD_PLANE_LD = 480, 408
FIELD_X3_BWD_PLANE_LD_RNE = 536  # the backward's one-plane buffer: + dza1's lo' plane (include/gfnerf_amd.h)
# column ranges of the planes (include/gfnerf_amd.h)
FWD_PLANE = dict(feat=(0, 32), a1=(32, 128), xp=(160, 64), c1=(224, 128), c2=(352, 128))
BWD_PLANE = dict(dz3=(0, 8), dh=(8, 16), dz2=(24, 128), dz1=(152, 128), dza1=(280, 128), dza1_lo=(408, 128))


def _plane_cap(ph, pl, ld, n):
    """Plane buffers are flat f16 tensors of cap * ld elements, plane-major (see include/gfnerf_amd.h).  pl None: ONE plane per
    operand, its values rounded to f16 once, to nearest (the weight gradient's operands only: csrc/x3_device.h, PL_RNE)."""
    if ph is None and pl is None:
        return 0
    if ph is None:
        raise _lib.GfnError("a lo' plane buffer needs its hi buffer")
    for t in (ph, pl):
        _chk(t, torch.float16, "planes")
    if (pl is not None and ph.numel() != pl.numel()) or ph.numel() % ld:
        raise _lib.GfnError("plane halves differ in size or are not a multiple of the row width")
    cap = ph.numel() // ld
    if cap < n or cap % 32:
        raise _lib.GfnError("plane capacity must cover the sample buffers and be a multiple of 32")
    return cap


def plane_flat(buf, ld, name, table):
    """Plane `name` of a plane-major buffer as the flat tile-major array the kernels address: (flat f16 [cap * width], width)."""
    cap = buf.numel() // ld
    c0, w = table[name]
    return buf.view(-1)[cap * c0: cap * (c0 + w)], w


def plane_view(buf, ld, name, table):
    """The [cap, width] rows of plane `name` (a strided view of the tile-major storage [cap/32][width/8][32][8]): for reading
    planes back in tests and tools."""
    flat, w = plane_flat(buf, ld, name, table)
    cap = flat.numel() // w
    return flat.view(cap // 32, w // 8, 32, 8).permute(0, 2, 1, 3).reshape(cap, w)


def field_pack_x3(layers, fwd_image=None, bwd_image=None, want_fwd=True, want_bwd=True):
    """Build the split-f16 weight images of the x3 decoder kernels (to be redone whenever the weights change)."""
    dev = layers[0][0].device
    if want_fwd and fwd_image is None:
        fwd_image = torch.empty(FIELD_X3_FWD_BYTES, dtype=torch.uint8, device=dev)
    if want_bwd and bwd_image is None:
        bwd_image = torch.empty(FIELD_X3_BWD_BYTES, dtype=torch.uint8, device=dev)
    wp = _weight_ptrs(layers)
    rc = _lib.lib().gfn_field_pack_x3(wp, _ptr(fwd_image) if want_fwd else None, _ptr(bwd_image) if want_bwd else None, _stream())
    _lib.check(rc, "gfn_field_pack_x3")
    return fwd_image, bwd_image


def field_outputs(n, device, save=True):
    """Uninitialised outputs of field_fwd for n sample slots (to be passed as `out`)."""
    return dict(sigma=torch.empty(n, dtype=torch.float32, device=device), rgb=torch.empty((n, 3), dtype=torch.float32, device=device),
                h0=torch.empty(n, dtype=torch.float32, device=device) if save else None,
                masks=torch.empty((n, 6), dtype=torch.int64, device=device) if save else None)


def field_fwd(feat, ray_id, sh, emb, cam_idx, layers, n_dev=None, save=False, precision="x3", image=None, planes=None, sample_list=None,
              out=None, opts=None):
    """Fused decoder forward -> dict(sigma [n], rgb [n,3], h0 [n]|None, masks [n,6] int64|None).
    precision "x3": split-f16 MFMA (f32 accuracy, ~5x the f32 matrix rate; `image` = a current forward image from
    field_pack_x3, built here when absent); "f32": exact f32 MFMA.
    feat: f32 [n,32], or the level-major [16,n,2] of hash_fwd_lm(planar=True) (x3 only).
    sample_list (x3 only) = (list i32, count i32 [1]): decode only those samples; the others keep what `out` (the dict an earlier
    call returned, or field_outputs(), written again) holds."""
    lst, lst_n = sample_list if sample_list is not None else (None, None)
    _chk(lst, torch.int32, "sample_list"); _chk(lst_n, torch.int32, "sample_list count")
    if (lst is not None or out is not None) and precision != "x3":
        raise _lib.GfnError("field_fwd: sample lists need the x3 decoder")
    planar = feat.dim() == 3
    _chk(feat, torch.float32, "feat", 2 if planar else 32); _chk(ray_id, torch.int32, "ray_id"); _chk(sh, torch.float32, "sh", 16)
    _chk(emb, torch.float32, "emb", 32); _chk(cam_idx, torch.int32, "cam_idx")
    if planar and (feat.shape[0] != 16 or precision != "x3"):
        raise _lib.GfnError("field_fwd: level-major features are [16,n,2] and need the x3 decoder")
    n = feat.shape[1] if planar else feat.shape[0]
    dev = feat.device
    if out is not None:
        sigma, rgb, h0, masks = out["sigma"], out["rgb"], out["h0"], out["masks"]
        if sigma.numel() != n or (save and masks is None):
            raise _lib.GfnError("field_fwd: `out` was made for another batch / option set")
    else:
        sigma = torch.empty(n, dtype=torch.float32, device=dev)
        rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
        h0 = torch.empty(n, dtype=torch.float32, device=dev) if save else None
        masks = torch.empty((n, 6), dtype=torch.int64, device=dev) if save else None
    if precision == "x3":
        if image is None:
            image, _ = field_pack_x3(layers, want_bwd=False)
        ph, pl = planes if planes is not None else (None, None)
        cap = _plane_cap(ph, pl, FIELD_X3_FWD_PLANE_LD, n)  # split-f16 activation planes for the weight gradients (global stage)
        rc = _lib.lib().gfn_field_fwd_x3(_ptr(feat), _ptr(ray_id), _ptr(sh), _ptr(emb), _ptr(cam_idx), _ptr(image), n, _ptr(n_dev),
                                         _ptr(sigma), _ptr(rgb), _ptr(h0), _ptr(masks), _ptr(ph), _ptr(pl), cap, n if planar else 0,
                                         _ptr(lst), _ptr(lst_n), _opt(opts), _stream())
        _lib.check(rc, "gfn_field_fwd_x3")
    elif precision == "f32":
        wp = _weight_ptrs(layers)
        rc = _lib.lib().gfn_field_fwd(_ptr(feat), _ptr(ray_id), _ptr(sh), _ptr(emb), _ptr(cam_idx), wp, n, _ptr(n_dev), _ptr(sigma),
                                      _ptr(rgb), _ptr(h0), _ptr(masks), _stream())
        _lib.check(rc, "gfn_field_fwd")
    else:
        raise _lib.GfnError(f"unknown decoder precision {precision!r}")
    return dict(sigma=sigma, rgb=rgb, h0=h0, masks=masks)


def field_bwd_dx(d_sigma, d_rgb, rgb, h0, masks, layers, n_dev=None, precision="x3", image=None, opts=None):
    """Fused decoder backward w.r.t. the hash features (decoder frozen) -> dfeat [n,32]."""
    _chk(d_sigma, torch.float32, "d_sigma"); _chk(d_rgb, torch.float32, "d_rgb", 3); _chk(rgb, torch.float32, "rgb", 3)
    _chk(h0, torch.float32, "h0"); _chk(masks, torch.int64, "masks", 6)
    n = d_sigma.shape[0]
    dfeat = torch.empty((n, 32), dtype=torch.float32, device=d_sigma.device)
    if precision == "x3":
        if image is None:
            _, image = field_pack_x3(layers, want_fwd=False)
        rc = _lib.lib().gfn_field_bwd_dx_x3(_ptr(d_sigma), _ptr(d_rgb), _ptr(rgb), _ptr(h0), _ptr(masks), _ptr(image), n, _ptr(n_dev),
                                            _ptr(dfeat), _opt(opts), _stream())
        _lib.check(rc, "gfn_field_bwd_dx_x3")
    elif precision == "f32":
        wp = _weight_ptrs(layers)
        rc = _lib.lib().gfn_field_bwd_dx(_ptr(d_sigma), _ptr(d_rgb), _ptr(rgb), _ptr(h0), _ptr(masks), wp, n, _ptr(n_dev), _ptr(dfeat),
                                         _stream())
        _lib.check(rc, "gfn_field_bwd_dx")
    else:
        raise _lib.GfnError(f"unknown decoder precision {precision!r}")
    return dfeat


def split_f16(x):
    """x f32 -> (hi, lo') f16 planes with x ~= hi + lo' * 2^-11 (the operand format of the x3 kernels)."""
    hi = x.to(torch.float16)
    lo = ((x - hi.float()) * 2048.0).to(torch.float16)
    return hi, lo


def wgrad_x3(dz, a, dW, db=None, n_dev=None, gmax=None, m_valid=None, k_valid=None, col_map=None, n=None, tiled=False):
    """dW[m, col(k)] += sum_s dz[s, m] a[s, k] / gfn_grad_scale(gmax) from split-f16 planes.  dz and a are
    (hi, lo') pairs of [n, width] tensors, or (hi, lo', col0, width) to address a column range of wider planes; with
    tiled=True they are (hi_flat, lo_flat, width): whole tile-major planes as the fused kernels write them (plane_flat), where
    lo_flat may be None: one plane, its values rounded to nearest (csrc/x3_device.h, PL_RNE)."""
    def unpack(p):
        hi, lo = p[0], p[1]
        _chk(hi, torch.float16, "plane"); _chk(lo, torch.float16, "plane")
        if tiled:
            width = p[2]
            if (lo is not None and hi.numel() != lo.numel()) or hi.numel() % (32 * width):
                raise _lib.GfnError("tile-major plane halves differ or do not hold a whole number of 32-sample tiles")
            return hi.data_ptr(), (None if lo is None else lo.data_ptr()), width, width, hi.numel() // width
        if lo is None:
            raise _lib.GfnError("one-plane operands (lo None) need the tile-major form")
        col0, width = (p[2], p[3]) if len(p) == 4 else (0, hi.shape[1])
        if hi.shape != lo.shape or col0 + width > hi.shape[1]:
            raise _lib.GfnError("plane halves differ or the column range is out of bounds")
        eb = hi.element_size()
        return hi.data_ptr() + col0 * eb, lo.data_ptr() + col0 * eb, hi.stride(0), width, hi.shape[0]
    dzh, dzl, ld_dz, w_dz, n_dz = unpack(dz)
    ah, al, ld_a, w_a, n_a = unpack(a)
    _chk(dW, torch.float32, "dW"); _chk(db, torch.float32, "db"); _chk(gmax, torch.float32, "gmax"); _chk(col_map, torch.int32, "col_map")
    n = min(n_dz, n_a) if n is None else n
    m_valid = w_dz if m_valid is None else m_valid
    k_valid = w_a if k_valid is None else k_valid
    if dW.shape[0] < m_valid or (col_map is None and dW.shape[1] < k_valid) or (col_map is not None and col_map.numel() < k_valid):
        raise _lib.GfnError("dW / col_map too small for the requested rows and columns")
    rc = _lib.lib().gfn_wgrad_x3(dzh, dzl, ld_dz, w_dz, ah, al, ld_a, w_a, n, _ptr(n_dev), _ptr(gmax), m_valid, k_valid, _ptr(col_map),
                                 _ptr(dW), dW.stride(0), _ptr(db), int(bool(tiled)), _stream())
    _lib.check(rc, "gfn_wgrad_x3")
    return dW


def wgrad_parts(grad_len, device):
    """Zeroed workspace for wgrad_x3_layers: one partial copy of the flat gradient vector per workgroup."""
    stride = (int(grad_len) + 63) // 64 * 64
    return torch.zeros((_lib.lib().gfn_wgrad_x3_parts(), stride), dtype=torch.float32, device=device)


def wgrad_x3_layers(layers, grad, n, n_dev=None, gmax=None, parts=None):
    """Weight (and bias) gradients of several decoder layers from tile-major split-f16 planes in ONE launch, added to `grad`.
    layers: [(dz, a, dW, db, m_valid, col_map)] with dz / a = (hi_flat, lo_flat, width) (plane_flat; lo_flat None = one plane,
    rounded to nearest: per layer and operand) and dW / db views INTO the flat f32 vector grad.  parts (wgrad_parts): the workgroups store partial sums there and a second small kernel adds them up
    (no atomics on the 31 k weights from 512 workgroups); None: f32 atomics straight into grad."""
    _chk(grad, torch.float32, "grad"); _chk(gmax, torch.float32, "gmax"); _chk(parts, torch.float32, "parts")
    arr = (_lib.WgradLayer * len(layers))()
    base, glen = grad.data_ptr(), grad.numel()
    for i, (dz, a, dW, db, m_valid, col_map) in enumerate(layers):
        for pl in (dz, a):
            _chk(pl[0], torch.float16, "plane"); _chk(pl[1], torch.float16, "plane")
            if (pl[1] is not None and pl[0].numel() != pl[1].numel()) or pl[0].numel() % (32 * pl[2]) or pl[0].numel() // pl[2] < n:
                raise _lib.GfnError("wgrad_x3_layers: plane halves differ, are not whole 32-sample tiles, or hold fewer than n samples")
        _chk(dW, torch.float32, "dW"); _chk(db, torch.float32, "db"); _chk(col_map, torch.int32, "col_map")
        for t in (dW, db):
            if t is not None and not (base <= t.data_ptr() and t.data_ptr() + t.numel() * 4 <= base + glen * 4):
                raise _lib.GfnError("wgrad_x3_layers: dW / db must be views into grad")
        k_valid = a[2]
        if col_map is not None:
            k_valid = min(k_valid, col_map.numel())
        if dW.shape[0] < m_valid or (col_map is None and dW.shape[1] < k_valid):
            raise _lib.GfnError("wgrad_x3_layers: dW too small for the requested rows and columns")
        arr[i] = _lib.WgradLayer(dz[0].data_ptr(), _ptr(dz[1]), a[0].data_ptr(), _ptr(a[1]), dz[2], a[2], int(m_valid), int(k_valid),
                                 _ptr(col_map), dW.data_ptr(), dW.stride(0), _ptr(db))
    if parts is not None and (parts.shape[0] != _lib.lib().gfn_wgrad_x3_parts() or parts.shape[1] < glen):
        raise _lib.GfnError("wgrad_x3_layers: parts must come from wgrad_parts(grad.numel(), ...)")
    stride = parts.shape[1] if parts is not None else 0
    rc = _lib.lib().gfn_wgrad_x3_layers(arr, len(layers), int(n), _ptr(n_dev), _ptr(gmax), base, glen, _ptr(parts), stride, _stream())
    _lib.check(rc, "gfn_wgrad_x3_layers")
    if parts is not None:
        n_parts = min((int(n) + 31) // 32, parts.shape[0])  # the launch's grid: workgroups beyond it wrote nothing this time
        _lib.check(_lib.lib().gfn_wgrad_reduce(_ptr(parts), n_parts, stride, glen, base, _stream()), "gfn_wgrad_reduce")


def grad_scale(gmax: float) -> float:
    """The power-of-two scale the global-stage decoder backward derives from the largest incoming gradient."""
    return float(_lib.lib().gfn_grad_scale_host(float(gmax)))


def grad_max(d_sigma, d_rgb, h0, gmax, n_dev=None):
    """gmax[0] = max(gmax[0], max |d rgb|, max |d sigma * exp(clamp(h0 + 1))|) (gmax zeroed by the caller)."""
    _chk(d_sigma, torch.float32, "d_sigma"); _chk(d_rgb, torch.float32, "d_rgb", 3); _chk(h0, torch.float32, "h0")
    _chk(gmax, torch.float32, "gmax")
    rc = _lib.lib().gfn_grad_max(_ptr(d_sigma), _ptr(d_rgb), _ptr(h0), d_sigma.shape[0], _ptr(n_dev), _ptr(gmax), _stream())
    _lib.check(rc, "gfn_grad_max")
    return gmax


def field_bwd_x3(d_sigma, d_rgb, rgb, h0, masks, image, gmax, planes, ray_id=None, cam_idx=None, demb=None, n_dev=None, opts=None):
    """Fused decoder backward of the global stage: dfeat [n,32] + dz planes (for wgrad_x3) + appearance-embedding gradient."""
    _chk(d_sigma, torch.float32, "d_sigma"); _chk(d_rgb, torch.float32, "d_rgb", 3); _chk(rgb, torch.float32, "rgb", 3)
    _chk(h0, torch.float32, "h0"); _chk(masks, torch.int64, "masks", 6); _chk(gmax, torch.float32, "gmax")
    _chk(ray_id, torch.int32, "ray_id"); _chk(cam_idx, torch.int32, "cam_idx"); _chk(demb, torch.float32, "demb", 32)
    ph, pl = planes
    n = d_sigma.shape[0]
    cap = _plane_cap(ph, pl, FIELD_X3_BWD_PLANE_LD if (pl is not None or ph is None) else FIELD_X3_BWD_PLANE_LD_RNE, n)
    dfeat = torch.empty((n, 32), dtype=torch.float32, device=d_sigma.device)
    rc = _lib.lib().gfn_field_bwd_x3(_ptr(d_sigma), _ptr(d_rgb), _ptr(rgb), _ptr(h0), _ptr(masks), _ptr(image), n, _ptr(n_dev),
                                     _ptr(dfeat), _ptr(gmax), _ptr(ph), _ptr(pl), cap, _ptr(ray_id), _ptr(cam_idx), _ptr(demb), _opt(opts), _stream())
    _lib.check(rc, "gfn_field_bwd_x3")
    return dfeat


def fused_forward(s, table_f16, prim_pool, n_volumes, log2_T, image, sh, emb, cam_idx, table2_f16=None, prim_pool2=None, bias_pool=None,
                  early_T=0.0, save=False, want_weights=True, want_feat=False, planes=None, opts=None):
    """gfn_fused_forward on the packed sampler outputs `s` (dict of ops.sample_rays with table_space=True).
    -> dict(sigma, rgb_samples, h0, masks, weights, alphas, feat, rgb [R,3], acc [R], depth_raw [R], t_minmax [2])."""
    po = s["pts_offsets"]
    R = po.numel() - 1
    n = s["ts"].shape[0]
    dev = po.device
    for nm, t, dt_, last in (("pts_offsets", po, torch.int32, None), ("warped_pts", s["warped_pts"], torch.float32, 3),
                             ("anchors", s["anchors"], torch.int32, 3), ("dists", s["dists"], torch.float32, None), ("ts", s["ts"], torch.float32, None),
                             ("table", table_f16, torch.float16, None), ("table2", table2_f16, torch.float16, None),
                             ("prim_pool", prim_pool, torch.int32, None), ("prim_pool2", prim_pool2, torch.int32, None),
                             ("bias_pool", bias_pool, torch.float32, None), ("image", image, torch.uint8, None), ("sh", sh, torch.float32, 16),
                             ("emb", emb, torch.float32, 32), ("cam_idx", cam_idx, torch.int32, None)):
        _chk(t, dt_, nm, last)
    for t in (table_f16, table2_f16):
        if t is not None and t.numel() != 16 * (1 << log2_T) * 2:
            raise _lib.GfnError("f16 table has the wrong size")
    if prim_pool.numel() != 16 * n_volumes * 3 or (prim_pool2 is not None and prim_pool2.numel() != prim_pool.numel()):
        raise _lib.GfnError("prim_pool does not match n_volumes")
    if image.numel() != FIELD_X3_FWD_BYTES or sh.shape[0] != R or (cam_idx is not None and cam_idx.numel() != R):
        raise _lib.GfnError("weight image / sh / cam_idx do not match the batch")
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    out = dict(sigma=f32(n), rgb_samples=f32(n, 3), h0=f32(n) if save else None,
               masks=torch.empty((n, 6), dtype=torch.int64, device=dev) if save else None,
               weights=f32(n) if want_weights else None, alphas=f32(n) if want_weights else None, feat=f32(n, 32) if want_feat else None,
               rgb=f32(R, 3), acc=f32(R), depth_raw=f32(R), t_minmax=f32(2))
    ph, pl = planes if planes is not None else (None, None)
    cap = _plane_cap(ph, pl, FIELD_X3_FWD_PLANE_LD, n)
    counter = torch.empty(8, dtype=torch.int32, device=dev)
    rc = _lib.lib().gfn_fused_forward(_ptr(po), R, _ptr(s["warped_pts"]), _ptr(s["anchors"]), _ptr(s["dists"]), _ptr(s["ts"]), _ptr(table_f16),
                                      _ptr(table2_f16), _ptr(prim_pool), _ptr(prim_pool2), _ptr(bias_pool), n_volumes, log2_T,
                                      _ptr(image),
                                      _ptr(sh), _ptr(emb), _ptr(cam_idx), float(early_T), _ptr(out["sigma"]), _ptr(out["rgb_samples"]),
                                      _ptr(out["h0"]), _ptr(out["masks"]), _ptr(out["weights"]), _ptr(out["alphas"]), _ptr(out["feat"]),
                                      _ptr(ph), _ptr(pl), cap, _ptr(out["rgb"]), _ptr(out["acc"]), _ptr(out["depth_raw"]), _ptr(out["t_minmax"]),
                                      _ptr(counter), _opt(opts), _stream())
    _lib.check(rc, "gfn_fused_forward")
    return out


def fused_render(s, table_f16, prim_pool, n_volumes, log2_T, image, sh, emb, cam_idx, table2_f16=None, prim_pool2=None, bias_pool=None,
                 early_T=0.0, tile_width=0, patch_width=8, opts=None, cam_uniform=False):
    """gfn_fused_render: inference through patches of 32 neighbouring rays walked in lockstep (no per-sample outputs).
    -> dict(rgb [R,3], acc [R], depth_raw [R], t_minmax [2]).  R must be a multiple of 32; tile_width: the rays are a block of
    pixels this wide, row major (0: a run of rays).  s: the packed samples of sample_rays, or the records of sample_rays_record
    (gfn_fused_render_rec: same images, no fill pass in between).  cam_uniform: every ray carries the camera cam_idx[0] (one image): the
    appearance embedding's share of the head's first layer is folded into its bias once per workgroup."""
    rec = s.get("pts_offsets") is None  # the record form: samples of sample_rays_record (counts + march_rec)
    if rec:
        po = None
        R = s["counts"].numel()
        dev = s["counts"].device
        _chk(s["counts"], torch.int32, "counts"); _chk(s["march_rec"], torch.float32, "march_rec"); _chk(s.get("page_table"), torch.int32, "page_table")
        if s.get("page_table") is not None:
            if s["page_table"].numel() < R * (MAX_SAMPLE_PER_RAY // REC_PAGE) or s["march_rec"].numel() < s["n_pages"] * REC_PAGE * MARCH_REC_FLOATS:
                raise _lib.GfnError("page_table / march_rec do not match the paged record form")
        elif s["march_rec"].numel() < R * MAX_SAMPLE_PER_RAY * MARCH_REC_FLOATS:
            raise _lib.GfnError("march_rec does not hold 1024 record slots per ray")
    else:
        po = s["pts_offsets"]
        R = po.numel() - 1
        dev = po.device
        for nm, t, dt_, last in (("pts_offsets", po, torch.int32, None), ("warped_pts", s["warped_pts"], torch.float32, 3),
                                 ("anchors", s["anchors"], torch.int32, 3), ("dists", s["dists"], torch.float32, None), ("ts", s["ts"], torch.float32, None)):
            _chk(t, dt_, nm, last)
    for nm, t, dt_, last in (("table", table_f16, torch.float16, None), ("table2", table2_f16, torch.float16, None),
                             ("prim_pool", prim_pool, torch.int32, None), ("prim_pool2", prim_pool2, torch.int32, None),
                             ("bias_pool", bias_pool, torch.float32, None), ("image", image, torch.uint8, None), ("sh", sh, torch.float32, 16),
                             ("emb", emb, torch.float32, 32), ("cam_idx", cam_idx, torch.int32, None)):
        _chk(t, dt_, nm, last)
    for t in (table_f16, table2_f16):
        if t is not None and t.numel() != 16 * (1 << log2_T) * 2:
            raise _lib.GfnError("f16 table has the wrong size")
    if prim_pool.numel() != 16 * n_volumes * 3 or (prim_pool2 is not None and prim_pool2.numel() != prim_pool.numel()):
        raise _lib.GfnError("prim_pool does not match n_volumes")
    if image.numel() != FIELD_X3_FWD_BYTES or sh.shape[0] != R or (cam_idx is not None and cam_idx.numel() != R):
        raise _lib.GfnError("weight image / sh / cam_idx do not match the batch")
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    out = dict(rgb=f32(R, 3), acc=f32(R), depth_raw=f32(R), t_minmax=f32(2))
    counter = torch.empty(8, dtype=torch.int32, device=dev)
    tail = (_ptr(table_f16), _ptr(table2_f16), _ptr(prim_pool), _ptr(prim_pool2), _ptr(bias_pool), n_volumes, log2_T,
            _ptr(image), _ptr(sh), _ptr(emb), _ptr(cam_idx), int(bool(cam_uniform)), float(early_T), int(tile_width), int(patch_width),
            _ptr(out["rgb"]), _ptr(out["acc"]), _ptr(out["depth_raw"]), _ptr(out["t_minmax"]), _ptr(counter), _opt(opts), _stream())
    if rec:
        rc = _lib.lib().gfn_fused_render_rec(_ptr(s["counts"]), R, _ptr(s["march_rec"]), _ptr(s.get("page_table")), *tail)
    else:
        rc = _lib.lib().gfn_fused_render(_ptr(po), R, _ptr(s["warped_pts"]), _ptr(s["anchors"]), _ptr(s["dists"]), _ptr(s["ts"]), *tail)
    _lib.check(rc, "gfn_fused_render")
    return out


def field_bwd_scatter_x3(d_sigma, d_rgb, rgb, h0, masks, image, pts, anchors, prim_pool, bias_pool, n_volumes, log2_T, grad_pool,
                         gmax=None, planes=None, ray_id=None, cam_idx=None, demb=None, n_dev=None, records=None, sample_list=None, opts=None):
    """Decoder backward + hash scatter in one kernel: accumulates into grad_pool (f16 x128-scaled or f32, like hash_bwd); with
    planes / gmax (global stage) it also writes the dz planes for wgrad_x3 and the appearance-embedding gradient."""
    _chk(d_sigma, torch.float32, "d_sigma"); _chk(d_rgb, torch.float32, "d_rgb", 3); _chk(rgb, torch.float32, "rgb", 3)
    _chk(h0, torch.float32, "h0"); _chk(masks, torch.int64, "masks", 6); _chk(image, torch.uint8, "image")
    _chk(pts, torch.float32, "pts", 3); _chk(anchors, torch.int32, "anchors", 3); _chk(prim_pool, torch.int32, "prim_pool")
    _chk(bias_pool, torch.float32, "bias_pool"); _chk(gmax, torch.float32, "gmax")
    _chk(ray_id, torch.int32, "ray_id"); _chk(cam_idx, torch.int32, "cam_idx"); _chk(demb, torch.float32, "demb", 32)
    if grad_pool.dtype not in (torch.float16, torch.float32) or grad_pool.numel() != 16 * (1 << log2_T) * 2:
        raise _lib.GfnError("grad_pool must be an f16 or f32 tensor of 16*T*2 elements")
    if image.numel() != FIELD_X3_BWD_BYTES or prim_pool.numel() != 16 * n_volumes * 3:
        raise _lib.GfnError("weight image or prim_pool of the wrong size")
    n = d_sigma.shape[0]
    ph, pl = planes if planes is not None else (None, None)
    cap = _plane_cap(ph, pl, FIELD_X3_BWD_PLANE_LD if (pl is not None or ph is None) else FIELD_X3_BWD_PLANE_LD_RNE, n)
    rec, cnt = (records[0], records[1]) if records is not None else (None, None)
    lst, lst_n = sample_list if sample_list is not None else (None, None)  # (no planes) walk only these samples: the others' gradients are zero
    _chk(lst, torch.int32, "sample_list"); _chk(lst_n, torch.int32, "sample_list count")
    if lst is not None and planes is not None:
        raise _lib.GfnError("field_bwd_scatter_x3: a sample list excludes the gradient planes (their unlisted rows would keep stale values)")
    if rec is not None:
        _chk(rec, torch.int64, "records"); _chk(cnt, torch.int32, "counts")
        if cnt.numel() != _lib.lib().gfn_grad_record_ranges(int(log2_T)) * 256 or grad_pool.dtype != torch.float16:
            raise _lib.GfnError("field_bwd_scatter_x3: records need counts int32 [ranges * 256] and the f16 gradient pool")
    rc = _lib.lib().gfn_field_bwd_scatter_x3(_ptr(d_sigma), _ptr(d_rgb), _ptr(rgb), _ptr(h0), _ptr(masks), _ptr(image), n, _ptr(n_dev), _ptr(pts),
                                             _ptr(anchors), _ptr(prim_pool), _ptr(bias_pool), n_volumes, log2_T, _ptr(grad_pool),
                                             int(grad_pool.dtype == torch.float16), _ptr(gmax), _ptr(ph), _ptr(pl), cap, _ptr(ray_id),
                                             _ptr(cam_idx), _ptr(demb), _ptr(rec), _ptr(cnt), rec.numel() if rec is not None else 0,
                                             _ptr(lst), _ptr(lst_n), _opt(opts), _stream())
    _lib.check(rc, "gfn_field_bwd_scatter_x3")
    return grad_pool


GRAD_RANGE_LOG2 = 14  # GFN_GRAD_RANGE_LOG2: f16 pairs per slot range of the record streams


def grad_record_ranges(log2_T):
    """Slot ranges of a table's record streams (gfn_grad_record_ranges)."""
    return int(_lib.lib().gfn_grad_record_ranges(int(log2_T)))


def grad_records_buffers(log2_T, capacity, device):
    """(records int64 [capacity], counters int32 [ranges * 256]) for the deferred-atomics backward: one record stream per
    (16 Ki-slot range, producing workgroup)."""
    n_ranges = _lib.lib().gfn_grad_record_ranges(int(log2_T))
    if capacity < n_ranges * 256 * 2:
        raise _lib.GfnError(f"grad_records_buffers: capacity must be at least {n_ranges * 512} records")
    # the counters start at zero and every gfn_grad_records_reduce leaves them zero again
    return (torch.empty(int(capacity), dtype=torch.int64, device=device), torch.zeros(n_ranges * 256, dtype=torch.int32, device=device))


def grad_records_reduce(records, grad_pool, log2_T, counts_seen=None, f16_accumulate=False, ranges=None):
    """Accumulate the (slot, value) records of field_bwd_scatter_x3(records=...) into grad_pool, one slot range per workgroup, in
    LDS: exact fixed-point sums rounded to f16 once wherever they provably fit, the reference's f16 accumulation elsewhere (or
    everywhere with f16_accumulate).  Consumes the stream counters (zero again afterwards); counts_seen (int32, same shape)
    receives a copy.  ranges: (begin, end) -- only those slot ranges (range r = pool elements [r << 15, (r + 1) << 15))."""
    rec, cnt = records
    r0, r1 = (0, 0) if ranges is None else ranges
    _chk(rec, torch.int64, "records"); _chk(cnt, torch.int32, "counts"); _chk(grad_pool, torch.float16, "grad_pool")
    _chk(counts_seen, torch.int32, "counts_seen")
    if cnt.numel() != _lib.lib().gfn_grad_record_ranges(int(log2_T)) * 256:
        raise _lib.GfnError("grad_records_reduce: counters must come from grad_records_buffers(log2_T, ...)")
    if counts_seen is not None and counts_seen.numel() != cnt.numel():
        raise _lib.GfnError("grad_records_reduce: counts_seen must have the size of the counters")
    rc = _lib.lib().gfn_grad_records_reduce(_ptr(rec), _ptr(cnt), rec.numel(), int(log2_T), _ptr(grad_pool), _ptr(counts_seen),
                                            int(bool(f16_accumulate)), int(r0), int(r1), _stream())
    _lib.check(rc, "gfn_grad_records_reduce")


def sh4(dirs, out=None):
    _chk(dirs, torch.float32, "dirs", 3); _chk(out, torch.float32, "out", 16)
    out = torch.empty((dirs.shape[0], 16), dtype=torch.float32, device=dirs.device) if out is None else out
    _lib.check(_lib.lib().gfn_sh4(_ptr(dirs), dirs.shape[0], _ptr(out), _stream()), "gfn_sh4")
    return out


def head_pack(h, sh, emb, cam_idx, ray_id, n_dev=None):
    _chk(h, torch.float32, "h", 16); _chk(sh, torch.float32, "sh", 16); _chk(emb, torch.float32, "emb", 32)
    _chk(cam_idx, torch.int32, "cam_idx"); _chk(ray_id, torch.int32, "ray_id")
    n = h.shape[0]
    sigma = torch.empty(n, dtype=torch.float32, device=h.device)
    xh = torch.empty((n, 64), dtype=torch.float32, device=h.device)
    rc = _lib.lib().gfn_head_pack(_ptr(h), _ptr(sh), _ptr(emb), _ptr(cam_idx), _ptr(ray_id), n, _ptr(n_dev), _ptr(sigma),
                                  _ptr(xh), _stream())
    _lib.check(rc, "gfn_head_pack")
    return sigma, xh


def head_unpack_bwd(h, dsigma, dxh, cam_idx, pts_offsets, demb=None, n_dev=None):
    _chk(h, torch.float32, "h", 16); _chk(dsigma, torch.float32, "dsigma"); _chk(dxh, torch.float32, "dxh", 64)
    _chk(cam_idx, torch.int32, "cam_idx"); _chk(pts_offsets, torch.int32, "pts_offsets"); _chk(demb, torch.float32, "demb", 32)
    n = h.shape[0]
    dh = torch.empty((n, 16), dtype=torch.float32, device=h.device)
    rc = _lib.lib().gfn_head_unpack_bwd(_ptr(h), _ptr(dsigma), _ptr(dxh), _ptr(cam_idx), _ptr(pts_offsets),
                                        pts_offsets.numel() - 1, n, _ptr(n_dev), _ptr(dh), _ptr(demb), _stream())
    _lib.check(rc, "gfn_head_unpack_bwd")
    return dh


# --------------------------------------------------------------------- compositing
def composite_fwd(pts_offsets, sigma, dt, t, rgb, early_T=0.0, want_weights=True, want_t_range=True):
    _chk(pts_offsets, torch.int32, "pts_offsets"); _chk(sigma, torch.float32, "sigma"); _chk(dt, torch.float32, "dt")
    _chk(t, torch.float32, "t"); _chk(rgb, torch.float32, "rgb", 3)
    R = pts_offsets.numel() - 1
    dev = sigma.device
    n = sigma.shape[0]
    weights = torch.empty(n, dtype=torch.float32, device=dev) if want_weights else None
    alphas = torch.empty(n, dtype=torch.float32, device=dev) if want_weights else None
    out_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    out_acc = torch.empty(R, dtype=torch.float32, device=dev)
    out_depth = torch.empty(R, dtype=torch.float32, device=dev)
    t_minmax = torch.empty(2, dtype=torch.float32, device=dev) if want_t_range else None  # (min t, max t) over the batch: the depth clip
    rc = _lib.lib().gfn_composite_fwd(_ptr(pts_offsets), R, _ptr(sigma), _ptr(dt), _ptr(t), _ptr(rgb), float(early_T),
                                      _ptr(weights), _ptr(alphas), _ptr(out_rgb), _ptr(out_acc), _ptr(out_depth),
                                      _ptr(t_minmax), _stream())
    _lib.check(rc, "gfn_composite_fwd")
    return dict(weights=weights, alphas=alphas, rgb=out_rgb, acc=out_acc, depth_raw=out_depth, t_minmax=t_minmax)


class PhaseState:
    """Device-side state of the phased (termination-aware) training forward for batches of n_rays rays / cap sample slots."""

    def __init__(self, n_rays, cap, device):
        i32 = lambda n, v=0: torch.full((n,), v, dtype=torch.int32, device=device)  # noqa: E731
        self.n_rays, self.cap = int(n_rays), int(cap)
        self.k = i32(1, 1024)        # K of the next forward (1024 = one phase); gfn_two_phase_plan rewrites it every step
        self.stats = i32(8)          # [samples, rays of phase A, of phase B, in front of the termination points] of the step in flight
        self.stats_last = i32(4)     # [samples of phase A, rays of phase B, samples of phase B, K] of the last finished step
        self.list_a, self.list_b, self.list_front = i32(cap), i32(cap), i32(cap)
        self.coff = i32(n_rays + 1)
        self.need_more = torch.zeros(n_rays, dtype=torch.uint8, device=device)
        self.term = i32(n_rays)

    def samples(self, phase_b=False, front=False):
        """(list, count) of a phase for hash_fwd_lm / field_fwd, or of the samples in front of the termination points for the backward."""
        if front:
            return (self.list_front, self.stats[4:5])
        return (self.list_b, self.stats[2:3]) if phase_b else (self.list_a, self.stats[0:1])


def phase_list(pts_offsets, st, phase_b=False, zero=None, front=False):
    """Build the sample list of phase A (the first K samples of every ray), of phase B (samples K.. of the rays with need_more), or
    (front) of the samples in front of the termination points (the first st.term[r] of every ray: what the backward has to walk).
    zero = (sigma [n], rgb [n,3]) (phase A): the samples that are NOT listed get sigma = rgb = 0."""
    _chk(pts_offsets, torch.int32, "pts_offsets")
    zs, zr = zero if zero is not None else (None, None)
    _chk(zs, torch.float32, "sigma"); _chk(zr, torch.float32, "rgb", 3)
    if st.n_rays != pts_offsets.numel() - 1:
        raise _lib.GfnError("phase_list: the PhaseState was made for another batch size")
    rc = _lib.lib().gfn_phase_list(_ptr(pts_offsets), st.n_rays, _ptr(st.k), _ptr(st.need_more) if phase_b else None, _ptr(st.term) if front else None,
                                   _ptr(st.coff), _ptr(st.list_front if front else (st.list_b if phase_b else st.list_a)), _ptr(st.stats), _ptr(zs), _ptr(zr),
                                   _stream())
    _lib.check(rc, "gfn_phase_list")
    return st.samples(phase_b, front)


def composite_fwd_phased(pts_offsets, sigma, dt, t, rgb, st, early_T, second=False, want_t_range=True, out=None):
    """composite_fwd inside the phased forward.  First pass (second=False): every ray, + which rays still need their samples behind K
    (st.need_more) and where every ray ends (st.term).  Second pass: only the rays with need_more, into the tensors of `out` (the
    first pass's result); nothing when phase B is empty."""
    _chk(pts_offsets, torch.int32, "pts_offsets"); _chk(sigma, torch.float32, "sigma"); _chk(dt, torch.float32, "dt")
    _chk(t, torch.float32, "t"); _chk(rgb, torch.float32, "rgb", 3)
    R = pts_offsets.numel() - 1
    dev = sigma.device
    n = sigma.shape[0]
    if st.n_rays != R or st.cap < n:
        raise _lib.GfnError("composite_fwd_phased: the PhaseState was made for another batch size")
    if second:
        if out is None:
            raise _lib.GfnError("composite_fwd_phased: the second pass writes into the first pass's outputs")
        o = out
        t_minmax = None
    else:
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        o = dict(weights=f32(n), alphas=f32(n), rgb=f32(R, 3), acc=f32(R), depth_raw=f32(R), t_minmax=f32(2) if want_t_range else None)
        t_minmax = o["t_minmax"]
    rc = _lib.lib().gfn_composite_fwd_phased(_ptr(pts_offsets), R, _ptr(sigma), _ptr(dt), _ptr(t), _ptr(rgb), float(early_T), _ptr(o["weights"]),
                                             _ptr(o["alphas"]), _ptr(o["rgb"]), _ptr(o["acc"]), _ptr(o["depth_raw"]), _ptr(t_minmax),
                                             _ptr(st.need_more) if second else None, _ptr(st.stats[2:3]) if second else None,
                                             None if second else _ptr(st.k), None if second else _ptr(st.need_more), _ptr(st.term), _stream())
    _lib.check(rc, "gfn_composite_fwd_phased")
    return o


def two_phase_plan(pts_offsets, st, fixed_cost_samples, save_enter=0.12, save_stay=0.06, enabled=True):
    _lib.check(_lib.lib().gfn_two_phase_plan(_ptr(pts_offsets), pts_offsets.numel() - 1, _ptr(st.term), _ptr(st.k), _ptr(st.stats), _ptr(st.stats_last),
                                            float(fixed_cost_samples), float(save_enter), float(save_stay), int(bool(enabled)), _stream()), "gfn_two_phase_plan")


def depth_finalize(depth_raw, t_minmax, scale_factor):
    d = depth_raw.clone()
    _lib.check(_lib.lib().gfn_depth_finalize(_ptr(d), d.numel(), _ptr(t_minmax), float(scale_factor), _stream()),
               "gfn_depth_finalize")
    return d


def composite_bwd(pts_offsets, sigma, dt, rgb, g_rgb, g_acc=None, early_T=0.0):
    _chk(g_rgb, torch.float32, "g_rgb", 3); _chk(g_acc, torch.float32, "g_acc")
    R = pts_offsets.numel() - 1
    n = sigma.shape[0]
    d_sigma = torch.empty(n, dtype=torch.float32, device=sigma.device)
    d_rgb = torch.empty((n, 3), dtype=torch.float32, device=sigma.device)
    rc = _lib.lib().gfn_composite_bwd(_ptr(pts_offsets), R, _ptr(sigma), _ptr(dt), _ptr(rgb), _ptr(g_rgb), _ptr(g_acc),
                                      float(early_T), _ptr(d_sigma), _ptr(d_rgb), _stream())
    _lib.check(rc, "gfn_composite_bwd")
    return d_sigma, d_rgb


# ------------------------------------------------------------------ loss / optimiser
def charbonnier(pred, gt, eps=1e-6, loss=None):
    """loss[0] += Charbonnier(pred, gt) (losses.py:73-84); loss: a zeroed f32 [1] to add to (allocated when absent)."""
    _chk(pred, torch.float32, "pred", 3); _chk(gt, torch.float32, "gt", 3); _chk(loss, torch.float32, "loss")
    if loss is None:
        loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
    g = torch.empty_like(pred)
    _lib.check(_lib.lib().gfn_charbonnier(_ptr(pred), _ptr(gt), pred.shape[0], float(eps), _ptr(loss), _ptr(g), _stream()),
               "gfn_charbonnier")
    return loss, g


TRAIN_LOSSES_WS = 132  # GFN_TRAIN_LOSSES_WS


def train_losses_workspace(device):
    """The (zeroed) workspace that lets train_losses spread over up to 64 workgroups: one per caller (an engine's own), never shared between streams."""
    return torch.zeros(TRAIN_LOSSES_WS, dtype=torch.float32, device=device)


def train_losses(pred, gt, eps=1e-6, want_s3im_grad=True, errmap=None, ws=None):
    """gfn_train_losses: what a training iteration of the plugin surface runs between the compositing and the backward, in ONE launch ->
    (out3 f32 [3] = [Charbonnier loss, 0 (for s3im to add into), PSNR], g_char [R,3], g_zero [R,3] zeroed | None).
    errmap = (error_map f32 [n_images,H,W(,1)], block_sums f64 | None, indices i32 [R,3] = (c, y, x), slot i32 [R] | None): the block stage's
    error-map write-back (errmap_update) rides along.  ws: train_losses_workspace() of the caller (several workgroups; None: one)."""
    _chk(pred, torch.float32, "pred", 3); _chk(gt, torch.float32, "gt", 3); _chk(ws, torch.float32, "ws")
    if pred.shape != gt.shape or (ws is not None and ws.numel() < TRAIN_LOSSES_WS):
        raise _lib.GfnError("train_losses: pred and gt differ in shape, or the workspace is too small")
    out3 = torch.empty(3, dtype=torch.float32, device=pred.device)
    g = torch.empty_like(pred)
    gz = torch.empty_like(pred) if want_s3im_grad else None
    em, bs, idx, slot, n_img, H, W = None, None, None, None, 0, 0, 0
    if errmap is not None:
        em, bs, idx, slot = errmap
        _chk(em, torch.float32, "error_map"); _chk(bs, torch.float64, "block_sums"); _chk(idx, torch.int32, "indices", 3); _chk(slot, torch.int32, "slot")
        if idx.shape[0] != pred.shape[0] or em.dim() < 3 or (slot is not None and slot.numel() != pred.shape[0]):
            raise _lib.GfnError("train_losses: one (c, y, x) row (and slot) per ray and an error map [n_images,H,W]")
        if bs is not None and bs.numel() != (em.numel() + ERRMAP_BLOCK - 1) // ERRMAP_BLOCK:
            raise _lib.GfnError("train_losses: block_sums does not belong to this error map")
        n_img, H, W = em.shape[:3]
    rc = _lib.lib().gfn_train_losses(_ptr(pred), _ptr(gt), pred.shape[0], float(eps), _ptr(out3), _ptr(g), _ptr(gz), _ptr(idx), _ptr(slot),
                                     int(n_img), int(H), int(W), _ptr(em), _ptr(bs), _ptr(ws), _stream())
    _lib.check(rc, "gfn_train_losses")
    return out3, g, gz


def loss_grad_combine(g_a, w_a, g_b=None, w_b=None, out=None):
    """out = g_a * w_a[0] + g_b * w_b[0] (w_*: device f32 scalars, e.g. the gradients autograd hands the backward of two losses)."""
    _chk(g_a, torch.float32, "g_a"); _chk(g_b, torch.float32, "g_b")
    for w in (w_a, w_b):
        if w is not None and not (w.is_cuda and w.dtype == torch.float32 and w.numel() == 1):
            raise _lib.GfnError("loss_grad_combine: weights are device float32 scalars")
    if g_b is not None and g_b.numel() != g_a.numel():
        raise _lib.GfnError("loss_grad_combine: gradients differ in size")
    out = torch.empty_like(g_a) if out is None else out
    _chk(out, torch.float32, "out")
    _lib.check(_lib.lib().gfn_loss_grad_combine(_ptr(g_a), _ptr(w_a), _ptr(g_b), _ptr(w_b), g_a.numel(), _ptr(out), _stream()), "gfn_loss_grad_combine")
    return out


def s3im_index(n_rays, repeat_time=10, generator=None, device=None, out=None):
    """The index vector S3IM.forward builds (losses.py:780-789): arange followed by repeat_time - 1 random permutations.
    The permutations come from ONE batched argsort of f64 uniforms (three launches instead of 9 x torch.randperm: this runs
    every training step)."""
    out = torch.empty((repeat_time, n_rays), dtype=torch.int32, device=device) if out is None else out.view(repeat_time, n_rays)
    out[0] = torch.arange(n_rays, device=device, dtype=torch.int32)
    if repeat_time > 1:
        keys = torch.rand((repeat_time - 1, n_rays), dtype=torch.float64, device=device, generator=generator)
        out[1:] = torch.argsort(keys, dim=1)
    return out.view(-1)


class StepDraws:
    """The per-step random draws of a training step from the device's own Philox stream (gfn_step_draws): march noise of the NEXT
    batch and S3IM's index vector, without torch's generator (which a captured step pays for with two fills per replay)."""

    def __init__(self, n_rays, repeat_time=10, seed=None, device="cuda"):
        self.R, self.repeat = int(n_rays), int(repeat_time)
        self.seed = int(torch.initial_seed() if seed is None else seed) & ((1 << 64) - 1)
        self.state = torch.zeros(2, dtype=torch.int64, device=device)
        self.keys = torch.empty((max(self.repeat - 1, 0), self.R), dtype=torch.int64, device=device)

    def draw(self, noise_out=None, fineness=1.0, s3im_out=None, write_arange=True):
        """noise_out f32 [1024 + R + 10] <- (U[0,1) + 0.5) * fineness (sampler_noise, training mode); s3im_out i32 [repeat * R] <- arange
        followed by repeat - 1 random permutations (s3im_index)."""
        _chk(noise_out, torch.float32, "noise_out"); _chk(s3im_out, torch.int32, "s3im_out")
        nk = self.keys.numel() if s3im_out is not None else 0
        if s3im_out is not None and s3im_out.numel() != self.repeat * self.R:
            raise _lib.GfnError("StepDraws: s3im_out must hold repeat_time * n_rays indices")
        rc = _lib.lib().gfn_step_draws(self.seed, _ptr(self.state), _ptr(noise_out), 0 if noise_out is None else noise_out.numel(), 0.5, float(fineness),
                                       _ptr(self.keys) if nk else None, nk, _stream())
        _lib.check(rc, "gfn_step_draws")
        if s3im_out is not None:
            o = s3im_out.view(self.repeat, self.R)
            if write_arange:  # (a buffer s3im_index has filled once keeps its first row)
                o[0] = torch.arange(self.R, device=o.device, dtype=torch.int32)
            if nk:
                o[1:] = torch.argsort(self.keys, dim=1)
        return noise_out, s3im_out


def sampler_noise(n_rays, fineness, train=True, generator=None, device=None, out=None):
    """rays_noise of GetSamples (PersSampler_cuda.cu:380-389): (U[0,1) + 0.5) * fineness in training, fineness in eval mode,
    1024 + R + 10 values (the march reads a sliding window of it, :204,270)."""
    n = 1024 + n_rays + 10
    if out is not None:
        _chk(out, torch.float32, "out")
        if out.numel() != n:
            raise _lib.GfnError(f"sampler_noise: out must hold {n} values")
        if not train:
            return out.fill_(float(fineness))
        return torch.rand(n, dtype=torch.float32, device=out.device, generator=generator, out=out).add_(0.5).mul_(float(fineness))
    if not train:
        return torch.full((n,), float(fineness), dtype=torch.float32, device=device)
    return torch.rand(n, dtype=torch.float32, device=device, generator=generator).add_(0.5).mul_(float(fineness))


def s3im(gt, pred, index, patch_h=32, mult=1.0, loss=None, g_pred=None, want_grad=True):
    """loss[0] += mult * S3IM(gt, pred); g_pred += mult * d/d pred.  -> (loss [1], g_pred [R,3])."""
    _chk(gt, torch.float32, "gt", 3); _chk(pred, torch.float32, "pred", 3); _chk(index, torch.int32, "index")
    if gt.shape != pred.shape:
        raise _lib.GfnError("s3im: gt and pred differ in shape")
    if loss is None:
        loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
    if g_pred is None and want_grad:
        g_pred = torch.zeros_like(pred)
    _chk(loss, torch.float32, "loss"); _chk(g_pred, torch.float32, "g_pred", 3)
    rc = _lib.lib().gfn_s3im(_ptr(gt), _ptr(pred), _ptr(index), index.numel(), int(patch_h), float(mult), _ptr(loss), _ptr(g_pred), _stream())
    _lib.check(rc, "gfn_s3im")
    return loss, g_pred


def adam_hyper(step, lr=1e-2, beta1=0.9, beta2=0.999):
    """(lr / (1 - beta1^step), sqrt(1 - beta2^step)) as the Adam kernels compute them on the host: what `hyper` must hold."""
    out = np.zeros(2, np.float32)
    _lib.lib().gfn_adam_hyper_host(int(step), float(lr), float(beta1), float(beta2), out.ctypes.data_as(C.c_void_p))
    return out


def adam_step(p, g, m, v, step, lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-15, hyper=None):
    """hyper: device f32 [2] = adam_hyper(step, lr, ...) read by the kernel instead of the host's `step` / `lr` (hipGraph replay)."""
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, torch.float32, nm)
    _chk(hyper, torch.float32, "hyper")
    if not (p.numel() == g.numel() == m.numel() == v.numel()):
        raise _lib.GfnError("adam_step: size mismatch")
    _lib.check(_lib.lib().gfn_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), int(step), float(lr), float(beta1),
                                        float(beta2), float(eps), _ptr(hyper), _stream()), "gfn_adam_step")


def adam_step_table(p, g, m, v, step, lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-15, shadow_f16=None, n=None, hyper=None):
    """Adam on (the first n elements of) a hash table; g float32, or float16 carrying the x128 loss scale."""
    _chk(p, torch.float32, "p"); _chk(m, torch.float32, "m"); _chk(v, torch.float32, "v"); _chk(shadow_f16, torch.float16, "shadow")
    if g.dtype not in (torch.float32, torch.float16):
        raise _lib.GfnError("adam_step_table: gradient must be float32 or float16")
    _chk(g, g.dtype, "g")
    n = p.numel() if n is None else n
    if not (p.numel() == g.numel() == m.numel() == v.numel()) or n > p.numel():
        raise _lib.GfnError("adam_step_table: size mismatch")
    g16 = g.dtype == torch.float16
    _lib.check(_lib.lib().gfn_adam_step_ex(_ptr(p), _ptr(g), int(g16), (1.0 / 128.0) if g16 else 1.0, _ptr(m), _ptr(v), n,
                                           int(step), float(lr), float(beta1), float(beta2), float(eps), _ptr(shadow_f16),
                                           _ptr(hyper), _stream()), "gfn_adam_step_ex")


def copy_list(pairs):
    """[(dst, src)]: up to 8 small same-dtype, same-size contiguous device tensors copied in ONE launch (gfn_copy_list)."""
    import ctypes as C

    for d, s in pairs:  # (a source may also be PINNED host memory: the device reads it in place -- small per-step scalars ride along)
        if d.dtype != s.dtype or d.numel() != s.numel() or not (d.is_cuda and (s.is_cuda or s.is_pinned()) and d.is_contiguous() and s.is_contiguous()):
            raise _lib.GfnError("copy_list: pairs of contiguous tensors of one dtype and size, device destinations, device or pinned host sources")
    pairs = [(d, s) for d, s in pairs if d.numel()]  # nothing to do for empty tensors (they have no address)
    n = len(pairs)
    if n == 0:
        return
    src = (C.c_void_p * n)(); dst = (C.c_void_p * n)(); nb = (C.c_int64 * n)()
    for i, (d, s) in enumerate(pairs):
        src[i], dst[i], nb[i] = s.data_ptr(), d.data_ptr(), d.numel() * d.element_size()
    _lib.check(_lib.lib().gfn_copy_list(src, dst, nb, n, _stream()), "gfn_copy_list")


_SIDE_STREAMS = {}  # (device index, role) -> the probed stream of that role


def concurrent_stream(device=None, role=None, priority=0):
    """A stream that demonstrably runs BESIDE the current one.  role (a name: "sampler", "aux", "adam", "comm", "draws" ...): ONE stream per
    device and role for the whole process, probed once and handed out again afterwards -- torch takes its streams from a round-robin pool of 32 per
    device, so engines that each made their own would soon be handed streams already in use by another role (safe, but the overlap the probe is for
    would be lost silently): a candidate whose handle another role already holds is skipped.  role None: a fresh probed stream every call.  HIP maps streams onto a few hardware queues in creation order, so whether a new
    stream shares its queue with the stream the step is launched on -- and then runs behind it, not beside it -- depends on how many streams the
    process made before (round 4: the plugin surface's focal iteration took 1.81 or 1.55 ms depending on whether bench.py had rendered first; with
    this probe 1.55 either way).  Up to 8 candidates, each probed once: a ~2.5 ms spin on the current stream, a tiny kernel on the candidate -- does
    the candidate's event complete while the spin still runs?  ~20 ms once per stream made; a plain stream under capture (graph branches are not
    queues) or with GFN_SIDE_STREAM_PROBE=0.  priority: 0, or -1 for a HIGH-priority stream -- its workgroups are dispatched first whenever the chip has
    room for one (the next batch's sampler beside a backward that fills every CU: its scan and fill otherwise wait until the accumulation AND Adam
    behind the backward have drained, and the next forward waits for them: round 6, 0.22 ms of a 2.13 ms plugin iteration)."""
    import os
    import time

    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), role)
    if role is not None and key in _SIDE_STREAMS:
        return _SIDE_STREAMS[key]
    priority = int(os.environ.get("GFN_SIDE_STREAM_PRIORITY", priority))
    if os.environ.get("GFN_SIDE_STREAM_PROBE", "1") != "1" or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device=device, priority=priority)  # (not remembered: made without a probe)
    main = torch.cuda.current_stream(device)
    taken = {s.cuda_stream for (d, _), s in _SIDE_STREAMS.items() if d == key[0]} | {main.cuda_stream}
    probe = torch.zeros(64, dtype=torch.float32, device=device)
    kept = []  # (candidates that share the queue stay allocated until one is found: a fresh one would land on the same queue again)
    for _ in range(8):
        cand = torch.cuda.Stream(device=device, priority=priority)
        kept.append(cand)
        if cand.cuda_stream in taken:  # the pool has wrapped onto a stream in use
            continue
        torch.cuda.synchronize(device)
        done, spun = torch.cuda.Event(), torch.cuda.Event()
        torch.cuda._sleep(6_000_000)
        spun.record(main)
        with torch.cuda.stream(cand):
            probe.add_(1.0)
            done.record(cand)
        t0 = time.perf_counter()
        beside = False
        while time.perf_counter() - t0 < 2.0e-3:
            if done.query():
                beside = not spun.query()
                break
        torch.cuda.synchronize(device)
        if beside:
            if role is not None:
                _SIDE_STREAMS[key] = cand
            return cand
    free = [c for c in kept if c.cuda_stream not in taken]
    cand = free[0] if free else kept[0]
    if role is not None:
        _SIDE_STREAMS[key] = cand
    return cand


def fill_zero(t):
    """t[...] = 0 on the current stream (a kernel, not a memset node: safe inside a captured graph, see csrc/gfn_common.h gfn_zero_async)."""
    if t.numel() == 0:
        return
    if not t.is_contiguous():
        raise _lib.GfnError("fill_zero: contiguous tensors only")
    _lib.check(_lib.lib().gfn_fill_zero(_ptr(t), t.numel() * t.element_size(), _stream()), "gfn_fill_zero")


# ---------------------------------------------------------------- octree (host side)
def reduce_shards(recv, n_shards, out, scale=1.0):
    """out = scale * sum of the n_shards consecutive copies in recv (f32 accumulation); f16 or f32, see gfn_reduce_shards."""
    if recv.dtype not in (torch.float16, torch.float32):
        raise _lib.GfnError("reduce_shards: f16 or f32")
    _chk(recv, recv.dtype, "recv"); _chk(out, recv.dtype, "out")
    shard = out.numel()
    if recv.numel() != n_shards * shard:
        raise _lib.GfnError("reduce_shards: recv must hold n_shards copies of out")
    rc = _lib.lib().gfn_reduce_shards(_ptr(recv), n_shards, shard, int(recv.dtype == torch.float16), float(scale), _ptr(out), _stream())
    _lib.check(rc, "gfn_reduce_shards")


def octree_build(c2w, intri, bounds, max_depth=16, bbox_side_len=512.0, split_dist_thres=1.5, seed=1234, threads=0):
    """Host C++ builder -> (nodes uint8 [n_nodes*128], transes uint8 [n_trans*576]) numpy blobs."""
    c2w = np.ascontiguousarray(c2w, np.float32)
    intri = np.ascontiguousarray(intri, np.float32)
    bounds = np.ascontiguousarray(bounds, np.float32)
    n = c2w.shape[0]
    assert c2w.shape == (n, 3, 4) and intri.shape == (n, 3, 3) and bounds.shape == (n, 2)
    np_, tp_ = C.c_void_p(), C.c_void_p()
    nn, nt = C.c_int64(), C.c_int64()
    L = _lib.lib()
    rc = L.gfn_octree_build_host(c2w.ctypes.data_as(C.c_void_p), intri.ctypes.data_as(C.c_void_p),
                                 bounds.ctypes.data_as(C.c_void_p), n, max_depth, bbox_side_len, split_dist_thres, seed,
                                 threads, C.byref(np_), C.byref(nn), C.byref(tp_), C.byref(nt))
    _lib.check(rc, "gfn_octree_build_host")
    try:
        nodes = np.ctypeslib.as_array(C.cast(np_, C.POINTER(C.c_uint8)), shape=(nn.value * TREE_NODE_BYTES,)).copy()
        nt_b = max(nt.value, 1) * TRANS_INFO_BYTES
        transes = np.ctypeslib.as_array(C.cast(tp_, C.POINTER(C.c_uint8)), shape=(nt_b,)).copy()
    finally:
        L.gfn_free_host(np_)
        L.gfn_free_host(tp_)
    return nodes, transes[: nt.value * TRANS_INFO_BYTES] if nt.value else transes[:0]


# ---- pieces of the host octree build (numpy in, numpy out), see include/gfnerf_amd.h
def _np32(a):
    return np.ascontiguousarray(a, np.float32)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def octree_distance_summary(dis):
    d = _np32(dis).reshape(-1)
    return float(_lib.lib().gfn_octree_distance_summary_host(_vp(d), d.size))


def octree_pca12(vals):
    """vals [n,12] -> (evals f32 [12] descending, evecs f32 [12,12], column k = k-th eigenvector)."""
    v = _np32(vals)
    if v.ndim != 2 or v.shape[1] != 12:
        raise _lib.GfnError("octree_pca12: vals must be [n, 12]")
    L, V = np.zeros(12, np.float32), np.zeros((12, 12), np.float32)
    _lib.check(_lib.lib().gfn_octree_pca12_host(_vp(v), v.shape[0], _vp(L), _vp(V)), "gfn_octree_pca12_host")
    return L, V


def octree_visible_cams(c2w, intri, bounds, center, side):
    c2w, intri, bounds, center = _np32(c2w), _np32(intri), _np32(bounds), _np32(center).reshape(3)
    out = np.zeros(c2w.shape[0], np.uint8)
    _lib.check(_lib.lib().gfn_octree_visible_cams_host(_vp(c2w), _vp(intri), _vp(bounds), c2w.shape[0], _vp(center), float(side), _vp(out)),
               "gfn_octree_visible_cams_host")
    return out.astype(bool)


def octree_construct_trans(c2w_visible, intri, center, side, first_cam=-1, pts=None, seed=0):
    """-> (trans_info blob u8 [576], good camera indices i64 [6])."""
    c2w, intri, center = _np32(c2w_visible), _np32(intri).reshape(-1), _np32(center).reshape(3)
    p = None if pts is None else _np32(pts)
    out = np.zeros(TRANS_INFO_BYTES, np.uint8)
    good = np.zeros(6, np.int64)
    rc = _lib.lib().gfn_octree_construct_trans_host(_vp(c2w), _vp(intri), c2w.shape[0], _vp(center), float(side), int(first_cam),
                                                   None if p is None else _vp(p), 0 if p is None else p.shape[0], int(seed), _vp(out), _vp(good))
    _lib.check(rc, "gfn_octree_construct_trans_host")
    return out, good


def octree_fit_weight(w2xz, pts):
    w, p = _np32(w2xz).reshape(12, 2, 4), _np32(pts)
    out = np.zeros((3, 12), np.float32)
    _lib.check(_lib.lib().gfn_octree_fit_weight_host(_vp(w), _vp(p), p.shape[0], _vp(out)), "gfn_octree_fit_weight_host")
    return out

"""End-to-end CPU restatement of GFNeRFModel.get_outputs (gfnerf/nerfacto.py:522-619) built from
the oracle's pieces.  TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg).  `P` is a dict of numpy arrays: nodes/transes blobs, table [16T,2], optional
blocks, prim [16,V,3], bias [16V,3], layers [(W,b)x5] (base x2, head x3), emb [n_cams,32].
"""
import numpy as np

import oracle as O


def _nodes(P):
    return np.ascontiguousarray(P["nodes"]).view(O.TREE_NODE_DTYPE)


def _transes(P):
    return np.ascontiguousarray(P["transes"]).view(O.TRANS_INFO_DTYPE)


def sample(P, rays_o, rays_d_raw, noise, sample_l=1.0 / 256, scale_by_dis=True, near=0.01, max_hits=1024, threads=0):
    """perssampler.generate_ray_samples -> GetSamples (PersSampler_cuda.cu:321-477), packed."""
    d = O.normalize_dirs(rays_d_raw)
    off, hi, nf = O.trace(rays_o, d, _nodes(P), near=near, far=1e8, max_hits=max_hits, threads=threads)
    m = O.march(rays_o, d, noise, sample_l, scale_by_dis, off, hi, nf, _nodes(P), _transes(P), threads=threads)
    m["dirs_unit"] = d
    m["oct_offsets"] = off
    # nerfacto_field.py:431  (sampled_pts + 1.5) / 3.0; torch on the device computes tensor / scalar
    # as a * (1/b) in f32 (ATen/native/cuda/BinaryDivTrueKernel.cu, is_cpu_scalar branch)
    m["x"] = ((m["warped_pts"] + np.float32(1.5)) * (np.float32(1.0) / np.float32(3.0))).astype(np.float32)
    m["ray_id"] = np.repeat(np.arange(rays_o.shape[0]), np.diff(m["pts_offsets"])).astype(np.int64)
    return m


def field_forward(P, m, cam_idx, log2_T, block=None, threads=0, use_emb=True):
    """GFNeRFField.get_density + get_outputs (nerfacto_field.py:412-591)."""
    V = P["prim"].shape[1]
    anchors = np.ascontiguousarray(m["anchors"][:, 0])
    feat = O.hash_fwd(P["table"], P["prim"], P["bias"], m["x"], anchors, V, log2_T, threads=threads)
    if block is not None:
        bprim = P["block_prims"][block] if "block_prims" in P else P["prim"]  # each table hashes with its own primes
        feat = feat + O.hash_fwd(P["blocks"][block], bprim, P["bias"], m["x"], anchors, V, log2_T, threads=threads)
    (bW0, bb0), (bW1, bb1), (hW0, hb0), (hW1, hb1), (hW2, hb2) = P["layers"]
    a1 = O.linear_fwd(feat, bW0, bb0, "relu", threads=threads)
    h = O.linear_fwd(a1, bW1, bb1, "none", threads=threads)
    sigma = O.density_fwd(h)
    d01 = ((m["dirs_unit"] + np.float32(1.0)) * np.float32(0.5)).astype(np.float32)  # nerfacto_field.py:64-70
    sh = O.sh4(d01, convention=0, round_f16=True)
    rid = m["ray_id"]
    emb = P["emb"][np.asarray(cam_idx)[rid]] if use_emb else np.zeros((len(rid), 32), np.float32)
    xh = np.concatenate([sh[rid], h[:, 1:16], emb], axis=1).astype(np.float32)  # :540-547
    c1 = O.linear_fwd(xh, hW0, hb0, "relu", threads=threads)
    c2 = O.linear_fwd(c1, hW1, hb1, "relu", threads=threads)
    rgb = O.linear_fwd(c2, hW2, hb2, "sigmoid", threads=threads)
    return dict(feat=feat, a1=a1, h=h, sigma=sigma, xh=xh, c1=c1, c2=c2, rgb=rgb)


def render(P, rays_o, rays_d_raw, cam_idx, noise, log2_T, block=None, early_T=0.0, scale_factor=10.0, threads=0,
           use_emb=True, **sampler_kw):
    m = sample(P, rays_o, rays_d_raw, noise, threads=threads, **sampler_kw)
    f = field_forward(P, m, cam_idx, log2_T, block, threads, use_emb)
    comp = O.composite_fwd(m["pts_offsets"], f["sigma"], m["dists"], m["ts"], f["rgb"], early_T=early_T, threads=threads)
    depth = O.depth_finalize(comp["depth_raw"], m["pts_offsets"], m["ts"], scale_factor)
    return dict(rgb=np.clip(np.nan_to_num(comp["rgb"]), 0, 1), rgb_raw=comp["rgb"], acc=comp["acc"], depth=depth,
                oct_depth=m["first_oct_dis"] * (np.float32(1.0) / np.float32(scale_factor)), samples=m, field=f, comp=comp)


def loss_and_grads(P, rays_o, rays_d_raw, cam_idx, gt_rgb, noise, log2_T, block=None, train_mlp=True, use_emb=True,
                   s3im_index=None, s3im_mult=1.0, bwd_threads=1, **sampler_kw):
    """Charbonnier loss (nerfacto.py:636) and every parameter gradient, by hand-written backward
    of the same pieces (reference: autograd + Hash3DAnchoredFunction::backward).
    bwd_threads: OpenMP threads of the decoder's backward (1: the serial sums the parity tests use; 0: all cores -- bench.py's
    cpu_baseline); every other piece runs on all cores either way with thread-count-independent results."""
    m = sample(P, rays_o, rays_d_raw, noise, **sampler_kw)
    f = field_forward(P, m, cam_idx, log2_T, block, use_emb=use_emb)
    comp = O.composite_fwd(m["pts_offsets"], f["sigma"], m["dists"], m["ts"], f["rgb"])
    loss, g_rgb = O.charbonnier(comp["rgb"], gt_rgb)
    if s3im_index is not None:  # + s3im_loss_mult * S3IM(image, rgb) (nerfacto.py:687-688, config.py:96)
        l2, g2 = O.s3im(gt_rgb, comp["rgb"], s3im_index, patch_h=32)
        loss = loss + s3im_mult * l2
        g_rgb = (g_rgb + np.float32(s3im_mult) * g2).astype(np.float32)
    d_sigma, d_rgb = O.composite_bwd(m["pts_offsets"], f["sigma"], m["dists"], f["rgb"], g_rgb, None)
    (bW0, _), (bW1, _), (hW0, _), (hW1, _), (hW2, _) = P["layers"]
    grads = {}
    dc2, grads["hW2"], grads["hb2"] = O.linear_bwd(f["c2"], hW2, f["rgb"], d_rgb, "sigmoid", threads=bwd_threads)
    dc1, grads["hW1"], grads["hb1"] = O.linear_bwd(f["c1"], hW1, f["c2"], dc2, "relu", threads=bwd_threads)
    dxh, grads["hW0"], grads["hb0"] = O.linear_bwd(f["xh"], hW0, f["c1"], dc1, "relu", threads=bwd_threads)
    dh = np.zeros_like(f["h"])
    dh[:, 0] = O.density_bwd(f["h"], d_sigma)
    dh[:, 1:16] = dxh[:, 16:31]
    demb = np.zeros(P["emb"].shape, np.float64)
    if use_emb:
        np.add.at(demb, np.asarray(cam_idx)[m["ray_id"]], dxh[:, 31:63].astype(np.float64))
    grads["emb"] = demb.astype(np.float32)
    da1, grads["bW1"], grads["bb1"] = O.linear_bwd(f["a1"], bW1, f["h"], dh, "none", threads=bwd_threads)
    dfeat, grads["bW0"], grads["bb0"] = O.linear_bwd(f["feat"], bW0, f["a1"], da1, "relu", threads=bwd_threads)
    V = P["prim"].shape[1]
    # focal stage (block given): the gradient belongs to the block table and scatters through ITS primes
    gprim = P["prim"] if block is None else P.get("block_prims", [P["prim"]] * (block + 1))[block]
    grads["table"] = O.hash_bwd(gprim, P["bias"], m["x"], np.ascontiguousarray(m["anchors"][:, 0]), dfeat, V, log2_T)
    return loss, grads, dict(samples=m, field=f, comp=comp, dfeat=dfeat)

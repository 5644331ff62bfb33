"""A procedural ground truth for the synthetic "Rubble-scale" capture (scene.py), so that training has something to converge to and
PSNR -- the third part of BASELINE.json's metric, reference: torchmetrics PeakSignalNoiseRatio in gfnerf/nerfacto.py:300,624 -- can
be reported: an analytic terrain (density + colour as closed forms of the world position) seen by the scene's cameras.  The colour
of a ray is the front-to-back quadrature of that field (rays.py:178-200's weights: alpha = 1 - exp(-sigma dt), T = exp(-sum), no
background, renderers.py:109-110), evaluated with plain torch ops on whatever device the rays live on -- the same closed form the
CPU fixture tests/golden/procedural_gt.npz was made with through the oracle's compositing (tests/golden/make_procedural.py).

`train_and_eval` trains the engine (global stage, then one focal block) on rays of this scene with the product's own step and
reports PSNR on held-out rays; `psnr_study` (bench.py's `psnr` sub-record) runs it for the split-f16 `x3` decoder and for the exact-f32
decoder -- the stand-in for the reference's f32 nn.Linear (gfnerf/mlp.py:35-57) -- over several seeds each.  Several, because one
training run is one draw: Adam with eps = 1e-15 (gfnerf/config.py:130) turns the first last-bit difference of a near-zero table gradient
into a step of the full learning rate, and from there the two arithmetics (or two seeds of one) follow different trajectories whose
PSNR scatters by ~0.5 dB (measured: tools/exp_x3_vs_f32_traj.py trains both in lockstep -- single steps from a common state agree to
1e-6 in every gradient, tools/exp_x3_vs_f32_step.py).  The two decoders are compared through their means."""
import math

import numpy as np
import torch

from . import hotpath, ops, scene

Z_TOP, Z_BOTTOM, N_QUAD = 4.0, -4.0, 384


def height(x, y):
    """Terrain height above z = 0, |h| < 2.3 world units."""
    return 1.5 * torch.sin(0.35 * x) * torch.cos(0.27 * y) + 0.8 * torch.sin(0.9 * x + 1.3) * torch.sin(0.7 * y + 0.4)


def field(xyz):
    """xyz [..., 3] world coordinates -> (sigma [...], rgb [..., 3]): opaque below the terrain surface (optical depth 6 per unit),
    coloured in smooth bands of position and height."""
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    h = height(x, y)
    sigma = 6.0 * torch.sigmoid(4.0 * (h - z))
    r = 0.5 + 0.5 * torch.sin(0.8 * x + 0.3 * h)
    g = 0.5 + 0.5 * torch.sin(0.6 * y + 1.0 + 0.5 * h)
    b = 0.5 + 0.5 * torch.cos(0.4 * (x + y)) * torch.cos(1.1 * h)
    return sigma, torch.stack([r, g, b], -1)


def quadrature_points(o, d):
    """Sample positions of the ground-truth quadrature: N_QUAD midpoints of equal steps between the planes z = Z_TOP and z = Z_BOTTOM
    along each (downward) ray.  o, d [R,3] (d need not be normalised) -> (xyz [R,N,3], dt [R,N] in units of |d|-normalised length)."""
    dn = d / d.norm(dim=-1, keepdim=True)
    dz = dn[:, 2].clamp(max=-1e-3)  # the capture looks down (tilt <= 15 degrees + the field of view)
    t0 = ((Z_TOP - o[:, 2]) / dz).clamp(min=0.0)
    t1 = (Z_BOTTOM - o[:, 2]) / dz
    step = (t1 - t0) / N_QUAD
    k = torch.arange(N_QUAD, device=o.device, dtype=o.dtype) + 0.5
    t = t0[:, None] + step[:, None] * k[None, :]
    return o[:, None, :] + t[..., None] * dn[:, None, :], step[:, None].expand(-1, N_QUAD)


def gt_colors(o, d):
    """Ground-truth colour [R,3] of rays (o, d): front-to-back compositing of `field` over `quadrature_points`, black background."""
    xyz, dt = quadrature_points(o, d)
    sigma, rgb = field(xyz)
    ds = sigma * dt
    T = torch.exp(-(torch.cumsum(ds, 1) - ds))
    w = (1.0 - torch.exp(-ds)) * T
    return (w[..., None] * rgb).sum(1)


def psnr(pred, gt):
    return float(-10.0 * torch.log10(torch.mean((pred - gt) ** 2)))


def train_and_eval(sc, precision="x3", steps_global=2000, steps_focal=1000, n_rays=4096, log2_T=21, seed=0, n_pool=64, eval_rays=16384, target_samples=128,
                   early_T=1e-4, progress=None, update_oct=True, **cfg_kw):
    """Train on the procedural scene with the product's native step (HotPath.train_step: sampler -> lookup -> decoder -> compositing ->
    Charbonnier + S3IM -> backward -> Adam; UpdateOctNodes every global step like GFNeRFModel.get_outputs), lr 1e-2 -> 1e-4 over the global
    stage and 5e-3 -> 5e-5 over the focal stage (GFNerfExponentialDecayScheduler's shape, schedulers.py:113-184), then render held-out rays
    of the training cameras.  -> dict(psnr_global, psnr, loss, ...)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = hotpath.HotPathConfig(log2_T=log2_T, mlp_precision=precision, early_T=early_T, **cfg_kw)
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=seed, n_blocks=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    pool = []
    for k in range(n_pool):  # a resident pool of training batches (rays + ground truth), cycled
        o, d, cam = scene.make_rays(sc["c2w"], n_rays, seed=seed * 100003 + k)
        o, d, cam = t(o), t(d), t(cam)
        pool.append((o, d, cam, gt_colors(o, d)))
    eo, ed, ecam = scene.make_rays(sc["c2w"], eval_rays, seed=seed * 100003 + 7_000_000)
    eo, ed, ecam = t(eo), t(ed), t(ecam)
    egt = gt_colors(eo, ed)
    # ray-march fineness for the sample budget, held (UpdateRayMarch anneals 16 -> 1 over 10,000 steps, PersSampler.cpp:958-967)
    lo, hi = 0.5, 64.0
    for _ in range(14):
        fin = (lo * hi) ** 0.5
        mean = hp.sample(pool[0][0], pool[0][1], ops.sampler_noise(n_rays, fin, train=False, device=dev))["pts_offsets"][-1].item() / n_rays
        if abs(mean - target_samples) / target_samples < 0.01:
            break
        lo, hi = (fin, hi) if mean > target_samples else (lo, fin)
    gen = torch.Generator(device=dev).manual_seed(seed + 11)

    def evaluate(block):
        out = []
        for a in range(0, eval_rays, n_rays):
            sl = slice(a, a + n_rays)
            noise = ops.sampler_noise(n_rays, fin, train=False, device=dev)
            out.append(hp.render(eo[sl].contiguous(), ed[sl].contiguous(), ecam[sl].contiguous(), noise, block=block)["rgb"])
        return psnr(torch.cat(out), egt)

    res = dict(precision=precision, fineness=fin, mean_samples_per_ray=mean, steps_global=steps_global, steps_focal=steps_focal)
    loss = None
    for stage, steps, lr0, block in (("global", steps_global, cfg.lr, None), ("focal", steps_focal, cfg.lr_block, 0)):
        for i in range(steps):
            o, d, cam, gt = pool[i % n_pool]
            mult = math.exp(math.log(1e-2) * i / max(steps, 1))  # lr_init -> lr_init / 100 over the stage
            if stage == "global":
                hp.cfg.lr = lr0 * mult
            else:
                hp.cfg.lr_block = lr0 * mult
            noise = ops.sampler_noise(n_rays, fin, train=True, generator=gen, device=dev)
            s3 = ops.s3im_index(n_rays, cfg.s3im_repeat_time, generator=gen, device=dev)
            loss, _ = hp.train_step(o, d, cam, gt, noise, stage=stage, block=block, s3im_index=s3, update_oct=update_oct and stage == "global")
            if progress is not None and (i + 1) % 500 == 0:
                progress(f"{precision} {stage} step {i + 1}/{steps} loss {float(loss):.4f}")
        if stage == "global":
            hp.cfg.lr = lr0
            res["psnr_global"] = evaluate(None)
            res["leaves_pruned"] = int((hp.nodes.view(-1, 128)[:, 96:104].contiguous().view(torch.int64) < 0).sum().item())
        else:
            hp.cfg.lr_block = lr0
    res["psnr"] = evaluate(0) if steps_focal > 0 else res["psnr_global"]
    res["loss"] = None if loss is None else float(loss)
    n = None
    comp, ctx = hp.forward_train(pool[0][0], pool[0][1], pool[0][2], ops.sampler_noise(n_rays, fin, train=False, device=dev), stage="focal" if steps_focal > 0 else "global",
                                 block=0 if steps_focal > 0 else None)
    n = int(ctx["samples"]["pts_offsets"][-1].item())
    res["alive_sample_frac_trained"] = float((comp["weights"][:n] > 0).float().mean())
    if hp._phase is not None:
        res["planned_K_trained"] = int(hp._phase.k.item())
    return res


def psnr_study(sc, seeds=(0, 1), precisions=("x3", "f32"), progress=None, **kw):
    """train_and_eval per (decoder arithmetic, seed) -> dict(psnr_<p> = mean, psnr_<p>_runs = [...], psnr_<p>_std, delta_x3_minus_f32 = mean over
    the seeds of PSNR(x3) - PSNR(f32), delta_runs, delta_sigma = the standard error of that mean)."""
    out = dict(seeds=list(seeds), workload="procedural terrain seen by the synthetic capture's cameras (gfnerf_amd.procedural); PSNR on 16,384 held-out rays")
    runs = {}
    for p in precisions:
        runs[p] = []
        for sd in seeds:
            r = train_and_eval(sc, precision=p, seed=sd, **kw)
            runs[p].append(r)
            if progress is not None:
                progress(f"psnr {p} seed {sd}: global {r['psnr_global']:.3f} final {r['psnr']:.3f}")
        vals = np.array([r["psnr"] for r in runs[p]])
        out[f"psnr_{p}"] = float(vals.mean())
        out[f"psnr_{p}_runs"] = [round(float(v), 3) for v in vals]
        out[f"psnr_{p}_std"] = float(vals.std(ddof=1)) if len(vals) > 1 else None
        out[f"psnr_global_{p}"] = float(np.mean([r["psnr_global"] for r in runs[p]]))
    r0 = runs[precisions[0]][0]
    out.update(steps=[r0["steps_global"], r0["steps_focal"]], mean_samples_per_ray=r0["mean_samples_per_ray"],
               alive_sample_frac_trained=r0.get("alive_sample_frac_trained"), planned_K_trained=r0.get("planned_K_trained"))
    if "x3" in runs and "f32" in runs:  # paired by seed: the same initialisation, batches and random draws for both decoders
        dl = np.array([a["psnr"] - b["psnr"] for a, b in zip(runs["x3"], runs["f32"])])
        out["delta_x3_minus_f32"] = float(dl.mean())
        out["delta_runs"] = [round(float(v), 3) for v in dl]
        if len(seeds) > 1:
            out["delta_sigma"] = float(dl.std(ddof=1) / math.sqrt(len(dl)))  # standard error of the mean paired difference
    return out

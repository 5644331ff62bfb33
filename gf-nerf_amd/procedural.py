"""A procedural ground truth for the synthetic "Rubble-scale" capture (scene.py), so that training has something to converge to and
PSNR -- the third part of BASELINE.json's metric, reference: torchmetrics PeakSignalNoiseRatio in gfnerf/nerfacto.py:300,624 -- can
be reported: an analytic terrain (density + colour as closed forms of the world position) seen by the scene's cameras.  The colour
of a ray is the front-to-back quadrature of that field (rays.py:178-200's weights: alpha = 1 - exp(-sigma dt), T = exp(-sum), no
background, renderers.py:109-110), evaluated with plain torch ops on whatever device the rays live on -- the same closed form the
CPU fixture tests/golden/procedural_gt.npz was made with through the oracle's compositing (tests/golden/make_procedural.py).

`train_and_eval` trains the engine (global stage, then the focal block(s)) on rays of this scene with the product's own step and reports PSNR on
held-out rays; `psnr_study` (bench.py's `psnr` sub-record, tools/psnr_study.py) runs it for two LEGS over several seeds, paired by seed: "fast" =
the product as shipped, "reference" = the reference's training semantics (every sample, exact-f32 decoder, f16 atomic adds, torch.optim.Adam).
Round 4's protocol (PROTOCOL, STUDY_*): fresh random rays every step, one fixed held-out set for every seed and leg, a five-octave colour texture in
the ground truth, 3,000 + 1,000 steps with the rate decayed x 0.1 per stage.  Several seeds, because one training run is one draw: Adam with
eps = 1e-15 (gfnerf/config.py:130) turns the first last-bit difference of a near-zero gradient into a step of the full learning rate, Charbonnier's
gradient is sign-like, and from there two runs (or two arithmetics of one) follow different trajectories whose PSNR scatters by ~0.5 dB; single steps
from a common state agree (grad_cosine; tools/exp_x3_vs_f32_step.py).  The legs are compared through their means over many seeds
(profiles/r04_psnr_study.json) and, without any training chaos, through the same trained states rendered with either decoder."""
import gc
import math

import numpy as np
import torch

from . import hotpath, ops, scene

Z_TOP, Z_BOTTOM, N_QUAD = 4.0, -4.0, 384
PROTOCOL = "r04b: fresh rays every step, one fixed held-out set, texture 0.5, 3000 + 1000 steps, lr x0.1 per stage, legs fast | reference"  # bumped when the study's meaning changes
# (round 6 adds the leg "plugin" under the same protocol: the same rays, schedule and held-out set, trained through the plugin surface)
STUDY_STEPS = (3000, 1000)  # global, focal steps of a study run
STUDY_LR_FINAL = 0.1        # learning-rate decay over each stage (1e-2 -> 1e-3, 5e-3 -> 5e-4).  With the rounds 2-3 schedule (2,000 steps, x0.01) the rate
                            # collapsed before the initial fog had cleared in some runs: a bimodal outcome (crisp ~32 dB | foggy 27-29 dB, chosen by
                            # the seed and by chaos) whose pair-to-pair scatter of 1.1 dB no affordable number of seeds averages out; this schedule
                            # brings every seed to 33-34 dB (tools/README.md, round 4)
STUDY_DETAIL = 0.5  # texture amplitude of the PSNR study: 3,000-step runs plateau at ~32 dB (without it: 45-52 dB, where PSNR is optimiser noise)


def height(x, y):
    """Terrain height above z = 0, |h| < 2.3 world units."""
    return 1.5 * torch.sin(0.35 * x) * torch.cos(0.27 * y) + 0.8 * torch.sin(0.9 * x + 1.3) * torch.sin(0.7 * y + 0.4)


TEX_WAVELENGTHS = (1.6, 0.8, 0.4, 0.2, 0.1)  # world units; a pixel of the capture covers ~0.012 units of ground


def texture(x, y, z, c):
    """Fine detail of colour channel c in [-1, 1]: five octaves of oriented ripples (wavelengths TEX_WAVELENGTHS, amplitude ~ sqrt(wavelength),
    directions and phases fixed per octave and channel) -- the rubble on the terrain: detail a 3,000-step run resolves only in part,
    which is what bounds PSNR on a real capture (Mega-NeRF's Rubble: 24-26 dB), not optimiser noise."""
    v, norm = 0.0, 0.0
    for k, lam in enumerate(TEX_WAVELENGTHS):
        f = 2.0 * math.pi / lam
        th, ph, ps = 0.7 + 2.399963 * (3 * k + c), 1.3 * (k + 1) * (c + 1), 0.9 * (k + 2) * (c + 1)
        a = math.sqrt(lam / TEX_WAVELENGTHS[0])
        v = v + a * torch.sin(f * (x * math.cos(th) + y * math.sin(th)) + ph) * torch.cos(0.5 * f * z + ps)
        norm += a
    return v / norm


def field(xyz, detail=0.0):
    """xyz [..., 3] world coordinates -> (sigma [...], rgb [..., 3]): opaque below the terrain surface (optical depth 6 per unit),
    coloured in smooth bands of position and height; detail > 0 adds detail * texture() to every channel (clamped to [0, 1])."""
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    h = height(x, y)
    sigma = 6.0 * torch.sigmoid(4.0 * (h - z))
    r = 0.5 + 0.5 * torch.sin(0.8 * x + 0.3 * h)
    g = 0.5 + 0.5 * torch.sin(0.6 * y + 1.0 + 0.5 * h)
    b = 0.5 + 0.5 * torch.cos(0.4 * (x + y)) * torch.cos(1.1 * h)
    rgb = torch.stack([r, g, b], -1)
    if detail > 0:
        rgb = (rgb + detail * torch.stack([texture(x, y, z, c) for c in range(3)], -1)).clamp(0.0, 1.0)
    return sigma, rgb


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


def gt_colors(o, d, detail=0.0):
    """Ground-truth colour [R,3] of rays (o, d): front-to-back compositing of `field` over `quadrature_points`, black background."""
    xyz, dt = quadrature_points(o, d)
    sigma, rgb = field(xyz, detail)
    ds = sigma * dt
    T = torch.exp(-(torch.cumsum(ds, 1) - ds))
    w = (1.0 - torch.exp(-ds)) * T
    return (w[..., None] * rgb).sum(1)


def psnr(pred, gt):
    return float(-10.0 * torch.log10(torch.mean((pred - gt) ** 2)))


LEGS = {
    # the product as shipped: split-f16 x3 MFMA decoder, early termination at T < 1e-4, phased forward, LDS fixed-point gradient accumulation, GfnAdam's kernel
    "fast": dict(precision="x3"),
    "x3": dict(precision="x3"),
    # the product with the exact-f32 decoder (isolates the decoder arithmetic inside the product's own training semantics)
    "f32": dict(precision="f32"),
    # the reference's training semantics (engine/trainer.py:382-441 + gfnerf/nerfacto.py:522-619,628-689): every sample evaluated and composited
    # (no transmittance cut), exact-f32 nn.Linear decoder (gfnerf/mlp.py:35-57), table gradient through f16 x128 atomic adds
    # (Hash3DAnchored_cuda.cu:151,209,222), torch.optim.Adam per parameter group (engine/optimizers.py:100-133)
    "reference": dict(precision="f32", early_T=0.0, two_phase=False, lds_accumulate=False, torch_adam=True),
    # the product AS THE PLUGIN SURFACE BUILDS AND DRIVES IT: GFNerfPipeline.get_train_loss_dict -> backward -> Optimizers.optimizer_step_all ->
    # scheduler_step_all with the callbacks of GFNeRFModel (engine/trainer.py:382-441) -- the engine gfnerf/nerfacto.py configures (record_merge_dense,
    # replayed forward + backward, the fused loss node, GfnAdam through Optimizers), on the same rays, schedule and held-out set as the other legs
    "plugin": dict(plugin=True),
    # "reference" with the reference's block-stage sampler mode (train_and_eval focal_jitter=False): the leg the plugin surface is to be held against -- the
    # surface mirrors that mode, and it alone is worth +0.198 +- 0.006 dB of focal-stage gain on this scene (training and evaluation then sample alike: profiles/r06_psnr_study.json)
    "reference_m1": dict(precision="f32", early_T=0.0, two_phase=False, lds_accumulate=False, torch_adam=True, focal_jitter=False),
}


def device_rays(c2w, n, gen, cams=None):
    """scene.make_rays on the device: uniformly random (camera, pixel) rays of the capture, fresh every step (a training set of
    1,600 x 1 Mpixel is never cycled inside a few thousand steps).  c2w f32 [n_cams,3,4] on the device; cams: int64 indices to draw from."""
    dev = c2w.device
    if cams is None:
        cam = torch.randint(0, c2w.shape[0], (n,), generator=gen, device=dev)
    else:
        cam = cams[torch.randint(0, cams.numel(), (n,), generator=gen, device=dev)]
    uv = torch.rand((n, 2), generator=gen, device=dev, dtype=torch.float32)
    px, py = uv[:, 0] * scene.WIDTH, uv[:, 1] * scene.HEIGHT
    d_cam = torch.stack([(px - scene.WIDTH / 2) / scene.FOCAL, -(py - scene.HEIGHT / 2) / scene.FOCAL, -torch.ones_like(px)], -1)
    d = torch.einsum("nij,nj->ni", c2w[cam, :, :3], d_cam)
    d = d / d.norm(dim=-1, keepdim=True)
    return c2w[cam, :, 3].contiguous(), d.contiguous(), cam.to(torch.int32)


class TorchAdam:
    """torch.optim.Adam per parameter group on the engine's parameters -- Optimizers.optimizer_scaler_step_all (engine/optimizers.py:100-133) --
    fed the dense f32 gradients the reference's autograd would hold: the table's = the f16 x128 pool / 128 (Hash3DAnchoredFunction::backward,
    Hash3DAnchored_cuda.cu:209-240)."""

    def __init__(self, hp):
        self.hp, self.opt = hp, {}

    def _group(self, name):
        hp = self.hp
        if name not in self.opt:
            n_live = 17 * (1 << hp.cfg.log2_T)
            if name == "table" or name.startswith("block"):
                p = (hp.table if name == "table" else hp.blocks[int(name[5:])]).view(-1)[:n_live]
                grad = lambda: hp.table_grad.view(-1)[:n_live].float() / (128.0 if hp.table_grad.dtype == torch.float16 else 1.0)  # noqa: E731
            elif name == "mlp":
                p, grad = hp.mlp_flat, lambda: hp.mlp_grad
            else:
                p, grad = hp.emb, lambda: hp.emb_grad
            p = torch.nn.Parameter(p, requires_grad=False)  # a leaf that aliases the engine's storage
            self.opt[name] = (p, grad, torch.optim.Adam([p], lr=1e-2, eps=hp.cfg.adam_eps))
        return self.opt[name]

    def step(self, groups):
        """groups: [(name, lr)] (HotPath.adam_groups with the schedule's rate)."""
        hp = self.hp
        hp._sync_opt()
        for name, lr in groups:
            p, grad, opt = self._group(name)
            p.grad = grad()
            opt.param_groups[0]["lr"] = lr
            opt.step()
        hp.step += 1
        hp.refresh_shadows()  # f16 shadows of the tables + the decoder's weight images follow the new values


GRADIENT_LEGS = {  # HotPathConfig fields of the gradient semantics grad_cosine compares from one state
    "fast": dict(mlp_precision="x3", early_T=1e-4, two_phase=True, lds_accumulate=True, grad_f16=True),         # the product: x3 decoder, cut at T < 1e-4, exact LDS sums
    "reference": dict(mlp_precision="f32", early_T=0.0, two_phase=False, lds_accumulate=False, grad_f16=True),  # every sample, exact-f32 decoder, f16 x128 atomics
    "exact": dict(mlp_precision="f32", early_T=0.0, two_phase=False, lds_accumulate=False, grad_f16=False),     # the same with an f32 gradient pool: the gradient both approximate
    "f32": dict(mlp_precision="f32", early_T=1e-4, two_phase=True, lds_accumulate=True, grad_f16=True),         # the product with the exact-f32 decoder only
    "fast0": dict(mlp_precision="x3", early_T=0.0, two_phase=True, lds_accumulate=True, grad_f16=True),         # the product without the cut (every sample)
}


def grad_cosine(hp, batch, noise, s3, stage, block, legs=("reference", "fast")):
    """One forward + backward of `batch` from the engine's current state under each of the gradient semantics `legs` (GRADIENT_LEGS; default: the
    reference's -- every sample, exact-f32 decoder, f16 atomic adds -- and the product's -- x3 decoder, early termination, phased forward,
    exact LDS accumulation): cosine and relative difference of the gradients per parameter group, for every pair of legs (no optimiser step,
    no octree update).  -> {"a|b": {group: {cos, rel_diff (relative to a)}}}."""
    o, d, cam, gt = batch
    c = hp.cfg
    keys = ("mlp_precision", "early_T", "two_phase", "lds_accumulate", "grad_f16")
    saved = {k: getattr(c, k) for k in keys}
    saved_pool = hp.table_grad
    got = {}
    n_live = 17 * (1 << c.log2_T)
    try:
        for leg in legs:
            for k, v in GRADIENT_LEGS[leg].items():
                setattr(c, k, v)
            if hp.table_grad is not None and (hp.table_grad.dtype == torch.float16) != c.grad_f16:
                hp.table_grad = None  # (allocated again in this leg's pool type by the forward's zero_grads)
            hp._x3_stale = True
            hp.forward_backward(o, d, cam, gt, noise, stage=stage, block=block, s3im_index=s3, update_oct=False)
            hp._sync_opt()
            scale = 1.0 / 128.0 if hp.table_grad.dtype == torch.float16 else 1.0
            g = dict(table=hp.table_grad.view(-1)[:n_live].double() * scale)
            if stage == hotpath.STAGE_GLOBAL:
                g.update(mlp=hp.mlp_grad.double(), emb=hp.emb_grad.double())
            got[leg] = {k: v.clone() for k, v in g.items()}
    finally:
        for k, v in saved.items():
            setattr(c, k, v)
        if saved_pool is not None and (hp.table_grad is None or hp.table_grad.dtype != saved_pool.dtype):
            hp.table_grad = saved_pool
        hp._x3_stale = True
    out = {}
    for i, la in enumerate(legs):
        for lb in legs[i + 1:]:
            pair = {}
            for k in got[la]:
                a, b = got[la][k], got[lb][k]
                na, nb = float(a.norm()), float(b.norm())
                pair[k] = dict(cos=float((a * b).sum()) / max(na * nb, 1e-300), rel_diff=float((a - b).norm()) / max(na, 1e-300))
            out[f"{la}|{lb}"] = pair
    return out


class _ProceduralDataManager:
    """The part of the host's datamanager GFNerfPipeline touches (gf_pipeline.py:94-192), fed by a batch function: next_train(step) -> fresh rays of the
    procedural capture with their ground-truth colours (no image stack, hence no error map: uniform rays, like the other legs)."""

    def __init__(self, c2w, intri, batch_fn, device):
        from types import SimpleNamespace

        from .gfnerf.gf_pipeline import _TrainDataset
        from .gfnerf.perssampler import _Cameras

        self.device = torch.device(device)
        self.batch_fn = batch_fn
        self.train_dataparser_outputs = SimpleNamespace(cameras=_Cameras(torch.as_tensor(np.ascontiguousarray(c2w)).float(), torch.as_tensor(np.ascontiguousarray(intri)).float()))
        self.train_dataset = _TrainDataset(c2w.shape[0], scene_box=SimpleNamespace(aabb=torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])))
        self.eval_dataset = self.train_dataset
        self.split_version = 0

    def to(self, device):
        return self

    def get_training_callbacks(self, attrs):
        return []

    def get_param_groups(self):
        return {}

    def next_train(self, step):
        from .gfnerf.rays import RayBundle

        o, d, cam, gt = self.batch_fn(step)
        steps = torch.full((o.shape[0], 1), int(step), dtype=torch.int64, device=o.device)
        bundle = RayBundle(origins=o, directions=d, camera_indices=cam[:, None], rel_camera_indices=cam[:, None], steps=steps)
        bundle._step_host = (steps, int(step))
        return bundle, {"image": gt}, None


class _DataManagerConfig:
    def __init__(self, make):
        self.make = make

    def setup(self, **kwargs):
        return self.make(kwargs.get("device", "cuda:0"))


def _plugin_trainer(sc, seed, steps_global, steps_focal, n_rays, log2_T, early_T, lr_final, batch_fn, cfg_kw):
    """The pipeline, optimizers, schedulers and callbacks of the `gf-nerf` method (gfnerf/config.py) on the procedural capture, and one Trainer iteration
    (engine/trainer.py:195-231,382-441) as a function of the step.  The study's schedule needs one scheduler per stage -- 1e-2 -> 1e-2 lr_final over the
    global stage for "fields" / "base_encoding_init", 5e-3 -> 5e-3 lr_final over block 0's period for "base_encoding_0" -- which Optimizers takes as
    per-group configs (a group WITHOUT its own takes config["fields"], engine/optimizers.py:86-87)."""
    import functools

    from .gfnerf._host import Optimizers, TrainingCallbackAttributes, TrainingCallbackLocation
    from .gfnerf.gf_pipeline import GFNerfPipelineConfig
    from .gfnerf.nerfacto import GFNeRFModelConfig
    from .gfnerf.optim import GfnAdamOptimizerConfig
    from .gfnerf.schedulers import GFNerfExponentialDecaySchedulerConfig

    mk = dict(log2_hashmap_size=log2_T, hidden_dim=128, hidden_dim_color=128, steps_perssampler_init=steps_global, steps_per_split_dataset=max(steps_focal, 1),
              n_split_dataset=1, n_blocks=1, scale_factor=10.0, use_appearance_embedding=True, use_ch_loss=True, s3im_loss_mult=1.0, early_T=early_T,
              engine_seed=seed, octree_seed=1234)  # (the scene's own tree for every seed: the legs differ in the training path, not in the octree's probe points)
    pcfg = GFNerfPipelineConfig(datamanager=_DataManagerConfig(lambda dev: _ProceduralDataManager(sc["c2w"], sc["intri"], batch_fn, dev)), model=GFNeRFModelConfig(**mk))
    # base_dir of its OWN: the field resumes a block's table (and primes) from base_dir/encodings_ckpt when the block becomes active (nerfacto_field.py
    # load_table, the reference's block swap) and saves it there at the end of the block's period -- with one directory shared by the runs of a study every
    # run but the first started its focal stage from the PREVIOUS seed's residual table (round 6: focal gains of 0.2 ... 1.2 dB by run order)
    import tempfile
    base_dir = tempfile.mkdtemp(prefix="gfn_plugin_run_")
    pipe = pcfg.setup(device="cuda:%d" % torch.cuda.current_device(), test_mode="val", world_size=1, local_rank=0, base_dir=base_dir)
    pipe._gfn_run_dir = base_dir
    model = pipe.model
    hp = model.hot
    assert hp.n_nodes * 128 == np.asarray(sc["nodes"]).size, "the plugin's sampler built another tree than the scene's"
    for k, v in cfg_kw.items():
        setattr(hp.cfg, k, v)
    sched = lambda lr0, n: GFNerfExponentialDecaySchedulerConfig(lr_final=lr0 * lr_final, max_steps=max(n, 1), n_split_dataset=1, n_dataset_circles=1,  # noqa: E731
                                                                 steps_per_split_dataset=max(steps_focal, 1), steps_perssampler_init=max(n, 1))
    group = lambda lr0, n: {"optimizer": GfnAdamOptimizerConfig(lr=lr0, eps=hp.cfg.adam_eps), "scheduler": sched(lr0, n)}  # noqa: E731
    optimizers = Optimizers({"fields": group(hp.cfg.lr, steps_global), "base_encoding_init": group(hp.cfg.lr, steps_global), "base_encoding_0": group(hp.cfg.lr_block, steps_focal)},
                            pipe.get_param_groups())
    callbacks = pipe.get_training_callbacks(TrainingCallbackAttributes(optimizers=optimizers, grad_scaler=None, pipeline=pipe))
    pipe.train()
    nat = model.persampler.sampler
    nat.compact_freq_, nat.sub_div_milestones_ = 1 << 30, []  # UpdateOctNodes (K5 / K6) every global step, no ProcOctree: what every leg of the study does
    nat.generator.manual_seed(seed + 11)

    def iteration(step):
        for cb in callbacks:
            cb.run_callback_at_location(step, location=TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
        optimizers.zero_grad_all()
        _, loss_dict, _ = pipe.get_train_loss_dict(step=step)
        loss = functools.reduce(torch.add, loss_dict.values())
        loss.backward()
        optimizers.optimizer_step_all()
        optimizers.scheduler_step_all(step)
        for cb in callbacks:
            cb.run_callback_at_location(step, location=TrainingCallbackLocation.AFTER_TRAIN_ITERATION)
        return loss

    return pipe, hp, nat, iteration


def train_and_eval(sc, precision="x3", steps_global=2000, steps_focal=1000, n_rays=4096, log2_T=21, seed=0, n_pool=64, eval_rays=16384, target_samples=128,
                   early_T=1e-4, progress=None, update_oct=True, torch_adam=False, fresh_rays=True, data_seed=None, eval_seed=0, lr_final=1e-2,
                   holdout_every=0, cosine_every=0, cosine_legs=("reference", "fast"), keep_engine=False, detail=0.0, n_blocks=1, after_global=None, lr_hold=0,
                   plugin=False, focal_jitter=True, **cfg_kw):
    """Train on the procedural scene with the product's native step (HotPath.train_step: sampler -> lookup -> decoder -> compositing ->
    Charbonnier + S3IM -> backward -> Adam; UpdateOctNodes every global step like GFNeRFModel.get_outputs), lr 1e-2 -> 1e-2 * lr_final over the
    global stage and 5e-3 -> 5e-3 * lr_final over the focal stage (GFNerfExponentialDecayScheduler's shape, schedulers.py:113-184), then render
    held-out rays.  -> dict(psnr_global, psnr, loss, ...).

    focal_jitter=False: the focal stage marches WITHOUT jitter from its second step on -- what the reference's block stage does: get_outputs ends every
    block-stage iteration with persampler.update_mode(1) (gfnerf/nerfacto.py:614-616), and in that mode GetSamples' rays_noise is ones * fineness
    (PersSampler.cpp mode_, PersSampler_cuda.cu:380-389); the plugin surface follows it (gfnerf/perssampler.py make_noise), the native legs jitter by default.
    lr_hold: global-stage steps at the initial learning rate before the decay starts.
    n_blocks: focal blocks resident; the focal stage trains them in rotation (step k: block k mod n_blocks), PSNR is block 0's.
    after_global(hp, lr_end): called between the stages (bench.py times the global step on the trained state there).
    detail: amplitude of the fine colour texture (procedural.texture) in the ground truth.
    seed: the model's initialisation (tables, primes, decoder, embedding) and the march noise; data_seed (default: seed): the training rays;
    eval_seed: the held-out rays -- ONE fixed set for every seed by default, so that runs are compared on the same pixels.
    fresh_rays: new random rays every step (device_rays); False: a resident pool of n_pool batches, cycled (rounds 2-3).
    torch_adam: step the parameters with torch.optim.Adam on dense gradients (TorchAdam) instead of the engine's Adam kernel.
    holdout_every = k > 0: cameras k-1, 2k-1, ... give no training ray; `psnr_heldout` renders their pixels with the nearest training
    camera's appearance embedding (nerfacto_field.py:533: "test mode use nearest rel camera indices").
    cosine_every = n > 0: every n steps, grad_cosine of the batch at hand under the two gradient semantics `cosine_legs` (GRADIENT_LEGS)
    (res["grad_cosine"] = [(stage, step, {group: {cos, rel_diff}})]).
    After training the SAME state is also rendered with the other decoder arithmetic (psnr_eval_x3 / psnr_eval_f32): the inference-only
    difference, free of training chaos."""
    dev = torch.device("cuda", torch.cuda.current_device())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    plug = None
    if plugin:  # the engine is the one the plugin surface builds (gfnerf/nerfacto.py); `batch` below feeds its datamanager
        if not fresh_rays or n_blocks != 1 or torch_adam or cosine_every:
            raise ValueError("train_and_eval(plugin=True): fresh rays, one block, the surface's own optimiser, no gradient-cosine probes")
        plug = _plugin_trainer(sc, seed, steps_global, steps_focal, n_rays, log2_T, early_T, lr_final, lambda step: batch(step), cfg_kw)
        hp = plug[1]
        cfg = hp.cfg
    else:
        cfg = hotpath.HotPathConfig(log2_T=log2_T, mlp_precision=precision, early_T=early_T, **cfg_kw)
        hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=seed, n_blocks=n_blocks)
    data_seed = seed if data_seed is None else data_seed
    c2w = t(sc["c2w"])
    n_cams = c2w.shape[0]
    all_cams = torch.arange(n_cams, device=dev)
    if holdout_every > 0:
        held = all_cams[holdout_every - 1:: holdout_every]
        train_cams = all_cams[(all_cams + 1) % holdout_every != 0]
    else:
        held, train_cams = None, None
    dgen = torch.Generator(device=dev).manual_seed(data_seed * 100003 + 17)
    pool = []
    if not fresh_rays:
        for k in range(n_pool):  # a resident pool of training batches (rays + ground truth), cycled
            o, d, cam = scene.make_rays(sc["c2w"], n_rays, seed=data_seed * 100003 + k, cams=None if train_cams is None else train_cams.cpu().numpy())
            o, d, cam = t(o), t(d), t(cam)
            pool.append((o, d, cam, gt_colors(o, d, detail)))

    def batch(i):
        if not fresh_rays:
            return pool[i % n_pool]
        o, d, cam = device_rays(c2w, n_rays, dgen, cams=train_cams)
        return o, d, cam, gt_colors(o, d, detail)

    egen = torch.Generator(device=dev).manual_seed(eval_seed * 100003 + 7_000_000)
    eo, ed, ecam = device_rays(c2w, eval_rays, egen, cams=train_cams)
    egt = gt_colors(eo, ed, detail)
    if held is not None:
        ho, hd, hcam = device_rays(c2w, eval_rays, egen, cams=held)
        hgt = gt_colors(ho, hd, detail)
        pos = c2w[:, :, 3]
        near = torch.cdist(pos[held], pos[train_cams]).argmin(1)  # nearest training camera of every held-out camera
        lut = torch.zeros(n_cams, dtype=torch.int64, device=dev)
        lut[held] = train_cams[near]
        hemb = lut[hcam.long()].to(torch.int32)
    # ray-march fineness for the sample budget, held (UpdateRayMarch anneals 16 -> 1 over 10,000 steps, PersSampler.cpp:958-967)
    lo, hi = 0.5, 64.0
    fo, fd = eo[:n_rays].contiguous(), ed[:n_rays].contiguous()
    for _ in range(14):
        fin = (lo * hi) ** 0.5
        mean = hp.sample(fo, fd, ops.sampler_noise(n_rays, fin, train=False, device=dev))["pts_offsets"][-1].item() / n_rays
        if abs(mean - target_samples) / target_samples < 0.01:
            break
        lo, hi = (fin, hi) if mean > target_samples else (lo, fin)
    gen = torch.Generator(device=dev).manual_seed(seed + 11)

    def evaluate(block, rays=None, prec=None):
        o_, d_, cam_, gt_ = (eo, ed, ecam, egt) if rays is None else rays
        saved = hp.cfg.mlp_precision
        if prec is not None and prec != saved:
            hp.cfg.mlp_precision = prec
            hp._x3_stale = True
        out = []
        for a in range(0, o_.shape[0], n_rays):
            sl = slice(a, a + n_rays)
            noise = ops.sampler_noise(n_rays, fin, train=False, device=dev)
            out.append(hp.render(o_[sl].contiguous(), d_[sl].contiguous(), cam_[sl].contiguous(), noise, block=block)["rgb"])
        hp.cfg.mlp_precision = saved
        img = torch.cat(out)
        if rays is None and prec is None:  # how the error is spread over the rays: a few broken rays (holes) or the texture everywhere?
            e = ((img - gt_) ** 2).mean(1).sort(descending=True).values
            res["eval_sqerr_top1pct_share"] = float(e[: max(e.numel() // 100, 1)].sum() / e.sum().clamp(min=1e-30))
        return psnr(img, gt_)

    res = dict(precision=precision, fineness=fin, mean_samples_per_ray=mean, steps_global=steps_global, steps_focal=steps_focal, seed=seed,
               early_T=early_T, torch_adam=bool(torch_adam), fresh_rays=bool(fresh_rays), detail=detail)
    adam = TorchAdam(hp) if torch_adam else None
    cosines = []
    loss = None
    if plug is not None:
        # the Trainer's loop through the plugin surface; the march's fineness held at the study's sample budget (UpdateRayMarch anneals 16 -> 1 over
        # 10,000 steps, PersSampler.cpp:958-967: every leg of the study holds it)
        pipe, _, nat, iteration = plug
        nat.ray_march_fineness_ = fin
        nat.UpdateRayMarch = lambda cur_step, _f=fin: setattr(nat, "ray_march_fineness_", _f)
        for step in range(steps_global + steps_focal):
            if step == steps_global:
                pipe.eval()
                res["psnr_global"] = evaluate(None)
                res["leaves_pruned"] = int((hp.nodes.view(-1, 128)[:, 96:104].contiguous().view(torch.int64) < 0).sum().item())
                pipe.train()
            loss = iteration(step)
            if progress is not None and (step + 1) % 500 == 0:
                progress(f"plugin step {step + 1}/{steps_global + steps_focal} loss {float(loss):.4f}")
        pipe.eval()
        import shutil
        shutil.rmtree(pipe._gfn_run_dir, ignore_errors=True)  # (the block table checkpoint the last iteration's callback wrote: this run's own, not the next one's start)
        if steps_focal <= 0:
            res["psnr_global"] = evaluate(None)
        res["plugin"] = dict(forward_replayed=bool(hp.last_forward_replayed), backward_replayed=bool(hp.last_backward_replayed), forward=hp.last_forward,
                             record_merge_dense=bool(cfg.record_merge_dense))
        loss = loss.detach()
    for stage, steps, lr0, block in (() if plug is not None else (("global", steps_global, cfg.lr, None), ("focal", steps_focal, cfg.lr_block, 0))):
        for i in range(steps):
            b = batch(i)
            if stage == "focal":
                block = i % n_blocks
            hold = lr_hold if stage == "global" else 0  # (the global stage may keep lr_init for its first lr_hold steps)
            mult = math.exp(math.log(lr_final) * max(i - hold, 0) / max(steps - hold, 1))  # lr_init -> lr_init * lr_final over the stage
            noise = ops.sampler_noise(n_rays, fin, train=bool(stage == "global" or focal_jitter or i == 0), generator=gen, device=dev)
            s3 = ops.s3im_index(n_rays, cfg.s3im_repeat_time, generator=gen, device=dev)
            if cosine_every > 0 and (i + 1) % cosine_every == 0:
                cosines.append((stage, i + 1, grad_cosine(hp, b, noise, s3, stage, block, legs=cosine_legs)))
            oct_now = update_oct and stage == "global"
            if adam is not None:
                loss, _ = hp.forward_backward(*b, noise, stage=stage, block=block, s3im_index=s3, update_oct=oct_now)
                adam.step([(name, lr * mult) for name, lr in hp.adam_groups(stage, block)])
            else:
                if stage == "global":
                    hp.cfg.lr = lr0 * mult
                else:
                    hp.cfg.lr_block = lr0 * mult
                loss, _ = hp.train_step(*b, noise, stage=stage, block=block, s3im_index=s3, update_oct=oct_now)
            if progress is not None and (i + 1) % 500 == 0:
                progress(f"{precision} {stage} step {i + 1}/{steps} loss {float(loss):.4f}")
        if stage == "global":
            hp.cfg.lr = lr0
            res["psnr_global"] = evaluate(None)
            res["leaves_pruned"] = int((hp.nodes.view(-1, 128)[:, 96:104].contiguous().view(torch.int64) < 0).sum().item())
            if after_global is not None:
                after_global(hp, lr0 * lr_final)
                hp.cfg.lr = lr0
        else:
            hp.cfg.lr_block = lr0
    blk = 0 if steps_focal > 0 else None
    res["psnr"] = evaluate(blk) if steps_focal > 0 else res["psnr_global"]
    # the trained state under both decoder arithmetics (inference only): what the x3 split costs once training chaos is taken out
    res["psnr_eval_x3"] = res["psnr"] if precision == "x3" else evaluate(blk, prec="x3")
    res["psnr_eval_f32"] = res["psnr"] if precision == "f32" else evaluate(blk, prec="f32")
    if held is not None:
        res["psnr_heldout"] = evaluate(blk, rays=(ho, hd, hemb, hgt))
    res["loss"] = None if loss is None else float(loss)
    if cosines:
        res["grad_cosine"] = cosines
    b0 = batch(0)
    comp, ctx = hp.forward_train(b0[0], b0[1], b0[2], ops.sampler_noise(n_rays, fin, train=False, device=dev), stage="focal" if steps_focal > 0 else "global", block=blk)
    n = int(ctx["samples"]["pts_offsets"][-1].item())
    res["alive_sample_frac_trained"] = float((comp["weights"][:n] > 0).float().mean())
    if hp.last_forward == "lm2" and hp._phase is not None:
        res["planned_K_trained"] = int(hp._phase.k.item())
    if keep_engine:
        res["engine"], res["fineness"] = hp, fin
    return res


def _paired(a, b):
    """Mean, standard error and runs of the paired differences a[i] - b[i]."""
    dl = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    se = float(dl.std(ddof=1) / math.sqrt(len(dl))) if len(dl) > 1 else None
    return float(dl.mean()), se, [round(float(v), 3) for v in dl]


def psnr_study(sc, seeds=(0, 1), legs=("fast", "reference"), progress=None, precisions=None, **kw):
    """train_and_eval per (leg of LEGS, seed), paired by seed (the same initialisation, training rays, random draws and held-out rays for
    every leg) -> dict(psnr_<leg> = mean, psnr_<leg>_runs, psnr_<leg>_std; delta / se / n_seeds = mean, standard error and count of the paired
    differences PSNR(legs[0]) - PSNR(legs[1]); decoder_delta / decoder_se = the same trained states (legs[0]'s) rendered with the x3 and with the
    exact-f32 decoder, inference only)."""
    if precisions is not None:  # rounds 2-3 spelling: ("x3", "f32")
        legs = tuple(precisions)
    out = dict(seeds=list(seeds), legs=list(legs), workload="procedural terrain seen by the synthetic capture's cameras (gfnerf_amd.procedural); PSNR on "
               f"{kw.get('eval_rays', 16384):,} held-out rays, one fixed set for every seed and leg; fresh training rays every step")
    runs = {}
    for leg in legs:
        runs[leg] = []
        for sd in seeds:
            r = train_and_eval(sc, seed=sd, **{**kw, **LEGS[leg]})
            runs[leg].append(r)
            gc.collect()  # (a run through the plugin surface leaves its pipeline in reference cycles: autograd nodes, callbacks' closures)
            torch.cuda.empty_cache()
            if progress is not None:
                progress(f"psnr {leg} seed {sd}: global {r['psnr_global']:.3f} final {r['psnr']:.3f} (same state: x3 {r['psnr_eval_x3']:.3f} f32 {r['psnr_eval_f32']:.3f})")
        vals = np.array([r["psnr"] for r in runs[leg]])
        out[f"psnr_{leg}"] = float(vals.mean())
        out[f"psnr_{leg}_runs"] = [round(float(v), 3) for v in vals]
        out[f"psnr_{leg}_std"] = float(vals.std(ddof=1)) if len(vals) > 1 else None
        out[f"psnr_global_{leg}"] = float(np.mean([r["psnr_global"] for r in runs[leg]]))
        out[f"psnr_global_{leg}_runs"] = [round(float(r["psnr_global"]), 3) for r in runs[leg]]  # (per seed: the focal stage's gain is final - global)
        if "psnr_heldout" in runs[leg][0]:
            out[f"psnr_heldout_{leg}"] = float(np.mean([r["psnr_heldout"] for r in runs[leg]]))
    for leg in legs:
        if "plugin" in runs[leg][0]:  # how the surface ran the last iterations of every seed (replayed forward / backward, the merge policy)
            out[f"{leg}_surface"] = [r["plugin"] for r in runs[leg]]
    r0 = runs[legs[0]][0]
    out.update(steps=[r0["steps_global"], r0["steps_focal"]], mean_samples_per_ray=r0["mean_samples_per_ray"], n_seeds=len(seeds),
               alive_sample_frac_trained=r0.get("alive_sample_frac_trained"), planned_K_trained=r0.get("planned_K_trained"))
    if len(legs) > 1:
        out["delta"], out["se"], out["delta_runs"] = _paired([r["psnr"] for r in runs[legs[0]]], [r["psnr"] for r in runs[legs[1]]])
        out["delta_of"] = f"PSNR({legs[0]}) - PSNR({legs[1]}), paired by seed"
    first = runs[legs[0]]
    out["decoder_delta"], out["decoder_se"], out["decoder_delta_runs"] = _paired([r["psnr_eval_x3"] for r in first], [r["psnr_eval_f32"] for r in first])
    out["decoder_delta_of"] = f"the states trained by the '{legs[0]}' leg rendered with the x3 decoder - with the exact-f32 decoder (inference only)"
    return out

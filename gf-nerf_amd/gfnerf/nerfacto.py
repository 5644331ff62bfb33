"""GFNeRFModel (gfnerf/nerfacto.py:200-747), the Model-level boundary: the reference's constructor protocol
(`config.model.setup(scene_box=, num_train_data=, metadata=, cameras=, bounds=, base_dir=)`, gf_pipeline.py:124-131),
populate_modules / get_param_groups / get_training_callbacks / get_outputs / get_metrics_dict / get_loss_dict /
get_image_metrics_and_images / load_state_dict with the reference's names, parameter groups and state-dict keys.

The engine is `HotPath` (packed samples, fused HIP kernels).  `self.field` is a real GFNeRFField whose tensors alias the
engine's storage, `self.persampler` shares its tree with the engine, and an autograd bridge (`_RenderFn`) makes the
reference Trainer's `loss.backward()` + torch.optim.Adam work unchanged: its backward runs the HIP backward and hands the
gradients to autograd -- averaged over the ranks first when the job is data parallel (the reference wraps the model in
DDP, gf_pipeline.py:136-138, which never sees the hash table: hash_3d_anchored.py:26,73-78).  (HotPath.train_step with
its fused Adam is the faster native loop; bench.py measures that one.)"""
import os
from dataclasses import dataclass, field as dc_field
from pathlib import Path
from typing import Any, Dict, List, Mapping, Tuple, Type

import numpy as np
import torch
from torch import nn

from .. import hotpath as hp_mod
from .. import ops
from ._host import Model, ModelConfig, TrainingCallback, TrainingCallbackAttributes, TrainingCallbackLocation
from .nerfacto_field import GFNeRFField
from .perssampler import PersSampler
from .rays import RayBundle


@dataclass
class GFNeRFModelConfig(ModelConfig):
    """gfnerf/nerfacto.py:72-196: the reference's fields with the reference's defaults (the method config, gfnerf/config.py,
    overrides them).  Options of other model families that GF-NeRF leaves off are carried for config-file compatibility and
    refused where they would change the result."""
    _target: Type = dc_field(default_factory=lambda: GFNeRFModel)
    near_plane: float = 0.05
    far_plane: float = 300.0
    background_color: str = "last_sample"
    patch_size: int = 32
    hidden_dim: int = 64
    hidden_dim_color: int = 64
    hidden_dim_transient: int = 64
    num_levels: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    proposal_update_every: int = 5
    proposal_warmup: int = 5000
    num_proposal_iterations: int = 2
    use_same_proposal_network: bool = False
    proposal_net_args_list: List[Dict] = dc_field(default_factory=list)
    proposal_initial_sampler: str = "piecewise"
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    orientation_loss_mult: float = 0.0001
    pred_normal_loss_mult: float = 0.001
    perceptual_loss_mult: float = 1.0
    mono_depth_loss_mult: float = 0.0
    use_proposal_weight_anneal: bool = True
    use_average_appearance_embedding: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    use_single_jitter: bool = True
    predict_normals: bool = False
    disable_scene_contraction: bool = False
    use_ch_loss: bool = False
    use_perceptual_loss: bool = False
    use_aabb_collider: bool = False
    use_mask: bool = False
    use_normal_loss: bool = False
    use_depth_loss: bool = False
    use_semantics: bool = False
    num_semantic_classes: int = 2
    semantic_loss_weight: float = 0.1
    n_blocks: int = 1
    n_split_dataset: int = 1
    steps_per_split_dataset: int = 30000
    n_activate_block: int = 3
    steps_perssampler_init: int = 10000
    use_appearance_embedding: bool = False
    scale_factor: float = 1.0
    s3im_loss_mult: float = 0.0
    s3im_kernel_size: int = 4
    s3im_stride: int = 4
    s3im_repeat_time: int = 10
    s3im_patch_height: int = 32
    # ---- this build's engine options (not in the reference)
    early_T: float = 1e-4               # early-termination transmittance of the compositing (0 = the reference: none)
    keep_blocks_resident: bool = True   # inactive block tables stay in HBM instead of being swapped through disk
    engine_seed: int = 0                # parameter initialisation of the engine (every rank uses the same)
    octree_seed: int = -1               # random streams of the host octree build (probe points of ConstructTrans); -1: 1234 + engine_seed
    wgrad_planes: str = "rne"           # HotPathConfig.wgrad_planes: operands of the decoder's weight gradient (global stage): "rne" one f16 plane each under
                                        # a per-sample dither (f64 bar 1e-4 of max |dW|), "x3" hi + lo' planes (f32-class per term), "mixed"
    record_capacity: int = 0            # HotPathConfig.record_capacity (8-byte gradient-record slots; 0: the engine's default, capped at 2048 per stream)
    render_colour_x2: bool = True       # HotPathConfig.render_colour_x2: full-image inference with the colour head's layer 2 on a one-f16 activation operand
                                        # (faster; density / accumulation / depth unchanged bit for bit, rgb within that layer's f16 rounding)
    render_tile_rays: int = 0           # rays per hipGraph-captured inference tile (get_outputs_for_camera_ray_bundle); 0 = a block of
                                        # ~128 Ki rays that divides the image (gfnerf_amd.render.choose_tile)


def focal_block(step, steps_init, steps_per_split, n_split, world=1, rank=0):
    """The focal block rank `rank` of `world` trains at `step` (None: the global stage): block (period * world + rank) mod n_split."""
    if step < 0 or (steps_init > 0 and step < steps_init):
        return None
    return (((step - steps_init) // steps_per_split) * world + rank) % n_split


class _RenderFn(torch.autograd.Function):
    """forward: HotPath.forward_train; backward: HotPath.backward (+ the data-parallel exchange).  The aliased parameters
    arrive in the order (table, block table, embedding weight, 10 decoder tensors).  A parameter that a GfnAdam steps
    (gfnerf/optim.py) gets NO autograd gradient: its gradient stays where the HIP backward wrote it (the f16 x128 table pool,
    the flat decoder / embedding vectors) and the optimiser reads it there; for any other optimiser the dense f32 gradient is
    materialised like Hash3DAnchoredFunction::backward does (Hash3DAnchored_cuda.cu:238: `pool.float() / 128`).  No host
    synchronisation either way."""

    @staticmethod
    def forward(ctx, model, stage, block, o, d, cam, noise, table, block_table, emb, *decoder):
        hot = model.hot
        s = hot.sample(o, d, noise)  # (collects the prefetched sampler, if these are its tensors)
        # static buffers (prefetch_samples): the focal stage's forward as a graph replay (global stage, measured: 2.07 against 2.02 ms per
        # iteration -- its iteration is not bound by the host)
        if stage == hp_mod.STAGE_FOCAL and any(s is st for st in model._static_samples):
            cam_s = s.get("cam_static")  # (prefetch_samples wrote the batch's camera column beside the samples: read in place by the replay)
            comp, hctx = hot.forward_train_graphed(cam_s if cam_s is not None else cam, stage, block, s, cam_static=cam_s is not None)
        else:
            # (init stage: update_oct_nodes follows every training forward, nerfacto.py:605-613, and reads every sample's alpha)
            comp, hctx = hot.forward_train(o, d, cam, noise, stage=stage, block=block, samples=s, all_alphas=stage == hp_mod.STAGE_GLOBAL)
        ctx.model, ctx.hctx, ctx.stage, ctx.block = model, hctx, stage, block
        ctx.params = (table, block_table, emb) + tuple(decoder)
        ctx.set_materialize_grads(False)  # an output the loss does not use arrives as None, not as zeros to be tested on the host
        depth = comp.get("depth")  # (a replayed forward made it inside its graph)
        depth = depth[:, 0] if depth is not None else ops.depth_finalize(comp["depth_raw"], comp["t_minmax"], hot.cfg.scale_factor)
        ctx.mark_non_differentiable(depth)
        # fresh tensor objects on the same storage: the outputs get this node as grad_fn, and the node reaches hctx -- handing out
        # hctx's own tensors would close a reference cycle through the autograd graph that no collector can see (a step's samples
        # and activations, ~1 GB at 4096 x 192, would never be freed)
        return comp["rgb"].detach(), comp["acc"].detach(), depth

    @staticmethod
    def backward(ctx, g_rgb, g_acc, _g_depth):
        hot = ctx.model.hot
        if g_rgb is None:
            g_rgb = torch.zeros_like(ctx.hctx["comp"]["rgb"])
        hot.backward_graphed(ctx.hctx, g_rgb.contiguous(), None if g_acc is None else g_acc.contiguous())  # (a replay when the forward was one)
        hot.sync_grads(ctx.stage, ctx.hctx)  # ray data parallelism: averaged over the ranks before any optimiser sees them
        table, block_table, emb = ctx.params[:3]
        decoder = ctx.params[3:]
        wants = ctx.needs_input_grad[7:]

        def dense_table_grad():
            tg = hot.table_grad
            return tg.float() / 128.0 if tg.dtype == torch.float16 else tg.clone()

        out = [None] * len(ctx.params)
        hot.grad_ready.clear()
        if ctx.stage == hp_mod.STAGE_GLOBAL:
            if wants[0]:
                if hot.fused_claims(table):
                    hot.grad_ready.add("table")
                else:
                    out[0] = dense_table_grad()
            if wants[2] and hot.cfg.use_appearance_embedding:
                if hot.fused_claims(emb):
                    hot.grad_ready.add("emb")
                else:
                    out[2] = hot.emb_grad.clone()
            if any(wants[3:]):
                if all(hot.fused_claims(p) for p in decoder):
                    hot.grad_ready.add("mlp")
                else:
                    k = 3
                    for gW, gB in hot.layer_grads:
                        out[k], out[k + 1] = (gW.clone() if wants[k] else None), (gB.clone() if wants[k + 1] else None)
                        k += 2
        elif wants[1]:
            if hot.fused_claims(block_table):
                hot.grad_ready.add(f"block{ctx.block}")
            else:
                out[1] = dense_table_grad()
        ctx.params = ctx.hctx = ctx.model = None  # the step's buffers go back to the allocator now, not when the graph dies
        return (None,) * 7 + tuple(out)


class _CharbonnierFn(torch.autograd.Function):
    """CharbonnierLoss (losses.py:73-84): sum(sqrt((x - y)^2 + eps^2)) / x.shape[0], value and gradient from one kernel."""

    @staticmethod
    def forward(ctx, pred, gt):
        loss, g = ops.charbonnier(pred.contiguous(), gt.contiguous())
        ctx.save_for_backward(g)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g_out):
        (g,) = ctx.saved_tensors
        return g * g_out, None


class _S3IMFn(torch.autograd.Function):
    """s3im_loss_mult * S3IM(image, rgb) (nerfacto.py:687-688; losses.py:713-794) on the device kernel."""

    @staticmethod
    def forward(ctx, pred, gt, index, patch_h, mult):
        loss, g = ops.s3im(gt.contiguous(), pred.contiguous(), index, patch_h=patch_h, mult=mult)
        ctx.save_for_backward(g)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g_out):
        (g,) = ctx.saved_tensors
        return g * g_out, None, None, None, None


class _TrainLossesFn(torch.autograd.Function):
    """What the pipeline runs between get_outputs and backward (get_metrics_dict's PSNR, nerfacto.py:300,624; get_loss_dict's Charbonnier rgb_loss +
    s3im_loss_mult * S3IM, :628-689; the block stage's error-map write-back, gf_pipeline.py:180-185) as TWO launches (gfn_train_losses, gfn_s3im) instead of
    ~twenty, and its backward -- d(rgb_loss) / d rgb * g_1 + d(s3im_loss) / d rgb * g_2 for whatever gradients g_1, g_2 autograd hands the two losses --
    as one (gfn_loss_grad_combine), written straight into the buffer the engine's replayed backward reads (`grad_out`, or a fresh tensor).
    -> (rgb_loss, s3im_loss | None, psnr): 0-dim views of one device vector."""

    @staticmethod
    def forward(ctx, pred, gt, s3im_index, patch_h, mult, errmap, grad_out, ws):
        pred, gt = pred.contiguous(), gt.contiguous()
        out3, g_char, g_s3 = ops.train_losses(pred, gt, want_s3im_grad=s3im_index is not None, errmap=errmap, ws=ws)
        if s3im_index is not None:
            ops.s3im(gt, pred, s3im_index, patch_h=patch_h, mult=mult, loss=out3[1:2], g_pred=g_s3)
        ctx.g, ctx.grad_out = (g_char, g_s3), grad_out
        ctx.set_materialize_grads(False)
        rgb_loss, s3_loss, psnr = out3[0], out3[1], out3[2]
        ctx.mark_non_differentiable(psnr)
        return rgb_loss, s3_loss, psnr

    @staticmethod
    def backward(ctx, go_rgb, go_s3, _go_psnr):
        g_char, g_s3 = ctx.g
        fix = lambda w: None if w is None else w.reshape(1).float().contiguous()  # noqa: E731
        if g_s3 is None:
            go_s3 = None
        g = ops.loss_grad_combine(g_char, fix(go_rgb), g_s3, fix(go_s3), out=ctx.grad_out)
        ctx.g = ctx.grad_out = None
        return g, None, None, None, None, None, None, None


def _ssim(img0, img1, data_range=1.0):
    """structural_similarity_index_measure (torchmetrics defaults: 11x11 gaussian window, sigma 1.5, k1 0.01, k2 0.03) of two
    [1,C,H,W] images; an evaluation-time metric on whole images (nerfacto.py:736), plain tensor arithmetic."""
    import torch.nn.functional as F

    C = img0.shape[1]
    k = torch.arange(11, dtype=torch.float32, device=img0.device) - 5
    g = torch.exp(-(k ** 2) / (2 * 1.5 ** 2))
    g = g / g.sum()
    win = (g[:, None] * g[None, :]).expand(C, 1, 11, 11).contiguous()
    pad = 5
    x = F.pad(img0, (pad, pad, pad, pad), mode="reflect")
    y = F.pad(img1, (pad, pad, pad, pad), mode="reflect")
    mu_x, mu_y = F.conv2d(x, win, groups=C), F.conv2d(y, win, groups=C)
    sxx = F.conv2d(x * x, win, groups=C) - mu_x ** 2
    syy = F.conv2d(y * y, win, groups=C) - mu_y ** 2
    sxy = F.conv2d(x * y, win, groups=C) - mu_x * mu_y
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    ssim = ((2 * mu_x * mu_y + c1) * (2 * sxy + c2)) / ((mu_x ** 2 + mu_y ** 2 + c1) * (sxx + syy + c2))
    return ssim[..., pad:-pad, pad:-pad].mean() if min(ssim.shape[-2:]) > 2 * pad else ssim.mean()


class GFNeRFModel(Model):
    config: GFNeRFModelConfig

    def populate_modules(self):
        """Set the fields and modules (nerfacto.py:209-301)."""
        super().populate_modules()
        c = self.config
        self.cameras = self.kwargs["cameras"]
        bounds = self.kwargs["bounds"]
        self.base_dir = self.kwargs.get("base_dir", "")
        dev = self.kwargs.get("device", None)
        if dev is None:
            cdev = self.cameras.camera_to_worlds.device
            dev = cdev if cdev.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
        dev = torch.device(dev)
        if c.predict_normals or c.use_normal_loss or c.use_semantics or c.use_perceptual_loss or c.use_mask:
            raise NotImplementedError("normals / semantics / LPIPS / masks: off in the gf-nerf method config (gfnerf/config.py:103-113), not on this path")
        self.density_fns = []
        self.persampler = PersSampler(cameras=self.cameras, bounds=bounds, bbox_levels=10, max_level=16,
                                      steps_per_split_dataset=c.steps_per_split_dataset, steps_perssampler_init=c.steps_perssampler_init,
                                      n_split_dataset=c.n_split_dataset, device=dev,
                                      seed=(1234 + c.engine_seed) if c.octree_seed < 0 else c.octree_seed)   # :223-227
        self.scale_factor = c.scale_factor
        self.sample_tmp_dir = None
        n_blocks = c.n_blocks
        step_n = max(len(self.cameras) // n_blocks, 1)
        self.block_centers = torch.stack([self.cameras.camera_to_worlds[min(i * step_n, len(self.cameras) - 1), :, 3] for i in range(n_blocks)], dim=0).to(dev)
        self.n_blocks = n_blocks
        # the engine: one tree shared with the sampler object, one set of tensors shared with the field
        nat = self.persampler.sampler
        assert c.hidden_dim == c.hidden_dim_color, "the fused decoder uses one hidden width (gfnerf/config.py:124-125: 128 / 128)"
        cfg = hp_mod.HotPathConfig(log2_T=c.log2_hashmap_size, hidden=c.hidden_dim, scale_factor=c.scale_factor, early_T=c.early_T,
                                   use_appearance_embedding=c.use_appearance_embedding, s3im_loss_mult=c.s3im_loss_mult,
                                   s3im_patch_height=c.s3im_patch_height, s3im_repeat_time=c.s3im_repeat_time, wgrad_planes=c.wgrad_planes, render_colour_x2=c.render_colour_x2,
                                   **({"record_capacity": c.record_capacity} if c.record_capacity > 0 else {}),
                                   record_merge_dense=True)  # (dense focal batches merge their record runs, decided on the device: HotPath._merge_gate)
        self.hot = hp_mod.HotPath(nat.tree_nodes_gpu_, nat.pers_trans_gpu_, n_cams=self.num_train_data, cfg=cfg, device=dev,
                                  seed=c.engine_seed, n_blocks=0, sync=self.kwargs.get("sync", None))
        if self.hot.sync.world > 1 and c.n_split_dataset % self.hot.sync.world != 0:
            raise ValueError(f"focal stage on {self.hot.sync.world} ranks: n_split_dataset = {c.n_split_dataset} must be a multiple of the number of "
                             "ranks (one block per rank and period; two ranks on one block would train diverging copies of its table)")
        self.hot.nodes, self.hot.transes, self.hot.nodes_compact = nat.tree_nodes_gpu_, nat.pers_trans_gpu_, nat.nodes_compact_
        if self.hot.sync.world > 1:  # K5's per-node maxima are MAX-reduced over the ranks before K6 (every replica keeps the same tree)
            nat.reduce_max = self.hot.sync.reduce_octree_stats
        self.field = GFNeRFField(
            None if self.scene_box is None else self.scene_box.aabb, hidden_dim=c.hidden_dim, num_levels=c.num_levels, max_res=c.max_res,
            log2_hashmap_size=c.log2_hashmap_size, hidden_dim_color=c.hidden_dim_color, hidden_dim_transient=c.hidden_dim_transient,
            spatial_distortion=None, num_images=self.num_train_data, use_pred_normals=c.predict_normals,
            use_average_appearance_embedding=c.use_average_appearance_embedding, n_blocks=self.n_blocks, block_centers=self.block_centers,
            n_active_block=c.n_activate_block, steps_perssampler_init=c.steps_perssampler_init, use_semantics=c.use_semantics,
            num_semantic_classes=c.num_semantic_classes, base_dir=self.base_dir, use_appearance_embedding=c.use_appearance_embedding,
            n_volumes=len(self.persampler.to_proposal_sampler_oct_tree_nodes()), device=dev, engine=self.hot,
            keep_blocks_resident=c.keep_blocks_resident)
        self._params_version = None
        self._tile_renderer = None
        self._trainer_step = None
        self._prefetched = None
        self._static_samples, self._static_turn = [None, None], 0  # prefetch_samples' two sets of sample buffers
        self._s3im_gen = torch.Generator(device=dev)
        self._s3im_gen.manual_seed(77 + c.engine_seed)

    # ----------------------------------------------------------------------------------------------
    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        """nerfacto.py:303-321: "fields" = the field's named parameters (decoder, embedding), "base_encoding_init" = the global
        table (the one extra entry of the overridden parameters()); block tables join through update_optimizer."""
        params = list(self.field.parameters())
        n_named = len(list(self.field.named_parameters()))
        return {"fields": params[0:n_named], "base_encoding_init": [params[n_named]]}

    def stage_of(self, step: int):
        """(stage, focal block) of a training step.  One process: the reference's schedule -- the blocks one after the other,
        steps_per_split_dataset steps each (nerfacto_field.py:248-330, perssampler.py:316-317).  W ranks: one block per rank:
        in period p = (step - init) // steps_per_split_dataset rank r trains block (p W + r) mod n_split on rays of that block's camera
        cluster (update_datamanager hands every rank's datamanager its own split), against the frozen global table and decoder every
        rank holds -- nothing is exchanged, no two ranks ever write one table, and a round over all blocks takes n_split / W periods."""
        c = self.config
        blk = focal_block(step, c.steps_perssampler_init, c.steps_per_split_dataset, c.n_split_dataset, self.hot.sync.world, self.hot.sync.rank)
        return (hp_mod.STAGE_GLOBAL, None) if blk is None else (hp_mod.STAGE_FOCAL, blk)

    def get_training_callbacks(self, training_callback_attributes: TrainingCallbackAttributes) -> List[TrainingCallback]:
        """nerfacto.py:323-520.  update_optimizer / render_init_error_maps / train_cameras_clustering / update_datamanager as the
        reference registers them, plus update_field_stage, which the reference defines (:432-436) and forgets to register --
        without it the block stage never starts."""
        c = self.config

        def update_field_stage(step, *args, **kwargs):
            self.field.cur_stage = "init_stage" if (c.steps_perssampler_init > 0 and 0 <= step < c.steps_perssampler_init) else "block_stage"
            self._trainer_step = step  # the Trainer's step as a host integer: get_outputs need not read ray_bundle.steps back

        def train_cameras_clustering(step, *args, **kwargs):
            if self.field.cur_stage == "block_stage" and self.persampler.cameras_labels is None:
                self.persampler.train_cameras_clustering_oct(kwargs["n_blocks"])

        def render_init_error_maps(step, *args, **kwargs):
            """:354-430: at the switch to the block stage render every training image at 1/8 resolution with the global-stage
            model and keep |gt - pred| summed over the channels as that image's initial error map.  The maps go to the
            datamanager (`set_error_maps`) when it keeps them in HBM, and to sample_tmp/npy like the reference otherwise."""
            if not (self.field.cur_stage == "block_stage" and self.persampler.cameras_labels is None):
                return
            dm = kwargs["pipeline"].datamanager
            if not hasattr(dm, "train_images_for_error_maps"):
                return
            self.sample_tmp_dir = str(Path(self.base_dir) / "sample_tmp")
            was_training = self.training
            self.eval()
            maps = []
            for image_idx, (gt, bundle) in enumerate(dm.train_images_for_error_maps(down_scale_factor=8, step=step)):
                pred = self.get_outputs_for_camera_ray_bundle(bundle)["rgb"]                 # [h,w,3]
                H, W = gt.shape[:2]
                up = torch.nn.functional.interpolate(pred.permute(2, 0, 1)[None], (H, W), mode="nearest")[0].permute(1, 2, 0)
                maps.append(torch.abs(gt.to(up) - up).sum(-1))
            self.train(was_training)
            if maps and hasattr(dm, "set_error_maps"):
                dm.set_error_maps(torch.stack(maps))
            else:
                os.makedirs(Path(self.sample_tmp_dir) / "npy", exist_ok=True)
                for i, m in enumerate(maps):
                    np.save(str(Path(self.sample_tmp_dir) / "npy" / f"{i:06d}.npy"), m.cpu().numpy())

        def update_datamanager(step, *args, **kwargs):
            split_idx = self.field.cur_split_dataset_idx
            dm = kwargs["pipeline"].datamanager
            if hasattr(dm, "setup_train_split_oct") and self.persampler.cameras_labels is not None and split_idx is not None:
                dm.setup_train_split_oct(self.persampler.cameras_labels, split_idx, self.sample_tmp_dir)

        def update_optimizer(step, *args, **kwargs):
            """:448-488: one optimiser per parameter group, created and dropped with the stage and the active blocks."""
            optimizers = kwargs["optimizer"]
            list_in_train, list_in_test = self.field.active_block_idxs, self.field.active_block_idxs_test
            rest = list(set(range(self.n_blocks)).difference(set(list_in_train + list_in_test)))
            if self.field.cur_stage == "init_stage":
                optimizers.add_optimizer("base_encoding_init")
                optimizers.add_optimizer("fields")
                for idx in range(self.n_blocks):
                    optimizers.delete_optimizer(f"base_encoding_{idx}")
            elif step == (c.steps_perssampler_init + c.n_split_dataset * c.steps_per_split_dataset) - 1:
                for idx in list_in_train:  # save last encoding ckpt
                    self.field.save_table(idx)
                    if not self.field.keep_blocks_resident:
                        self.field.del_table(idx)
            else:
                optimizers.delete_optimizer("base_encoding_init")
                optimizers.delete_optimizer("fields")
                for idx in list_in_train:
                    params = list(getattr(self.field, f"base_encoding_{idx}").parameters())
                    optimizers.add_optimizer(f"base_encoding_{idx}", lr_init=5e-3, params=params)
                for idx in rest:
                    optimizers.delete_optimizer(f"base_encoding_{idx}")

        after = [TrainingCallbackLocation.AFTER_TRAIN_ITERATION]
        pipe = training_callback_attributes.pipeline
        return [
            TrainingCallback(where_to_run=[TrainingCallbackLocation.BEFORE_TRAIN_ITERATION], func=update_field_stage, update_every_num_iters=1),
            TrainingCallback(where_to_run=after, func=update_optimizer, update_every_num_iters=True, args=None,
                             kwargs={"optimizer": training_callback_attributes.optimizers}),
            TrainingCallback(where_to_run=after, func=render_init_error_maps, update_every_num_iters=True, args=None, kwargs={"pipeline": pipe}),
            TrainingCallback(where_to_run=after, func=train_cameras_clustering, update_every_num_iters=True, args=None,
                             kwargs={"n_blocks": self.n_blocks}),
            TrainingCallback(where_to_run=after, func=update_datamanager, update_every_num_iters=True, args=None, kwargs={"pipeline": pipe}),
        ]

    # ----------------------------------------------------------------------------------------------
    def _sync_after_external_update(self):
        """torch.optim (or load_state_dict) wrote the aliased masters in place: the f16 shadows and split-f16 weight images the
        kernels read must follow."""
        vers = [p._version for p in self.field.parameters()]
        for i in range(self.n_blocks):
            enc = getattr(self.field, f"base_encoding_{i}", None)
            vers.append(-1 if enc is None else enc.hash_3d.feat_pool_._version)
        if vers != self._params_version:
            self.hot.refresh_shadows()
            self._params_version = vers

    def _bundle(self, ray_bundle: RayBundle):
        """(origins [R,3], directions [R,3], training step or -1).  The reference reads the step back from the device tensor
        (`steps[0][0].item()`, perssampler.py:363: a host synchronisation per step); here it comes, in this order, from the host
        integer this build's datamanagers stamp on the bundle, from the Trainer's step recorded by the update_field_stage
        callback (trainer.py:212: callbacks run with the same `step` get_train_loss_dict receives), and only then from the tensor."""
        o = ray_bundle.origins.reshape(-1, 3).contiguous().float()
        d = ray_bundle.directions.reshape(-1, 3).contiguous().float()
        stamp = getattr(ray_bundle, "_step_host", None)  # (the steps tensor it describes, its value)
        if ray_bundle.steps is None:
            step = -1
        elif stamp is not None and stamp[0] is ray_bundle.steps:
            step = int(stamp[1])
        elif self.training and self._trainer_step is not None:
            step = int(self._trainer_step)
        else:
            step = int(ray_bundle.steps.reshape(-1)[0])
        return o, d, step

    def prefetch_samples(self, ray_bundle: RayBundle):
        """Start the sampler (K3 + K4) of the NEXT training batch now, on the engine's second stream, beside this iteration's
        backward; get_outputs(ray_bundle) -- the very same object -- collects it.  Called by GFNerfPipeline.get_train_loss_dict after
        this iteration's forward, i.e. after update_ray_march / update_mode / UpdateOctNodes (nerfacto.py:598-616) left the sampler
        in the state the next GetSamples sees: same noise stream, same fineness, same tree as without the prefetch."""
        if not (self.training and self.hot._fused_ok(None)):
            return
        o, d, _ = self._bundle(ray_bundle)
        noise = self.persampler.sampler.make_noise(o.shape[0])
        self._relink_tree()
        # two sets of sample buffers, written alternately: static addresses let the engine replay the forward that reads them as a
        # hipGraph (HotPath.forward_train_graphed); a set made for another batch size is dropped
        st, k = self._static_samples, self._static_turn
        if st[k] is not None and st[k]["pts_offsets"].numel() != o.shape[0] + 1:
            st[0] = st[1] = None
            self.hot.forget_graphs()
        self.hot.prefetch_sample(o, d, noise, out=st[k])
        st[k] = self.hot._pre[4]
        cam = ray_bundle.rel_camera_indices
        if cam is not None:  # the batch's camera column beside its samples, at a fixed address: a replayed forward reads it in place (no copy on the main stream)
            with torch.cuda.stream(self.hot.side_stream()):
                cam = cam.reshape(-1)
                if st[k].get("cam_static") is None or st[k]["cam_static"].numel() != cam.numel():
                    st[k]["cam_static"] = torch.empty(cam.numel(), dtype=torch.int32, device=o.device)
                st[k]["cam_static"].copy_(cam)
        self._static_turn = 1 - k
        self._prefetched = (ray_bundle, o, d, noise)

    def get_outputs(self, ray_bundle: RayBundle):
        """nerfacto.py:522-619."""
        if self.field.persampler is None:
            object.__setattr__(self.field, "persampler", self.persampler)  # for trans_query_frame speed up (:532-533); not re-registered
        pre, self._prefetched = self._prefetched, None
        o, d, step = self._bundle(ray_bundle)
        noise = None
        if pre is not None and pre[0] is ray_bundle:  # sampled already (prefetch_samples): the same tensors collect it
            _, o, d, noise = pre
        R = o.shape[0]
        nat = self.persampler.sampler
        stage, blk = self.stage_of(step)
        if step < 0:  # eval / render: one block per image, the split of the nearest training camera (perssampler.py:365-376)
            if self.persampler.cameras_labels is not None or self.field.stage_of(step) == "block_stage":
                blk, nearest = self.persampler.split_of_origin(o[0], None)
                stage = hp_mod.STAGE_FOCAL
        cam = ray_bundle.rel_camera_indices
        if cam is None:  # test mode: the nearest training image's embedding (perssampler.py:431-433)
            _, nearest = self.persampler.split_of_origin(o[0], None)
            cam = torch.full((R,), int(nearest), dtype=torch.int32, device=o.device)
        cam = cam.reshape(-1).to(torch.int32).contiguous()
        self.field.cur_stage = "block_stage" if stage == hp_mod.STAGE_FOCAL else "init_stage"
        self.field.cur_split_dataset_idx = -1 if blk is None else blk
        if noise is None:
            noise = nat.make_noise(R)
        if self.training and torch.is_grad_enabled():
            if stage == hp_mod.STAGE_FOCAL:
                self.field.update_active_blocks(blk)                   # nerfacto_field.py:470
            self._sync_after_external_update()
            bt = getattr(self.field, f"base_encoding_{blk}").hash_3d.feat_pool_ if stage == hp_mod.STAGE_FOCAL else None
            rgb, acc, depth = _RenderFn.apply(self, stage, blk, o, d, cam, noise, self.field.base_encoding_init.hash_3d.feat_pool_, bt,
                                              self.field.embedding_appearance.weight, *self.field.decoder_parameters())
            ctx = self.hot._last_train_ctx
            oct_depth = ctx["comp"].get("oct_depth") if self.hot.last_forward_replayed else None  # (a replayed forward made it inside its graph)
            if oct_depth is None:
                oct_depth = ctx["samples"]["first_oct_dis"][:, None] / self.scale_factor  # :555
            outputs = {"rgb": rgb, "accumulation": acc[:, None], "depth": depth[:, None], "oct_depth": oct_depth}
            self.persampler.update_ray_march(step)                      # :603-604
            self.persampler.update_mode(0)
            if self.field.cur_stage == "init_stage":                    # only update oct nodes on init stage (:605-613)
                nat.last_packed = ctx["samples"]
                self.persampler.update_oct_nodes(sampled_anchors=None, pts_idx_bounds=None, sampled_weights=ctx["comp"]["weights"],
                                                 sampled_alpha=ctx["comp"]["alphas"], iter_step=step)
                self._relink_tree()  # ProcOctree (milestones / compact_freq) replaces the node blob
            else:
                self.persampler.update_mode(1)                          # :616
            return outputs
        with torch.no_grad():
            use_block = None
            if stage == hp_mod.STAGE_FOCAL and blk is not None:
                was = self.field.training
                self.field.train(False)
                try:
                    self.field.update_active_blocks(blk)                # eval: the block must be resident or have a checkpoint
                finally:
                    self.field.train(was)
                use_block = blk
            self._sync_after_external_update()
            out = self.hot.render(o, d, cam, noise, block=use_block)
        assert R == 0 or float(out["oct_depth"].min()) > 0             # :556
        return {"rgb": out["rgb"], "accumulation": out["accumulation"], "depth": out["depth"], "oct_depth": out["oct_depth"]}

    def _relink_tree(self):
        """The sampler object owns the tree; the engine samples from the same blob and 64-byte image."""
        nat = self.persampler.sampler
        if self.hot.nodes is not nat.tree_nodes_gpu_ or self.hot.nodes_compact is not nat.nodes_compact_:
            self.hot._drop_stale_samples()
            self.hot.nodes, self.hot.nodes_compact, self.hot.n_nodes = nat.tree_nodes_gpu_, nat.nodes_compact_, nat.n_nodes
            self.hot._sampler_ws = None

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> Dict[str, torch.Tensor]:
        """models/base_model.py:163-186 drives `forward` over chunks of eval_num_rays_per_chunk = 2048 rays: 1013 Python round
        trips per full-HD frame.  Same contract here ([H,W,.] outputs of every key), but the rays go through the hipGraph tile
        renderer (gfnerf_amd.render.TileRenderer: sampler -> lookup -> decoder -> compositing captured once, replayed per tile of
        config.render_tile_rays rays).  One block per image (perssampler.py:365-376)."""
        from .. import render as rmod

        H, W = camera_ray_bundle.origins.shape[:2]
        o, d, step = self._bundle(camera_ray_bundle)
        blk = None
        nearest = 0
        if step >= 0:  # a bundle stamped with a training step (render_init_error_maps, nerfacto.py:392-394): that step's stage and block
            stage, blk = self.stage_of(step)
        elif self.persampler.cameras_labels is not None or self.field.cur_stage == "block_stage":
            blk, nearest = self.persampler.split_of_origin(o[0], None)
        if blk is not None:
            was = self.field.training
            self.field.train(False)
            try:
                self.field.update_active_blocks(blk)
            finally:
                self.field.train(was)
        cam = camera_ray_bundle.rel_camera_indices
        if cam is None:
            cam = camera_ray_bundle.camera_indices
        embed_idx = int(nearest) if cam is None else int(cam.reshape(-1)[0])
        self._sync_after_external_update()
        key = (blk, float(self.persampler.sampler.ray_march_fineness_), id(self.hot.nodes_compact))
        if self._tile_renderer is None or self._tile_renderer[0] != key:
            tr = rmod.TileRenderer(self.hot, None, None, tile_rays=self.config.render_tile_rays or None, block=blk,
                                   fineness=float(self.persampler.sampler.ray_march_fineness_), use_graph=True,
                                   pipeline=1 if self.config.render_tile_rays else None)
            self._tile_renderer = (key, tr)
        out = self._tile_renderer[1].render_rays(o, d, embed_idx, H, W)
        return {k: v.view(H, W, -1) for k, v in out.items()}

    # ---- the iteration's losses and metric as one fused node (<= 15 launches outside the two graphs)
    def draw_s3im_index(self, n_rays):
        """S3IM's index vector (arange + repeat_time - 1 random permutations, losses.py:780-789) of the NEXT get_loss_dict, drawn now on the current
        stream (the pipeline calls this from its draw-ahead on the engine's second stream: a radix sort and four small launches less between the forward
        and the backward).  One draw per iteration from the model's own generator either way: the same sequence of indices."""
        c = self.config
        if c.s3im_loss_mult <= 0:
            return
        idx = ops.s3im_index(n_rays, c.s3im_repeat_time, generator=self._s3im_gen, device=self._s3im_gen.device)
        ev = torch.cuda.Event()
        ev.record()
        self._s3im_next = (int(n_rays), idx, ev)

    def _s3im_index(self, n_rays, device):
        nxt, self._s3im_next = getattr(self, "_s3im_next", None), None
        if nxt is not None and nxt[0] == n_rays:
            torch.cuda.current_stream().wait_event(nxt[2])
            nxt[1].record_stream(torch.cuda.current_stream())
            return nxt[1]
        return ops.s3im_index(n_rays, self.config.s3im_repeat_time, generator=self._s3im_gen, device=device)

    def set_error_map_sink(self, sink):
        """(error_map f32 [n,H,W(,1)], block_sums f64 | None, indices i32 [R,3], slot i32 [R] | None) of THIS iteration's error-map write-back
        (gf_pipeline.py:180-185), or None: the pipeline hands it over so that the write-back rides in the fused loss launch (get_metrics_dict / get_loss_dict)."""
        self._errmap_sink = sink

    def _fused_losses(self, outputs, batch):
        """The fused node's results for this (outputs, batch) pair -- computed by whichever of get_metrics_dict / get_loss_dict the pipeline calls first
        (:172-173), read by the other -- or None when the fast form does not apply (not training, MSE loss, odd shapes: the plain tensor code below)."""
        c = self.config
        y, image = outputs.get("rgb"), batch.get("image")
        if not (self.training and c.use_ch_loss and isinstance(y, torch.Tensor) and isinstance(image, torch.Tensor) and y.is_cuda and image.device == y.device
                and y.dtype == torch.float32 and image.dtype == torch.float32 and y.dim() == 2 and y.shape[1] == 3 and image.shape == y.shape and y.shape[0] > 0):
            return None
        cache = getattr(self, "_loss_cache", None)
        if cache is not None and cache[0] is y and cache[1] is image:
            return cache[2]
        R = y.shape[0]
        idx = None
        if c.s3im_loss_mult > 0:
            if c.s3im_kernel_size != 4 or c.s3im_stride != 4:
                raise NotImplementedError("the S3IM kernel is the method config's 4x4 window at stride 4 (gfnerf/config.py:97-98)")
            assert (R * c.s3im_repeat_time) % c.s3im_patch_height == 0, "S3IM folds repeat_time * rays into rows of s3im_patch_height"
            idx = self._s3im_index(R, y.device)
        sink, self._errmap_sink = getattr(self, "_errmap_sink", None), None
        grad_out = self.hot.grad_in_buffer(self.hot._last_train_ctx) if (y.requires_grad and self.hot.last_forward_replayed) else None
        if grad_out is not None and grad_out.shape != y.shape:
            grad_out = None
        if getattr(self, "_loss_ws", None) is None:
            self._loss_ws = ops.train_losses_workspace(y.device)  # (this model's own: the kernel leaves it zeroed)
        res = _TrainLossesFn.apply(y, image, idx, c.s3im_patch_height, c.s3im_loss_mult, sink, grad_out, self._loss_ws)
        self._loss_cache = (y, image, res)
        return res

    def get_metrics_dict(self, outputs, batch):
        fused = self._fused_losses(outputs, batch)
        if fused is not None:
            return {"psnr": fused[2]}
        image = batch["image"].to(outputs["rgb"].device)
        mse = torch.mean((outputs["rgb"].detach() - image) ** 2)
        return {"psnr": -10.0 * torch.log10(mse)}   # torchmetrics PeakSignalNoiseRatio(data_range=1.0), nerfacto.py:300,624

    def get_loss_dict(self, outputs, batch, metrics_dict=None):
        """nerfacto.py:628-689: rgb_loss (Charbonnier when use_ch_loss, losses.py:73-84, else MSE) + s3im_loss_mult * S3IM."""
        c = self.config
        fused = self._fused_losses(outputs, batch)
        if fused is not None:
            self._loss_cache = None  # (both readers served: the step's tensors are not held past the iteration)
            loss_dict = {"rgb_loss": fused[0]}
            if c.s3im_loss_mult > 0:
                loss_dict["s3im_loss"] = fused[1]
            return loss_dict
        image = batch["image"].to(outputs["rgb"].device)
        x, y = image, outputs["rgb"]
        loss_dict = {}
        if c.use_ch_loss:
            if y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.shape[1] == 3 and x.shape == y.shape and x.dtype == y.dtype:
                loss_dict["rgb_loss"] = _CharbonnierFn.apply(y, x)
            else:
                loss_dict["rgb_loss"] = torch.sum(torch.sqrt((x - y).pow(2) + 1e-6 ** 2)) / x.shape[0]
        else:
            loss_dict["rgb_loss"] = torch.mean((x - y) ** 2)
        if c.s3im_loss_mult > 0:
            R = y.shape[0]
            if c.s3im_kernel_size != 4 or c.s3im_stride != 4:
                raise NotImplementedError("the S3IM kernel is the method config's 4x4 window at stride 4 (gfnerf/config.py:97-98)")
            assert (R * c.s3im_repeat_time) % c.s3im_patch_height == 0, "S3IM folds repeat_time * rays into rows of s3im_patch_height"
            idx = ops.s3im_index(R, c.s3im_repeat_time, generator=self._s3im_gen, device=y.device)
            loss_dict["s3im_loss"] = _S3IMFn.apply(y, image, idx, c.s3im_patch_height, c.s3im_loss_mult)
        return loss_dict

    def get_image_metrics_and_images(self, outputs: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor]):
        """nerfacto.py:716-747: psnr / ssim (/ lpips where a pretrained LPIPS network is installed) of a full image + the panels."""
        image = batch["image"].to(outputs["rgb"].device)
        rgb = outputs["rgb"]
        acc = _colormap(outputs["accumulation"])
        depth = _colormap(outputs["depth"], normalize=True, weight=outputs["accumulation"])
        combined_rgb = torch.cat([image, rgb], dim=1)
        im, pr = torch.moveaxis(image, -1, 0)[None, ...], torch.moveaxis(rgb, -1, 0)[None, ...]
        psnr = -10.0 * torch.log10(torch.mean((im - pr) ** 2))
        metrics_dict = {"psnr": float(psnr.item()), "ssim": float(_ssim(im, pr))}
        try:  # the reference's LearnedPerceptualImagePatchSimilarity needs torchmetrics + downloaded VGG weights
            from torchmetrics.image.lpip import LearnedPerceptualImagePatchSimilarity

            if not hasattr(self, "_lpips"):
                object.__setattr__(self, "_lpips", LearnedPerceptualImagePatchSimilarity().to(rgb.device))
            metrics_dict["lpips"] = float(self._lpips(im, pr))
        except Exception:  # noqa: BLE001 -- not installed / no weights offline: the metric is omitted, never faked
            pass
        return metrics_dict, {"img": combined_rgb, "accumulation": acc, "depth": depth}

    # ----------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Mapping[str, Any], strict: bool = True):
        """nerfacto.py:691-714: the global table and the sampler state are loaded by hand from their own keys
        (field.base_encoding_init.{feat_pool,prime_pool,bias_pool,n_volumes}; persampler.{tree_nodes_gpu,pers_trans_gpu,
        tree_visit_cnt,milestones_ts} -- the copies under field.persampler.* are dropped), the rest goes to nn.Module."""
        state_dict = dict(state_dict)
        self.field.base_encoding_init.load_state_dict(state_dict, prefix="field.base_encoding_init")
        for k in ("tree_nodes_gpu", "pers_trans_gpu", "tree_visit_cnt", "milestones_ts"):
            state_dict.pop(f"field.persampler.{k}", None)
        self.persampler.load_state_dict(state_dict)
        for i in range(self.n_blocks):  # resident block tables travel inside the model checkpoint too
            pre = f"field.base_encoding_{i}"
            if f"{pre}.feat_pool" in state_dict:
                if not hasattr(self.field, f"base_encoding_{i}"):
                    self.field.add_table(i)
                getattr(self.field, f"base_encoding_{i}").load_state_dict(state_dict, prefix=pre)
                self.hot.block_prims[i] = getattr(self.field, f"base_encoding_{i}").hash_3d.prim_pool_
        for k in [k for k in state_dict if k.startswith("lpips.") or k.startswith("field.persampler.")]:
            del state_dict[k]   # the reference's LPIPS network weights / the second registration of the sampler's buffers
        # the global table's primes may have been replaced (LoadStates): the engine follows; the tree came with the sampler state
        nat = self.field.base_encoding_init.hash_3d
        self.hot.prim, self.hot.bias = nat.prim_pool_, nat.bias_pool_
        self._relink_tree()
        self.hot.transes = self.persampler.sampler.pers_trans_gpu_
        self.hot.weight_stats, self.hot.alpha_stats = self.persampler.sampler.tree_weight_stats_, self.persampler.sampler.tree_alpha_stats_
        self.hot.visit_cnt = self.persampler.sampler.tree_visit_cnt_
        self._params_version = None
        self._tile_renderer = None
        ret = super().load_state_dict(state_dict, strict)
        self.hot.refresh_shadows()
        return ret


def _colormap(x, normalize=False, weight=None):
    """utils/colormaps.py apply_colormap / apply_depth_colormap: [H,W,1] -> [H,W,3]; matplotlib's turbo when installed, grey otherwise."""
    x = x.float()
    if normalize:
        lo, hi = float(x.min()), float(x.max())
        x = (x - lo) / (hi - lo + 1e-10)
    x = x.clamp(0, 1)
    try:
        from matplotlib import cm

        lut = torch.tensor(cm.get_cmap("turbo")(np.linspace(0, 1, 256))[:, :3], dtype=torch.float32, device=x.device)
        out = lut[(x[..., 0] * 255).long()]
    except Exception:  # noqa: BLE001
        out = x.expand(*x.shape[:-1], 3)
    if weight is not None:
        out = out * weight + (1 - weight)
    return out

"""GFNeRFModel (gfnerf/nerfacto.py:200-747), the Model-level boundary: populate_modules builds the sampler and the
field, get_outputs(ray_bundle) -> {"rgb", "accumulation", "depth", "oct_depth"} (nerfacto.py:522-619).

The engine is `HotPath` (packed samples, fused kernels); this class gives it the reference's Model surface and an
autograd bridge, so `loss.backward()` + `torch.optim.Adam` of the reference's Trainer work unchanged: parameters
are nn.Parameters aliasing HotPath's tensors, `_RenderFn.backward` runs the HIP backward and hands the gradients to
autograd.  (HotPath.train_step with its fused Adam is the faster native loop; bench.py measures that one.)"""
from dataclasses import dataclass, field as dc_field
from typing import Dict, List

import numpy as np
import types

import torch
from torch import nn

from .. import hotpath as hp_mod
from .. import ops
from .perssampler import PersSampler
from .rays import RayBundle


@dataclass
class GFNeRFModelConfig:
    """Subset of gfnerf/nerfacto.py:72-196 that the hot path reads (defaults = gfnerf/config.py)."""
    hidden_dim: int = 128
    hidden_dim_color: int = 128
    log2_hashmap_size: int = 21
    n_blocks: int = 10
    n_split_dataset: int = 10
    steps_per_split_dataset: int = 10000
    steps_perssampler_init: int = 30000
    scale_factor: float = 10.0
    use_appearance_embedding: bool = True
    use_ch_loss: bool = True
    early_T: float = 1e-4
    resident_blocks: List[int] = dc_field(default_factory=list)  # block tables to keep in HBM (288 GB: all of them fit)


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, mlp_flat, emb, block_table, model, o, d, cam, noise, stage, block):
        hot = model.hot
        comp, hctx = hot.forward_train(o, d, cam, noise, stage=stage, block=block)
        ctx.model, ctx.hctx, ctx.stage, ctx.block = model, hctx, stage, block
        depth = ops.depth_finalize(comp["depth_raw"], comp["t_minmax"], hot.cfg.scale_factor)
        ctx.mark_non_differentiable(depth)
        return comp["rgb"], comp["acc"], depth

    @staticmethod
    def backward(ctx, g_rgb, g_acc, _g_depth):
        hot = ctx.model.hot
        g_acc = None if g_acc is None or not bool(g_acc.any()) else g_acc.contiguous()
        hot.backward(ctx.hctx, g_rgb.contiguous(), g_acc)
        tg = hot.table_grad
        tg = tg.float() / 128.0 if tg.dtype == torch.float16 else tg.clone()
        if ctx.stage == hp_mod.STAGE_GLOBAL:
            emb_g = hot.emb_grad.clone() if hot.cfg.use_appearance_embedding else None
            return tg, hot.mlp_grad.clone(), emb_g, None, None, None, None, None, None, None, None
        return None, None, None, tg, None, None, None, None, None, None, None


class GFNeRFModel(nn.Module):
    def __init__(self, config: GFNeRFModelConfig, cameras, bounds, num_train_data, device="cuda:0", base_dir="", seed=0):
        super().__init__()
        self.config = config
        self.cameras = cameras
        self.num_train_data = num_train_data
        self.device_ = torch.device(device)
        self.base_dir = base_dir
        self.scale_factor = config.scale_factor
        self.populate_modules(bounds, seed)

    def populate_modules(self, bounds, seed=0):
        c = self.config
        self.persampler = PersSampler(cameras=self.cameras, bounds=bounds, bbox_levels=10, max_level=16,
                                      steps_per_split_dataset=c.steps_per_split_dataset,
                                      steps_perssampler_init=c.steps_perssampler_init, n_split_dataset=c.n_split_dataset,
                                      device=self.device_, seed=1234 + seed)   # nerfacto.py:223-227
        nat = self.persampler.sampler
        cfg = hp_mod.HotPathConfig(log2_T=c.log2_hashmap_size, hidden=c.hidden_dim, scale_factor=c.scale_factor,
                                   early_T=c.early_T, use_appearance_embedding=c.use_appearance_embedding)
        n_res = (max(c.resident_blocks) + 1) if c.resident_blocks else 0
        self.hot = hp_mod.HotPath(nat.tree_nodes_gpu_, nat.pers_trans_gpu_, n_cams=self.num_train_data, cfg=cfg,
                                  device=self.device_, seed=seed, n_blocks=n_res)
        # the sampler object and the engine share one tree (blob + compact copy)
        self.hot.nodes, self.hot.transes, self.hot.nodes_compact = nat.tree_nodes_gpu_, nat.pers_trans_gpu_, nat.nodes_compact_
        # nn.Parameters that ALIAS the engine's tensors (what get_param_groups hands to the optimisers)
        self.feat_pool = nn.Parameter(self.hot.table, requires_grad=True)
        self.mlp_params = nn.Parameter(self.hot.mlp_flat, requires_grad=True)
        self.embedding_appearance = nn.Parameter(self.hot.emb, requires_grad=True)
        self.block_pools = nn.ParameterList([nn.Parameter(b, requires_grad=True) for b in self.hot.blocks])
        self._params_version = None
        # the reference reads model.field.cur_stage (gf_pipeline.py:180); here the field is fused into the engine
        self.field = types.SimpleNamespace(cur_stage="init_stage")

    # ----------------------------------------------------------------------------------------------
    def stage_of(self, step: int):
        c = self.config
        if step < 0 or (c.steps_perssampler_init > 0 and step < c.steps_perssampler_init):
            return hp_mod.STAGE_GLOBAL, None
        return hp_mod.STAGE_FOCAL, ((step - c.steps_perssampler_init) // c.steps_per_split_dataset) % c.n_split_dataset

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        return {"fields": [self.feat_pool, self.mlp_params, self.embedding_appearance] + list(self.block_pools)}

    def _sync_after_external_update(self):
        """torch.optim writes the aliased masters in place; the f16 shadows the lookups read must follow."""
        ver = (self.feat_pool._version, tuple(p._version for p in self.block_pools))
        if ver != self._params_version:
            self.hot.refresh_shadows()
            self._params_version = ver

    def get_outputs(self, ray_bundle: RayBundle, block=None):
        c = self.config
        o = ray_bundle.origins.contiguous().float()
        d = ray_bundle.directions.contiguous().float()
        cam = ray_bundle.rel_camera_indices if ray_bundle.rel_camera_indices is not None else ray_bundle.camera_indices
        cam = cam.reshape(-1).to(torch.int32).contiguous()
        step = -1 if ray_bundle.steps is None else int(ray_bundle.steps.reshape(-1)[0])
        stage, blk = self.stage_of(step)
        self.field.cur_stage = "block_stage" if stage == hp_mod.STAGE_FOCAL else "init_stage"
        blk = blk if block is None else block
        nat = self.persampler.sampler
        noise = nat.make_noise(o.shape[0])
        self._sync_after_external_update()
        if self.training and torch.is_grad_enabled():
            if stage == hp_mod.STAGE_FOCAL and blk >= len(self.hot.blocks):
                raise RuntimeError(f"block {blk} is not resident; list it in GFNeRFModelConfig.resident_blocks")
            bt = self.block_pools[blk] if stage == hp_mod.STAGE_FOCAL else None
            rgb, acc, depth = _RenderFn.apply(self.feat_pool, self.mlp_params, self.embedding_appearance, bt, self, o, d, cam,
                                              noise, stage, blk)
            first = self.hot_last_first_oct_dis()
            outputs = {"rgb": rgb, "accumulation": acc[:, None], "depth": depth[:, None], "oct_depth": first[:, None] / self.scale_factor}
            nat.UpdateRayMarch(step)                                    # nerfacto.py:603-604
            nat.UpdateMode(0)
            if stage == hp_mod.STAGE_GLOBAL:                           # only update oct nodes on init stage (:605-613)
                ctx = self._last_ctx()
                nat.last_packed = ctx["samples"]
                nat.UpdateOctNodes(None, None, ctx["comp"]["weights"], ctx["comp"]["alphas"], step)
                self._relink_tree()  # ProcOctree (milestones / compact_freq) replaces the node blob
            else:
                nat.UpdateMode(1)                                       # :616
            return outputs
        with torch.no_grad():
            use_block = blk if (stage == hp_mod.STAGE_FOCAL and blk is not None and blk < len(self.hot.blocks)) else None
            out = self.hot.render(o, d, cam, noise, block=use_block)
        assert float(out["oct_depth"].min()) > 0 or o.shape[0] == 0    # nerfacto.py:556
        return {"rgb": out["rgb"], "accumulation": out["accumulation"], "depth": out["depth"], "oct_depth": out["oct_depth"]}

    def _relink_tree(self):
        """The sampler object owns the tree; the engine samples from the same blob and 64-byte image."""
        nat = self.persampler.sampler
        if self.hot.nodes is not nat.tree_nodes_gpu_ or self.hot.nodes_compact is not nat.nodes_compact_:
            self.hot.nodes, self.hot.nodes_compact, self.hot.n_nodes = nat.tree_nodes_gpu_, nat.nodes_compact_, nat.n_nodes
            self.hot._sampler_ws = None

    def _last_ctx(self):
        return self.hot._last_train_ctx

    def hot_last_first_oct_dis(self):
        return self.hot._last_train_ctx["samples"]["first_oct_dis"]

    def forward(self, ray_bundle: RayBundle):
        return self.get_outputs(ray_bundle)

    def get_metrics_dict(self, outputs, batch):
        mse = torch.mean((outputs["rgb"].detach() - batch["image"].to(outputs["rgb"].device)) ** 2)
        return {"psnr": -10.0 * torch.log10(mse)}   # torchmetrics PeakSignalNoiseRatio(data_range=1.0), nerfacto.py:300,624

    def get_loss_dict(self, outputs, batch, metrics_dict=None):
        image = batch["image"].to(outputs["rgb"].device)
        x, y = image, outputs["rgb"]                 # self.rgb_loss(image, outputs["rgb"]) (nerfacto.py:636)
        if self.config.use_ch_loss:                  # CharbonnierLoss, out_norm "b" (losses.py:73-84)
            loss = torch.sum(torch.sqrt((x - y).pow(2) + 1e-6 ** 2)) / x.shape[0]
        else:
            loss = torch.mean((x - y) ** 2)
        return {"rgb_loss": loss}

"""GFNeRFField (gfnerf/nerfacto_field.py:73-603): global anchored hash encoding `base_encoding_init`, residual block
encodings `base_encoding_{i}`, `base_network`, `mlp_head`, appearance embedding, SH(4) view encoding.

Two ways to run it, on the SAME parameters:
  * piecewise, through the Field API (get_density / get_outputs, base_field.py:40-129) with autograd, built from the mirror
    modules (Hash3DAnchored, MLPNetwork); it evaluates the VALID samples only and scatters into the dense [R,1024,.] tensors
    the reference returns (padded samples have delta 0 and never reach the image);
  * fused, through the engine (gfnerf_amd.hotpath.HotPath) that GFNeRFModel.get_outputs drives: with `engine=` the field's
    tensors ARE the engine's (the global table, the decoder weights, the embedding and every resident block table alias the
    engine's storage), so optimisers, checkpoints and `update_active_blocks` act on what the fused kernels read.
The stage switch the reference computes and then overwrites (nerfacto_field.py:448-449) is honoured here
(`follow_reference_dead_switch=False`) -- SURVEY fact 0.4."""
import os
from pathlib import Path

import torch
from torch import nn

from .. import ops
from ._host import Field, FieldHeadNames
from .hash_3d_anchored import Hash3DAnchored
from .mlp import MLPNetwork
from .rays import RaySamples


class _TruncExp(torch.autograd.Function):  # nerfstudio/field_components/activations.py:24-42
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(ctx.saved_tensors[0].clamp(-15, 15))


trunc_exp = _TruncExp.apply


def get_normalized_directions(directions):
    """SH encoding perform a linear transformation to input range [0,1] (nerfacto_field.py:64-70)."""
    return (directions + 1.0) / 2.0


class Embedding(nn.Module):
    """nerfstudio/field_components/embedding.py: an nn.Embedding under `.embedding` (the key the checkpoints carry)."""

    def __init__(self, in_dim: int, out_dim: int) -> None:
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.embedding = nn.Embedding(in_dim, out_dim)

    @property
    def weight(self):
        return self.embedding.weight

    def forward(self, in_tensor):
        return self.embedding(in_tensor)


class GFNeRFField(Field):
    def __init__(self, aabb, num_images: int, num_layers: int = 2, hidden_dim: int = 64, geo_feat_dim: int = 15, num_levels: int = 16,
                 max_res: int = 2048, log2_hashmap_size: int = 19, num_layers_color: int = 3, num_layers_transient: int = 2,
                 hidden_dim_color: int = 64, hidden_dim_transient: int = 64, appearance_embedding_dim: int = 32,
                 transient_embedding_dim: int = 16, use_transient_embedding: bool = False, use_semantics: bool = False,
                 num_semantic_classes: int = 100, pass_semantic_gradients: bool = False, use_pred_normals: bool = False,
                 use_average_appearance_embedding: bool = False, use_appearance_embedding: bool = False, spatial_distortion=None,
                 n_blocks: int = 1, n_active_block: int = 3, steps_perssampler_init: int = 10000, block_centers=None, base_dir: str = "",
                 n_volumes: int = 0, device="cuda", seed=None, follow_reference_dead_switch: bool = False, engine=None,
                 keep_blocks_resident: bool = False) -> None:
        """The reference's arguments in the reference's order (nerfacto_field.py:99-129); transient / semantic / normal heads and the
        scene contraction are other model families' options that GF-NeRF's config leaves off (gfnerf/config.py:103-113) and are
        refused here rather than ignored.  device / seed / follow_reference_dead_switch / engine / keep_blocks_resident are this
        build's (keyword) extras; keep_blocks_resident leaves inactive block tables in HBM (frozen) instead of writing them to
        disk and freeing them -- the reference swaps to fit one GPU's memory, 288 GB hold all of them."""
        super().__init__()
        if use_transient_embedding or use_semantics or use_pred_normals:
            raise NotImplementedError("transient / semantic / predicted-normal heads are not on the GF-NeRF path (gfnerf/config.py:103-113)")
        dev = torch.device(device)
        self.register_buffer("aabb", None if aabb is None else torch.as_tensor(aabb, dtype=torch.float32))
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.spatial_distortion = spatial_distortion  # never applied: the sampler's warp replaces the contraction (:431)
        self.geo_feat_dim = geo_feat_dim
        self.num_images = num_images
        self.appearance_embedding_dim = appearance_embedding_dim
        self.use_average_appearance_embedding = use_average_appearance_embedding
        self.use_appearance_embedding = use_appearance_embedding
        self.embedding_appearance = Embedding(num_images, appearance_embedding_dim)
        self.persampler = None            # set by the model at its first get_outputs (nerfacto.py:532-533)
        self.cur_split_dataset_idx = None
        self.base_network = MLPNetwork(16 * 2, 1 + geo_feat_dim, {"otype": "FullyFusedMLP", "activation": "ReLU",
                                                                  "output_activation": "None", "n_neurons": hidden_dim,
                                                                  "n_hidden_layers": num_layers - 1})
        self.mlp_head = MLPNetwork(16 + geo_feat_dim + appearance_embedding_dim, 3,
                                   {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid",
                                    "n_neurons": hidden_dim_color, "n_hidden_layers": num_layers_color - 1})
        self.base_dir = base_dir
        self.encodings_ckpt_dir = Path(base_dir) / "encodings_ckpt"
        self.n_volumes = n_volumes
        self.log2_table_size = log2_hashmap_size
        self.n_blocks = n_blocks
        self.block_centers = block_centers
        self.n_active_block = n_active_block
        self.steps_perssampler_init = steps_perssampler_init
        self.single_mlp = True
        self.follow_reference_dead_switch = follow_reference_dead_switch
        self.device_ = dev
        self.engine = engine
        self.keep_blocks_resident = keep_blocks_resident
        self.base_encoding_init = Hash3DAnchored(log2_hashmap_size, n_volumes, device=dev, seed=seed, allocate=engine is None)
        if engine is None:
            self.base_encoding_init.reset()
        self.active_block_idxs = []
        self.active_block_idxs_test = []
        self.cur_stage = "init_stage"
        self.to(dev)
        if engine is not None:
            self._adopt_engine(engine)
        self.update_active_blocks(-1)  # active block initialization (:233)

    # ---- engine-backed storage --------------------------------------------------------------------
    def _adopt_engine(self, hp):
        """Every parameter becomes a view of the engine's storage: torch.optim steps, load_state_dict copies and table
        add / delete / save / load then act on exactly what the fused kernels read."""
        nat = self.base_encoding_init.hash_3d
        nat.feat_pool_ = hp.table.requires_grad_(True)
        nat.feat_pool_._gfn_link = ("table", hp)   # gfnerf.optim.GfnAdam: which gradient buffer of the engine belongs to this tensor
        nat.prim_pool_, nat.bias_pool_, nat.n_volumes_ = hp.prim, hp.bias, hp.n_volumes
        layers = list(self.base_network.layers) + list(self.mlp_head.layers)
        assert len(layers) == len(hp.layers)
        for layer, (W, B) in zip(layers, hp.layers):
            assert tuple(layer.weight.shape) == tuple(W.shape), "decoder shapes of field and engine differ"
            layer.weight = nn.Parameter(W)
            layer.bias = nn.Parameter(B)
            layer.weight._gfn_link = layer.bias._gfn_link = ("mlp", hp)
        self.embedding_appearance.embedding.weight = nn.Parameter(hp.emb)
        self.embedding_appearance.embedding.weight._gfn_link = ("emb", hp)
        for i, b in enumerate(hp.blocks):  # block tables the engine already holds
            if b is not None and not hasattr(self, f"base_encoding_{i}"):
                self.add_table(i)

    def decoder_parameters(self):
        """The 10 decoder tensors in the engine's order (base W0 b0 W1 b1, head W0 b0 W1 b1 W2 b2)."""
        out = []
        for layer in list(self.base_network.layers) + list(self.mlp_head.layers):
            out += [layer.weight, layer.bias]
        return out

    # ---- block tables (nerfacto_field.py:248-403) -------------------------------------------------
    def update_active_blocks(self, cur_split_data_idx, ray_samples=None):
        """:248-330.  Training: the active block is resident and trainable (created, and loaded from its checkpoint when one
        exists); evaluation: resident and frozen (its checkpoint must exist unless the table is resident already); every other
        resident block is saved and released."""
        active_idxs = [] if cur_split_data_idx == -1 else [int(cur_split_data_idx)]
        if self.training:
            list_in_train, list_in_test = active_idxs, self.active_block_idxs_test
        else:
            list_in_train, list_in_test = self.active_block_idxs, active_idxs
        for i in list_in_test:
            if not hasattr(self, f"base_encoding_{i}"):
                self.add_table(i)
                self.load_table(i, strict=True)
            getattr(self, f"base_encoding_{i}").set_require_grad(False)
        for i in list_in_train:
            if not hasattr(self, f"base_encoding_{i}"):
                self.add_table(i)
                self.load_table(i, strict=False)
            enc = getattr(self, f"base_encoding_{i}")
            enc.set_require_grad(True)
            enc.train()
        for i in set(range(self.n_blocks)).difference(list_in_train + list_in_test):
            if hasattr(self, f"base_encoding_{i}"):
                if self.keep_blocks_resident:
                    getattr(self, f"base_encoding_{i}").set_require_grad(False)
                else:
                    self.save_table(i)
                    self.del_table(i)
        if self.training:
            self.active_block_idxs = active_idxs
        else:
            self.active_block_idxs_test = active_idxs

    def add_table(self, table_idx):
        assert not hasattr(self, f"base_encoding_{table_idx}")
        enc = Hash3DAnchored(self.log2_table_size, self.n_volumes, device=self.device_, allocate=self.engine is None)
        if self.engine is not None:
            hp = self.engine
            if table_idx >= len(hp.blocks) or hp.blocks[table_idx] is None:
                hp.add_block(table_idx)
            nat = enc.hash_3d
            nat.feat_pool_ = hp.blocks[table_idx].requires_grad_(True)
            nat.feat_pool_._gfn_link = (f"block{table_idx}", hp)
            nat.prim_pool_, nat.bias_pool_, nat.n_volumes_ = hp.block_prims[table_idx], hp.bias, hp.n_volumes
        else:
            enc.zero()
        enc.set_require_grad(True)
        setattr(self, f"base_encoding_{table_idx}", enc)

    def del_table(self, table_idx):
        assert hasattr(self, f"base_encoding_{table_idx}")
        enc = getattr(self, f"base_encoding_{table_idx}")
        enc.unregister_hooks()
        enc.release_resources()
        delattr(self, f"base_encoding_{table_idx}")
        if self.engine is not None:
            self.engine.drop_block(table_idx)
        torch.cuda.empty_cache()

    def save_table(self, table_idx):
        assert hasattr(self, f"base_encoding_{table_idx}")
        os.makedirs(self.encodings_ckpt_dir, exist_ok=True)
        path = self.encodings_ckpt_dir / f"base_encoding_{table_idx}.ckpt"
        tmp = self.encodings_ckpt_dir / f".base_encoding_{table_idx}.ckpt.{os.getpid()}.tmp"  # written privately, published atomically
        torch.save(getattr(self, f"base_encoding_{table_idx}").state_dict(), tmp)
        os.replace(tmp, path)

    def load_table(self, table_idx, strict=False):
        assert hasattr(self, f"base_encoding_{table_idx}")
        path = self.encodings_ckpt_dir / f"base_encoding_{table_idx}.ckpt"
        if strict:
            assert os.path.exists(path), f"block table checkpoint {path} is missing"
        if os.path.exists(path):
            enc = getattr(self, f"base_encoding_{table_idx}")
            enc.load_state_dict(torch.load(path, map_location="cpu"))
            if self.engine is not None:  # LoadStates replaced the primes tensor: hand it to the engine, refresh the f16 shadow
                self.engine.block_prims[table_idx] = enc.hash_3d.prim_pool_
                self.engine.refresh_shadows()

    def memory_stats(self):
        a, c = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
        print(f"Allocated Memory: {a / 1024**3:.2f} GB")
        print(f"Cached Memory: {c / 1024**3:.2f} GB")
        return a, c

    # ---- Field API ---------------------------------------------------------------------------------
    def _valid(self, ray_samples: RaySamples):
        f2 = ray_samples.f2samples
        R, S = f2.sampled_pts.shape[:2]
        counts = (f2.pts_idx_start_end[:, 0, 1] - f2.pts_idx_start_end[:, 0, 0])
        return torch.arange(S, device=counts.device)[None, :] < counts[:, None]

    def stage_of(self, cur_step: int):
        if self.follow_reference_dead_switch or (cur_step < 0 and not (self.active_block_idxs or self.active_block_idxs_test)):
            return "init_stage"                                           # :449
        return "init_stage" if (self.steps_perssampler_init > 0 and 0 <= cur_step < self.steps_perssampler_init) else "block_stage"

    def get_density(self, ray_samples: RaySamples):
        f2 = ray_samples.f2samples
        sampled_pts = (f2.sampled_pts + 1.5) / 3.0                       # :431
        mask = self._valid(ray_samples)
        x = sampled_pts[mask].contiguous()
        anchors = f2.sampled_anchors[..., 0][mask].contiguous()
        cur_step = int(ray_samples.cur_step)
        self.cur_split_dataset_idx = ray_samples.cur_split_dataset_idx
        stage = self.cur_stage = self.stage_of(cur_step)
        if stage == "init_stage":
            self.base_encoding_init.set_require_grad(True)
            self.base_network.requires_grad_(True)
            hash_feats = self.base_encoding_init([x, anchors])
        else:
            self.base_encoding_init.set_require_grad(False)                # :460-461
            self.base_network.requires_grad_(False)
            self.update_active_blocks(ray_samples.cur_split_dataset_idx, ray_samples)
            active = self.active_block_idxs if self.training else self.active_block_idxs_test
            assert len(active) == 1                                        # :476
            hash_feats = self.base_encoding_init([x, anchors])
            hash_feats = hash_feats + getattr(self, f"base_encoding_{active[0]}")([x, anchors])  # :485-486
        h_valid = self.base_network(hash_feats)
        R, S = mask.shape
        h = torch.zeros((R, S, 1 + self.geo_feat_dim), dtype=torch.float32, device=x.device)
        h[mask] = h_valid
        density_before_activation, base_mlp_out = torch.split(h, [1, self.geo_feat_dim], dim=-1)
        density = trunc_exp(density_before_activation + 1.0) * mask[..., None]   # :499-503
        return density, base_mlp_out

    def get_outputs(self, ray_samples: RaySamples, density_embedding=None):
        f2 = ray_samples.f2samples
        mask = self._valid(ray_samples)
        R, S = mask.shape
        dirs = f2.sampled_dirs[:, 0, :].contiguous()
        d = ops.sh4(dirs)                                                  # tcnn SH(4) on (d+1)/2, :519-521
        ray_of = torch.arange(R, device=d.device)[:, None].expand(R, S)[mask]
        if self.use_appearance_embedding:
            if self.cur_stage == "block_stage":
                self.embedding_appearance.requires_grad_(False)            # :528
            emb = self.embedding_appearance(ray_samples.rel_camera_indices.reshape(-1).long())[ray_of]
        else:
            emb = torch.zeros((ray_of.numel(), self.appearance_embedding_dim), device=d.device)
        h = torch.cat([d[ray_of], density_embedding[mask], emb], dim=-1)   # :540-547
        self.mlp_head.requires_grad_(self.cur_stage != "block_stage")      # :548-551
        rgb_valid = self.mlp_head(h)
        rgb = torch.zeros((R, S, 3), dtype=torch.float32, device=d.device)
        rgb[mask] = rgb_valid
        return {FieldHeadNames.RGB: rgb}

    def parameters(self, recurse: bool = True):
        """:593-599: the registered parameters, then the global table (which the reference keeps out of named_parameters)."""
        for p in super().parameters(recurse):
            yield p
        for p in self.base_encoding_init.parameters(recurse):
            yield p

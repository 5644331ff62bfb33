"""GFNeRFField (gfnerf/nerfacto_field.py:73-603): global anchored hash encoding `base_encoding_init`, residual block
encodings `base_encoding_{i}`, `base_network`, `mlp_head`, appearance embedding, SH(4) view encoding.

Piecewise Field API (get_density / get_outputs, base_field.py:40-129) with autograd, built from the mirror modules;
it evaluates the VALID samples only and scatters into the dense [R,1024,.] tensors the reference returns (padded
samples have delta 0 and never reach the image).  The stage switch the reference computes and then overwrites
(nerfacto_field.py:448-449) is honoured here (`follow_reference_dead_switch=False`) -- SURVEY fact 0.4."""
import os
from pathlib import Path

import torch
from torch import nn

from .. import ops
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
    return (directions + 1.0) / 2.0


class GFNeRFField(nn.Module):
    def __init__(self, num_images, hidden_dim=128, geo_feat_dim=15, log2_hashmap_size=21, hidden_dim_color=128,
                 appearance_embedding_dim=32, use_appearance_embedding=True, n_blocks=1, steps_perssampler_init=30000,
                 base_dir="", n_volumes=0, device="cuda", follow_reference_dead_switch=False, seed=None):
        super().__init__()
        self.geo_feat_dim = geo_feat_dim
        self.num_images = num_images
        self.appearance_embedding_dim = appearance_embedding_dim
        self.use_appearance_embedding = use_appearance_embedding
        self.embedding_appearance = nn.Embedding(num_images, appearance_embedding_dim)
        self.base_network = MLPNetwork(16 * 2, 1 + geo_feat_dim, {"otype": "FullyFusedMLP", "activation": "ReLU",
                                                                  "output_activation": "None", "n_neurons": hidden_dim,
                                                                  "n_hidden_layers": 1})
        self.mlp_head = MLPNetwork(16 + geo_feat_dim + appearance_embedding_dim, 3,
                                   {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid",
                                    "n_neurons": hidden_dim_color, "n_hidden_layers": 2})
        self.base_dir = base_dir
        self.encodings_ckpt_dir = Path(base_dir) / "encodings_ckpt"
        self.n_volumes = n_volumes
        self.log2_table_size = log2_hashmap_size
        self.n_blocks = n_blocks
        self.steps_perssampler_init = steps_perssampler_init
        self.follow_reference_dead_switch = follow_reference_dead_switch
        self.device_ = device
        self.base_encoding_init = Hash3DAnchored(log2_hashmap_size, n_volumes, device=device, seed=seed)
        self.base_encoding_init.reset()
        self.active_block_idxs = []
        self.cur_stage = "init_stage"
        self.to(device)

    # ---- block tables (nerfacto_field.py:336-403) -------------------------------------------------
    def add_table(self, table_idx):
        assert not hasattr(self, f"base_encoding_{table_idx}")
        enc = Hash3DAnchored(self.log2_table_size, self.n_volumes, device=self.device_)
        # every table of a model shares the primes of base_encoding_init? No: the reference draws fresh primes per
        # table (Hash3DAnchored.cpp:43-50); kept.
        enc.zero()
        enc.set_require_grad(True)
        setattr(self, f"base_encoding_{table_idx}", enc)

    def del_table(self, table_idx):
        enc = getattr(self, f"base_encoding_{table_idx}")
        enc.unregister_hooks()
        enc.release_resources()
        delattr(self, f"base_encoding_{table_idx}")
        torch.cuda.empty_cache()

    def save_table(self, table_idx):
        os.makedirs(self.encodings_ckpt_dir, exist_ok=True)
        torch.save(getattr(self, f"base_encoding_{table_idx}").state_dict(), self.encodings_ckpt_dir / f"base_encoding_{table_idx}.ckpt")

    def load_table(self, table_idx, strict=False):
        path = self.encodings_ckpt_dir / f"base_encoding_{table_idx}.ckpt"
        if strict:
            assert os.path.exists(path)
        if os.path.exists(path):
            getattr(self, f"base_encoding_{table_idx}").load_state_dict(torch.load(path, map_location="cpu"))

    def update_active_blocks(self, cur_split_data_idx):
        active = [] if cur_split_data_idx == -1 else [cur_split_data_idx]
        for i in active:
            if not hasattr(self, f"base_encoding_{i}"):
                self.add_table(i)
                self.load_table(i, strict=not self.training)
            getattr(self, f"base_encoding_{i}").set_require_grad(self.training)
        for i in range(self.n_blocks):
            if i not in active and hasattr(self, f"base_encoding_{i}"):
                self.save_table(i)
                self.del_table(i)
        self.active_block_idxs = active

    # ---- Field API ---------------------------------------------------------------------------------
    def _valid(self, ray_samples: RaySamples):
        f2 = ray_samples.f2samples
        R, S = f2.sampled_pts.shape[:2]
        counts = (f2.pts_idx_start_end[:, 0, 1] - f2.pts_idx_start_end[:, 0, 0])
        return torch.arange(S, device=counts.device)[None, :] < counts[:, None]

    def get_density(self, ray_samples: RaySamples):
        f2 = ray_samples.f2samples
        sampled_pts = (f2.sampled_pts + 1.5) / 3.0                       # :431
        mask = self._valid(ray_samples)
        x = sampled_pts[mask].contiguous()
        anchors = f2.sampled_anchors[..., 0][mask].contiguous()
        cur_step = ray_samples.cur_step
        stage = "init_stage" if (self.steps_perssampler_init > 0 and 0 <= cur_step < self.steps_perssampler_init) else "block_stage"
        if self.follow_reference_dead_switch or cur_step < 0 and not self.active_block_idxs:
            stage = "init_stage"                                          # :449
        self.cur_stage = stage
        if stage == "init_stage":
            self.base_encoding_init.set_require_grad(True)
            self.base_network.requires_grad_(True)
            hash_feats = self.base_encoding_init([x, anchors])
        else:
            self.base_encoding_init.set_require_grad(False)                # :460-461
            self.base_network.requires_grad_(False)
            self.update_active_blocks(ray_samples.cur_split_dataset_idx)
            assert len(self.active_block_idxs) == 1                        # :476
            hash_feats = self.base_encoding_init([x, anchors])
            hash_feats = hash_feats + getattr(self, f"base_encoding_{self.active_block_idxs[0]}")([x, anchors])  # :485-486
        h_valid = self.base_network(hash_feats)
        R, S = mask.shape
        h = torch.zeros((R, S, 1 + self.geo_feat_dim), dtype=torch.float32, device=x.device)
        h[mask] = h_valid
        density_before_activation, base_mlp_out = torch.split(h, [1, self.geo_feat_dim], dim=-1)
        density = trunc_exp(density_before_activation + 1.0) * mask[..., None]   # :499-503
        return density, base_mlp_out

    def get_outputs(self, ray_samples: RaySamples, density_embedding):
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
        return {"rgb": rgb}

    def forward(self, ray_samples: RaySamples):
        density, emb = self.get_density(ray_samples)
        out = self.get_outputs(ray_samples, emb)
        out["density"] = density
        return out

    def parameters(self, recurse: bool = True):
        for p in super().parameters(recurse):
            yield p
        for p in self.base_encoding_init.parameters(recurse):
            yield p

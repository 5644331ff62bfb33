"""BASELINE configs[0]: the reference's pure-PyTorch CPU field path (Nerfstudio's TorchNerfactoField with the torch
HashEncoding, no tcnn, no GPU), restated -- TEST / BASELINE INFRASTRUCTURE ONLY, never imported by the product.

The reference cannot travel to the GPU box, so its CPU path is timed there through this restatement: the same sequence of
torch operations as nerfstudio/fields/nerfacto_field.py:370-461 (TorchNerfactoField), field_components/encodings.py:289-349
(HashEncoding.hash_fn / pytorch_fwd), :560-580 (SHEncoding), field_components/mlp.py:63-99, spatial_distortions.py:66-69
(SceneContraction), field_heads.py (softplus density, sigmoid rgb), cameras/rays.py:51-60,155-177 (positions, get_weights)
and model_components/renderers.py (RGB without background, accumulation, expected depth).  Pinned by `torchfield_*` in
tests/golden/reference_vectors.npz (captured by tests/golden/make_golden.py from the reference classes themselves,
parameters included); state-dict keys are the reference's.
"""
import math

import numpy as np
import torch
from torch import nn


def components_from_spherical_harmonics(levels, directions):
    """nerfstudio/utils/math.py:29-76 (levels <= 4)."""
    num = levels**2
    c = torch.zeros((*directions.shape[:-1], num), device=directions.device)
    x, y, z = directions[..., 0], directions[..., 1], directions[..., 2]
    xx, yy, zz = x**2, y**2, z**2
    c[..., 0] = 0.28209479177387814
    if levels > 1:
        c[..., 1] = 0.4886025119029199 * y
        c[..., 2] = 0.4886025119029199 * z
        c[..., 3] = 0.4886025119029199 * x
    if levels > 2:
        c[..., 4] = 1.0925484305920792 * x * y
        c[..., 5] = 1.0925484305920792 * y * z
        c[..., 6] = 0.9461746957575601 * zz - 0.31539156525251999
        c[..., 7] = 1.0925484305920792 * x * z
        c[..., 8] = 0.5462742152960396 * (xx - yy)
    if levels > 3:
        c[..., 9] = 0.5900435899266435 * y * (3 * xx - yy)
        c[..., 10] = 2.890611442640554 * x * y * z
        c[..., 11] = 0.4570457994644658 * y * (5 * zz - 1)
        c[..., 12] = 0.3731763325901154 * z * (5 * zz - 3)
        c[..., 13] = 0.4570457994644658 * x * (5 * zz - 1)
        c[..., 14] = 1.445305721320277 * z * (xx - yy)
        c[..., 15] = 0.5900435899266435 * x * (xx - 3 * yy)
    return c


class HashEncoding(nn.Module):
    def __init__(self, num_levels=16, min_res=16, max_res=1024, log2_hashmap_size=19, features_per_level=2, hash_init_scale=0.001):
        super().__init__()
        self.num_levels, self.features_per_level = num_levels, features_per_level
        self.hash_table_size = 2**log2_hashmap_size
        levels = torch.arange(num_levels)
        growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1))
        self.scalings = torch.floor(min_res * growth**levels)
        self.hash_offset = levels * self.hash_table_size
        self.hash_table = nn.Parameter((torch.rand(size=(self.hash_table_size * num_levels, features_per_level)) * 2 - 1) * hash_init_scale)

    def get_out_dim(self):
        return self.num_levels * self.features_per_level

    def hash_fn(self, t):
        t = t * torch.tensor([1, 2654435761, 805459861])
        x = torch.bitwise_xor(t[..., 0], t[..., 1])
        x = torch.bitwise_xor(x, t[..., 2])
        x %= self.hash_table_size
        x += self.hash_offset
        return x

    def forward(self, p):
        p = p[..., None, :]
        scaled = p * self.scalings.view(-1, 1)
        c = torch.ceil(scaled).type(torch.int32)
        f = torch.floor(scaled).type(torch.int32)
        off = scaled - f
        cat = lambda a, b, d: torch.cat([a[..., 0:1], b[..., 1:2], d[..., 2:3]], dim=-1)  # noqa: E731
        h0, h1, h2, h3 = self.hash_fn(c), self.hash_fn(cat(c, f, c)), self.hash_fn(cat(f, f, c)), self.hash_fn(cat(f, c, c))
        h4, h5, h6, h7 = self.hash_fn(cat(c, c, f)), self.hash_fn(cat(c, f, f)), self.hash_fn(f), self.hash_fn(cat(f, c, f))
        T = self.hash_table
        f0, f1, f2, f3, f4, f5, f6, f7 = T[h0], T[h1], T[h2], T[h3], T[h4], T[h5], T[h6], T[h7]
        ox, oy, oz = off[..., 0:1], off[..., 1:2], off[..., 2:3]
        f03 = f0 * ox + f3 * (1 - ox)
        f12 = f1 * ox + f2 * (1 - ox)
        f56 = f5 * ox + f6 * (1 - ox)
        f47 = f4 * ox + f7 * (1 - ox)
        f0312 = f03 * oy + f12 * (1 - oy)
        f4756 = f47 * oy + f56 * (1 - oy)
        return torch.flatten(f0312 * oz + f4756 * (1 - oz), start_dim=-2, end_dim=-1)


class MLP(nn.Module):
    def __init__(self, in_dim, num_layers, layer_width, out_dim=None, skip_connections=None, out_activation=None):
        super().__init__()
        out_dim = out_dim if out_dim is not None else layer_width
        self._skip = set(skip_connections) if skip_connections else set()
        layers = []
        if num_layers == 1:
            layers.append(nn.Linear(in_dim, out_dim))
        else:
            for i in range(num_layers - 1):
                if i == 0:
                    layers.append(nn.Linear(in_dim, layer_width))
                elif i in self._skip:
                    layers.append(nn.Linear(layer_width + in_dim, layer_width))
                else:
                    layers.append(nn.Linear(layer_width, layer_width))
            layers.append(nn.Linear(layer_width, out_dim))
        self.layers = nn.ModuleList(layers)
        self.out_dim, self.out_activation = out_dim, out_activation

    def forward(self, x0):
        x = x0
        for i, layer in enumerate(self.layers):
            if i in self._skip:
                x = torch.cat([x0, x], -1)
            x = layer(x)
            if i < len(self.layers) - 1:
                x = torch.relu(x)
        return self.out_activation(x) if self.out_activation is not None else x


class _Head(nn.Module):
    def __init__(self, in_dim, out_dim, act):
        super().__init__()
        self.net, self.act = nn.Linear(in_dim, out_dim), act

    def forward(self, x):
        return self.act(self.net(x))


class _Embedding(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.embedding = nn.Embedding(n, d)

    def forward(self, i):
        return self.embedding(i)


class TorchNerfactoFieldPort(nn.Module):
    def __init__(self, num_images, log2_hashmap_size=19, appearance_embedding_dim=40):
        super().__init__()
        self.aabb = nn.Parameter(torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), requires_grad=False)
        self.appearance_embedding_dim = appearance_embedding_dim
        self.embedding_appearance = _Embedding(num_images, appearance_embedding_dim)
        self.position_encoding = HashEncoding(log2_hashmap_size=log2_hashmap_size)
        self.mlp_base = MLP(32, 3, 64, skip_connections=(4,), out_activation=nn.ReLU())
        self.mlp_head = MLP(64 + 16 + appearance_embedding_dim, 2, 32, out_activation=nn.ReLU())
        self.field_output_density = _Head(64, 1, nn.Softplus())
        self.field_heads = nn.ModuleList([_Head(32, 3, nn.Sigmoid())])

    def forward(self, origins, directions, starts, ends, camera_indices):
        """origins / directions [R,3], starts / ends [R,S,1], camera_indices [R] -> (density [R,S,1], rgb [R,S,3]); the reference
        evaluates the field on the FLATTENED samples (its get_outputs only works that way)."""
        R, S = starts.shape[:2]
        pos = origins[:, None, :] + directions[:, None, :] * (starts + ends) / 2          # Frustums.get_positions
        pos = pos.reshape(R * S, 3)
        mag = torch.linalg.norm(pos, ord=None, dim=-1)[..., None]                          # SceneContraction
        pos = torch.where(mag < 1, pos, (2 - (1 / mag)) * (pos / mag))
        base = self.mlp_base(self.position_encoding(pos))
        density = self.field_output_density(base)
        cam = camera_indices[:, None].expand(R, S).reshape(R * S)
        if self.training:
            emb = self.embedding_appearance(cam)
        else:
            emb = torch.zeros((R * S, self.appearance_embedding_dim))
        dirs = directions[:, None, :].expand(R, S, 3).reshape(R * S, 3)
        with torch.no_grad():
            enc_dir = components_from_spherical_harmonics(4, dirs)
        out = self.mlp_head(torch.cat([enc_dir, base, emb.view(-1, self.appearance_embedding_dim)], dim=-1))
        rgb = self.field_heads[0](out)
        return density.view(R, S, 1), rgb.view(R, S, 3)


def render(field, origins, directions, starts, ends, camera_indices):
    """Field + RaySamples.get_weights + RGB (no background term, renderers.py:109-110) / accumulation / expected depth."""
    density, rgb = field(origins, directions, starts, ends, camera_indices)
    deltas = ends - starts
    dd = deltas * density
    alphas = 1 - torch.exp(-dd)
    tr = torch.cumsum(dd[..., :-1, :], dim=-2)
    tr = torch.cat([torch.zeros((*tr.shape[:1], 1, 1)), tr], dim=-2)
    w = torch.nan_to_num(alphas * torch.exp(-tr))
    comp_rgb = torch.sum(w * rgb, dim=-2)
    acc = torch.sum(w, dim=-2)
    steps = (starts + ends) / 2
    depth = torch.sum(w * steps, dim=-2) / (torch.sum(w, -2) + 1e-10)
    depth = torch.clip(depth, steps.min(), steps.max())
    return dict(density=density, rgb_samples=rgb, weights=w, rgb=comp_rgb, accumulation=acc, depth=depth)


def lego_batch(n_rays=1024, n_samples=48, n_cams=100, seed=0):
    """rays from 100 cameras on a radius-4 hemisphere looking at the origin, 48 uniform samples in [2, 6]."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n_cams, generator=g)
    phi = 2 * math.pi * torch.rand(n_cams, generator=g)
    z = u
    r = torch.sqrt(1 - z * z)
    cam_pos = 4.0 * torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], -1)
    cam = torch.randint(0, n_cams, (n_rays,), generator=g)
    target = (torch.rand((n_rays, 3), generator=g) - 0.5)  # points of the unit cube the pixels look at
    o = cam_pos[cam]
    d = torch.nn.functional.normalize(target - o, dim=-1)
    bins = torch.linspace(2.0, 6.0, n_samples + 1)[None].expand(n_rays, n_samples + 1)
    return o, d, bins[:, :-1, None].contiguous(), bins[:, 1:, None].contiguous(), cam


def time_config0(n_rays=1024, n_samples=48, warmup=5, iters=20, log2_hashmap_size=19, seed=0):
    """-> dict(fwd_ms, fwd_bwd_ms, threads): forward + compositing, and forward + compositing + backward of an MSE loss."""
    import time

    torch.manual_seed(seed)
    field = TorchNerfactoFieldPort(100, log2_hashmap_size=log2_hashmap_size)
    field.train()
    o, d, s, e, cam = lego_batch(n_rays, n_samples, seed=seed)
    gt = torch.rand(n_rays, 3)

    def fwd():
        with torch.no_grad():
            return render(field, o, d, s, e, cam)["rgb"]

    def fwd_bwd():
        field.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(render(field, o, d, s, e, cam)["rgb"], gt)
        loss.backward()

    res = {}
    for name, fn in (("fwd_ms", fwd), ("fwd_bwd_ms", fwd_bwd)):
        for _ in range(warmup):
            fn()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        res[name] = (time.perf_counter() - t0) / iters * 1e3
    res["threads"] = torch.get_num_threads()
    return res

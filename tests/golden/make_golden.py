"""Generate golden vectors by IMPORTING the reference's pure-PyTorch pieces.

Run only in the build container (needs /root/reference, which never travels to the
GPU box).  The outputs (*.npz next to this file) are committed; tests read only
those.  Annotation-only third-party packages the reference imports but this image
lacks (torchtyping, jaxtyping, nerfacc, wandb, tensorboard) are replaced IN
PROCESS by empty stand-ins -- they carry type annotations / logging only and take
no part in the arithmetic captured here.

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Sub(torch.Tensor):
        def __class_getitem__(cls, item):
            return cls

    class _CT:
        AABB = 0
        UN_BOUNDED_SPHERE = 1
        UN_BOUNDED_TANH = 2

    stub("torchtyping", TensorType=_Sub, patch_typeguard=lambda *a, **k: None)
    stub("jaxtyping", Float=_Sub, Int=_Sub, Shaped=_Sub, Bool=_Sub)
    stub("nerfacc", ContractionType=_CT)
    stub("wandb")
    stub("torch.utils.tensorboard", SummaryWriter=object)
    # cameras.py imports cv2 / torchvision for Cameras.to_json (JPEG thumbnails) only; generate_rays never touches them
    # (data_utils.py names cv2.INTER_NEAREST as a default argument of its depth-image loader, never called here)
    for name in ("cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        stub(name, INTER_NEAREST=0)


def gen_mlp(out):
    """a8: gfnerf/mlp.py MLPNetwork fwd + grads for base_network and mlp_head shapes
    (nerfacto_field.py:174-179, 217-227 with hidden_dim=128 from config.py:124-125)."""
    from gfnerf.mlp import MLPNetwork

    for tag, n_in, n_out, n_hidden, out_act, seed in (
        ("base", 32, 16, 1, "None", 4),
        ("head", 63, 3, 2, "Sigmoid", 5),
    ):
        torch.manual_seed(seed)
        net = MLPNetwork(n_in, n_out, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": out_act,
                                       "n_neurons": 128, "n_hidden_layers": n_hidden})
        x = (torch.rand(257, n_in) * 2 - 1).requires_grad_(True)
        y = net(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        out[f"mlp_{tag}_x"] = x.detach().numpy()
        out[f"mlp_{tag}_y"] = y.detach().numpy()
        out[f"mlp_{tag}_gy"] = gy.numpy()
        out[f"mlp_{tag}_gx"] = x.grad.numpy()
        for i, layer in enumerate(net.layers):
            out[f"mlp_{tag}_W{i}"] = layer.weight.detach().numpy()
            out[f"mlp_{tag}_b{i}"] = layer.bias.detach().numpy()
            out[f"mlp_{tag}_gW{i}"] = layer.weight.grad.numpy()
            out[f"mlp_{tag}_gb{i}"] = layer.bias.grad.numpy()


def gen_composite(out):
    """a12-a13: RaySamples.get_weights_f2nerf (rays.py:178-200) + RGB / Accumulation /
    Depth('expected') renderers (renderers.py) on a dense [R,S] batch whose tails are
    zero-padded exactly like PersSampler's dense output (PersSampler_cuda.cu:437-444)."""
    from nerfstudio.cameras.rays import Frustums, RaySamples
    from nerfstudio.model_components.renderers import AccumulationRenderer, DepthRenderer, RGBRenderer

    torch.manual_seed(11)
    R, S = 37, 24
    counts = torch.randint(0, S + 1, (R,))
    counts[0] = 0
    counts[1] = S
    valid = (torch.arange(S)[None, :] < counts[:, None]).float()[..., None]
    sigma = torch.exp(torch.randn(R, S, 1) * 2.0) * valid
    sigma[2, 3, 0] = float("inf")  # alpha = 1 there
    sigma[3, 1, 0] = 1e30
    delta = (torch.rand(R, S, 1) * 0.05 + 1e-3) * valid  # zero padding -> alpha 0
    t = (torch.cumsum(torch.rand(R, S, 1) * 0.1 + 0.01, dim=1) + 0.3) * valid
    rgb = torch.rand(R, S, 3)
    rgb[4, 0, 1] = float("nan")  # hits nan_to_num only in eval mode
    sigma.requires_grad_(True)
    rgb_g = rgb.clone().requires_grad_(True)

    fr = Frustums(origins=torch.zeros(R, S, 3), directions=torch.ones(R, S, 3), starts=t, ends=t,
                  pixel_area=torch.ones(R, S, 1))
    rs = RaySamples(frustums=fr, deltas=delta)
    w, a, T = rs.get_weights_f2nerf(sigma)
    ren = RGBRenderer(background_color="black")
    ren.train()
    rgb_train = ren(rgb=rgb_g, weights=w)
    acc = AccumulationRenderer()(weights=w)
    depth = DepthRenderer(method="expected")(weights=w, ray_samples=rs)
    # gradient of a scalar through rgb (train mode) and accumulation; rows with NaN rgb are excluded
    g_rgb = torch.randn(R, 3)
    g_rgb[4] = 0.0
    g_acc = torch.randn(R, 1)
    (rgb_train * g_rgb).nansum().add((acc * g_acc).sum()).backward()
    ren.eval()
    with torch.no_grad():
        rgb_eval = ren(rgb=rgb.clone(), weights=w.detach())

    out["cmp_counts"] = counts.numpy()
    out["cmp_sigma"] = sigma.detach().numpy()
    out["cmp_delta"] = delta.numpy()
    out["cmp_t"] = t.numpy()
    out["cmp_rgb"] = rgb.numpy()
    out["cmp_weights"] = w.detach().numpy()
    out["cmp_alphas"] = a.detach().numpy()
    out["cmp_trans"] = T.detach().numpy()
    out["cmp_rgb_train"] = rgb_train.detach().numpy()
    out["cmp_rgb_eval"] = rgb_eval.numpy()
    out["cmp_acc"] = acc.detach().numpy()
    out["cmp_depth"] = depth.detach().numpy()
    out["cmp_g_rgb"] = g_rgb.numpy()
    out["cmp_g_acc"] = g_acc.numpy()
    out["cmp_d_sigma"] = sigma.grad.numpy()
    out["cmp_d_rgb"] = rgb_g.grad.numpy()


def gen_activations(out):
    """a11: trunc_exp fwd/bwd (activations.py:24-42) incl. the +-15 clamp."""
    from nerfstudio.field_components.activations import trunc_exp

    x = torch.tensor([-20.0, -15.0, -3.0, 0.0, 1.0, 2.5, 15.0, 20.0, 30.0], requires_grad=True)
    y = trunc_exp(x)
    y.backward(torch.ones_like(y))
    out["texp_x"] = x.detach().numpy()
    out["texp_y"] = y.detach().numpy()
    out["texp_gx"] = x.grad.numpy()


def gen_sh(out):
    """SH degree 4, nerfstudio convention (utils/math.py:27-87).  tcnn's own kernel is
    third-party and absent; it differs from this by the sign of the odd-m terms."""
    from nerfstudio.utils.math import components_from_spherical_harmonics

    torch.manual_seed(3)
    d = torch.randn(200, 3)
    d = d / d.norm(dim=-1, keepdim=True)
    out["sh_dirs"] = d.numpy()
    out["sh_ns"] = components_from_spherical_harmonics(levels=4, directions=d).numpy()


def gen_losses(out):
    """Charbonnier (losses.py:73-84) as called by GFNeRFModel.get_loss_dict (nerfacto.py:636)."""
    from nerfstudio.model_components.losses import CharbonnierLoss

    torch.manual_seed(6)
    pred = torch.rand(64, 3, requires_grad=True)
    gt = torch.rand(64, 3)
    loss = CharbonnierLoss()(gt, pred)
    loss.backward()
    out["charb_pred"] = pred.detach().numpy()
    out["charb_gt"] = gt.numpy()
    out["charb_loss"] = np.array(loss.item(), np.float64)
    out["charb_gpred"] = pred.grad.numpy()


def gen_adam(out):
    """torch.optim.Adam(lr=1e-2, eps=1e-15) -- the optimiser config.py:130 hands the table to."""
    torch.manual_seed(8)
    p = torch.randn(300, requires_grad=True)
    opt = torch.optim.Adam([p], lr=1e-2, eps=1e-15)
    out["adam_p0"] = p.detach().numpy().copy()
    gs = []
    for _ in range(3):
        g = torch.randn(300) * torch.tensor(10.0).pow(torch.randint(-6, 1, (300,)).float())
        gs.append(g.numpy())
        p.grad = g.clone()
        opt.step()
    out["adam_g"] = np.stack(gs)
    out["adam_p3"] = p.detach().numpy()


def gen_rays(out):
    """Cameras.generate_rays (cameras.py:560-734, perspective, no distortion) incl. GF's
    lookat_directions (:704) -- random poses, explicit (camera, pixel) pairs like the pixel sampler produces and a
    row of full-image coordinates (get_image_coords, pixel_offset 0.5) like render.py:109 uses."""
    from nerfstudio.cameras.cameras import Cameras, CameraType

    rng = np.random.default_rng(21)
    n_cams = 5
    c2w = np.zeros((n_cams, 3, 4), np.float32)
    for i in range(n_cams):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        c2w[i, :, :3] = q * np.sign(np.linalg.det(q))
        c2w[i, :, 3] = rng.uniform(-40, 40, 3)
    fx = rng.uniform(800, 1200, n_cams).astype(np.float32); fy = rng.uniform(800, 1200, n_cams).astype(np.float32)
    cx = rng.uniform(500, 650, n_cams).astype(np.float32); cy = rng.uniform(380, 480, n_cams).astype(np.float32)
    cams = Cameras(camera_to_worlds=torch.from_numpy(c2w), fx=torch.from_numpy(fx), fy=torch.from_numpy(fy), cx=torch.from_numpy(cx),
                   cy=torch.from_numpy(cy), width=1152, height=864, camera_type=CameraType.PERSPECTIVE)
    n = 600
    cam_idx = rng.integers(0, n_cams, n)
    coords = np.stack([rng.integers(0, 864, n) + 0.5, rng.integers(0, 1152, n) + 0.5], -1).astype(np.float32)  # (y, x)
    coords[:40] = np.stack([rng.uniform(0, 864, 40), rng.uniform(0, 1152, 40)], -1)  # sub-pixel positions too
    rb = cams.generate_rays(camera_indices=torch.from_numpy(cam_idx)[:, None], coords=torch.from_numpy(coords))
    out["rays_c2w"] = c2w; out["rays_intr"] = np.stack([fx, fy, cx, cy], -1)
    out["rays_cam_idx"] = cam_idx.astype(np.int32); out["rays_coords"] = coords
    out["rays_origins"] = rb.origins.numpy(); out["rays_directions"] = rb.directions.numpy()
    out["rays_pixel_area"] = rb.pixel_area.numpy(); out["rays_lookat"] = rb.lookat_directions.numpy()
    out["rays_dir_norm"] = rb.metadata["directions_norm"].numpy()
    # one image row through the camera-level entry point render.py uses
    full = cams.generate_rays(camera_indices=2)
    out["rays_row_directions"] = full.directions[431].numpy()  # [W,3], row y = 431
    out["rays_row_pixel_area"] = full.pixel_area[431].numpy()


def gen_scheduler(out):
    """GFNerfExponentialDecayScheduler (engine/schedulers.py:113-184) LR multiplier curve as configured
    by gfnerf/config.py:135 (lr_final 1e-4, max_steps = steps_perssampler_init), through torch's LambdaLR."""
    from nerfstudio.engine.schedulers import GFNerfExponentialDecayScheduler, GFNerfExponentialDecaySchedulerConfig

    cfgs = dict(default=dict(lr_final=1e-4, max_steps=300, n_split_dataset=4, steps_per_split_dataset=100, steps_perssampler_init=300),
                warm=dict(lr_final=1e-4, max_steps=200, warmup_steps=40, n_split_dataset=3, steps_per_split_dataset=80,
                          steps_perssampler_init=150))
    for tag, kw in cfgs.items():
        conf = GFNerfExponentialDecaySchedulerConfig(**kw)
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=1e-2)
        sched = GFNerfExponentialDecayScheduler(conf).get_scheduler(opt, 1e-2)
        lrs = []
        for _ in range(900):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step(); sched.step()
        out[f"sched_{tag}_lr"] = np.asarray(lrs, np.float64)
        out[f"sched_{tag}_cfg"] = np.asarray([kw.get("lr_final"), kw.get("max_steps"), kw.get("warmup_steps", 0), kw["n_split_dataset"],
                                              kw["steps_per_split_dataset"], kw["steps_perssampler_init"]], np.float64)


def gen_s3im(out):
    """S3IM (losses.py:713-794) as configured by gfnerf/config.py:96-100 (kernel 4, stride 4, 10 repeats,
    virtual patch height 32) and called by get_loss_dict (nerfacto.py:688): s3im_loss(image, outputs["rgb"]).
    The 9 random permutations are re-drawn from the same seed so the fixture carries the index vector."""
    from nerfstudio.model_components.losses import S3IM

    R = 256
    torch.manual_seed(31)
    image = torch.rand(R, 3)
    pred = (image + 0.1 * torch.randn(R, 3)).clamp(0, 1).requires_grad_(True)
    crit = S3IM(s3im_kernel_size=4, s3im_stride=4, s3im_repeat_time=10, s3im_patch_height=32)
    torch.manual_seed(77)
    loss = crit(image, pred)
    loss.backward()
    torch.manual_seed(77)
    idx = torch.cat([torch.arange(R)] + [torch.randperm(R) for _ in range(9)])
    out["s3im_image"] = image.numpy(); out["s3im_pred"] = pred.detach().numpy(); out["s3im_index"] = idx.numpy().astype(np.int32)
    out["s3im_loss"] = np.asarray(loss.item(), np.float64); out["s3im_gpred"] = pred.grad.numpy()


def gen_pixel_samplers(out):
    """SURVEY 8(f) row 3, batch assembly: PixelSampler / ErrorPixelSampler (nerfstudio/data/pixel_samplers.py:26-209,594-844)
    and CacheDataloader._update_error_map (nerfstudio/data/utils/dataloaders.py:140-142) fed the error of
    gfnerf/gf_pipeline.py:181-184.  The uniform sampler's torch.rand draws are re-drawn from the same seed so the fixture
    carries them; the weighted sampler is captured as inclusion counts over many seeded calls (its random streams --
    torch.multinomial + Python's random.sample -- cannot be injected)."""
    import random

    from nerfstudio.data.pixel_samplers import ErrorPixelSampler, PixelSampler
    from nerfstudio.data.utils.dataloaders import CacheDataloader

    g = torch.Generator().manual_seed(5)
    n, H, W, B = 5, 12, 17, 96
    batch = dict(image=torch.rand(n, H, W, 3, generator=g), image_idx=torch.tensor([7, 3, 9, 0, 4]), rel_camera_idx=torch.tensor([2, 1, 4, 0, 3]))
    ps = PixelSampler(B)
    torch.manual_seed(123)
    got = ps.sample(batch)
    torch.manual_seed(123)
    u = torch.rand((B, 3))
    out["pix_image"] = batch["image"].numpy(); out["pix_image_idx"] = batch["image_idx"].numpy().astype(np.int32)
    out["pix_rel_idx"] = batch["rel_camera_idx"].numpy().astype(np.int32); out["pix_u"] = u.numpy()
    out["pix_indices"] = got["indices"].numpy().astype(np.int32); out["pix_gt"] = got["image"].numpy()
    out["pix_rel"] = got["rel_camera_indices"].numpy().astype(np.int32)
    # the same arithmetic at Rubble-scale extents (f32 rounding of u * extent)
    torch.manual_seed(321)
    big = ps.sample_method(4096, 1600, 864, 1152)
    torch.manual_seed(321)
    out["pix_big_u"] = torch.rand((4096, 3)).numpy(); out["pix_big_indices"] = big.numpy().astype(np.int32)

    # weighted sampler: 2048 pixels, four heavy ones (without replacement shows as saturation), a zero-weight region
    n, H, W, B = 4, 16, 32, 40
    em = torch.rand(n, H, W, 1, generator=g) * 0.05
    em[1, :, 16:] = 0.0
    for (c, y, x) in ((0, 3, 5), (2, 9, 30), (3, 0, 0), (3, 15, 31)):
        em[c, y, x, 0] = 20.0
    eps = ErrorPixelSampler(B)
    calls = 1500
    cnt_w = np.zeros(n * H * W, np.int64); cnt_u = np.zeros(n * H * W, np.int64)
    k_w = int(B * eps.weighted_choice_ratio)
    for k in range(calls):
        torch.manual_seed(1000 + k); random.seed(1000 + k)
        idx = eps.sample_method(B, n, H, W, error_map=em, device="cpu")
        lin = ((idx[:, 0] * H + idx[:, 1]) * W + idx[:, 2]).numpy()
        assert len(set(lin[:k_w])) == k_w and len(set(lin[k_w:])) == B - k_w
        np.add.at(cnt_w, lin[:k_w], 1); np.add.at(cnt_u, lin[k_w:], 1)
    out["errpix_map"] = em.numpy(); out["errpix_cfg"] = np.asarray([B, k_w, calls], np.int64)
    out["errpix_count_weighted"] = cnt_w; out["errpix_count_uniform"] = cnt_u

    # error-map write
    class _Loader:
        cached_collated_batch = {"error_map": em.clone()}
    R = 64
    ray_idx = torch.stack([torch.randint(0, n, (R,), generator=g), torch.randint(0, H, (R,), generator=g), torch.randint(0, W, (R,), generator=g)], 1)
    pred, gt = torch.rand(R, 3, generator=g), torch.rand(R, 3, generator=g)
    error = torch.abs(gt - pred)          # gf_pipeline.py:183
    error = torch.sum(error, dim=-1)      # :184
    CacheDataloader._update_error_map(_Loader, ray_idx, error)
    out["errupd_indices"] = ray_idx.numpy().astype(np.int32); out["errupd_pred"] = pred.numpy(); out["errupd_gt"] = gt.numpy()
    out["errupd_map_after"] = _Loader.cached_collated_batch["error_map"].numpy()


def gen_torch_field(out):
    """BASELINE configs[0] / SURVEY 8(c) item 5: the reference's pure-PyTorch field (TorchNerfactoField with the torch
    HashEncoding) + get_weights + renderers on a small seeded batch, parameters included (a 2^12-entry table keeps the
    fixture small), training mode (appearance embedding live), field evaluated on the flattened samples."""
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.field_components.encodings import HashEncoding, SHEncoding
    from nerfstudio.field_components.field_heads import FieldHeadNames
    from nerfstudio.fields.nerfacto_field import TorchNerfactoField
    from nerfstudio.model_components.renderers import AccumulationRenderer, DepthRenderer, RGBRenderer

    torch.manual_seed(11)
    n_img, R, S = 7, 24, 12
    f = TorchNerfactoField(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=n_img,
                           position_encoding=HashEncoding(log2_hashmap_size=12, implementation="torch"), direction_encoding=SHEncoding())
    with torch.no_grad():
        f.position_encoding.hash_table.mul_(300.0)  # the 1e-3 init gives an almost constant field
    f.train()
    o = torch.randn(R, 3) * 1.5
    d = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1)
    cam = torch.randint(0, n_img, (R, 1))
    rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones(R, 1), lookat_directions=d.clone(), camera_indices=cam)
    bins = torch.linspace(0.2, 4.0, S + 1)[None].expand(R, S + 1)
    rs = rb.get_ray_samples(bin_starts=bins[:, :-1, None], bin_ends=bins[:, 1:, None])
    fo = f(rs.flatten())
    density = fo[FieldHeadNames.DENSITY].view(R, S, 1)
    rgb = fo[FieldHeadNames.RGB].view(R, S, 3)
    w = rs.get_weights(density)
    out["torchfield_origins"] = o.numpy(); out["torchfield_directions"] = d.numpy(); out["torchfield_cam"] = cam[:, 0].numpy().astype(np.int64)
    out["torchfield_starts"] = bins[:, :-1, None].numpy().copy(); out["torchfield_ends"] = bins[:, 1:, None].numpy().copy()
    for k, v in f.state_dict().items():
        out["torchfield_param/" + k] = v.detach().numpy()
    out["torchfield_density"] = density.detach().numpy(); out["torchfield_rgb_samples"] = rgb.detach().numpy()
    out["torchfield_weights"] = w.detach().numpy()
    out["torchfield_rgb"] = RGBRenderer(background_color="random")(rgb=rgb, weights=w).detach().numpy()
    out["torchfield_acc"] = AccumulationRenderer()(weights=w).detach().numpy()
    out["torchfield_depth"] = DepthRenderer(method="expected")(weights=w, ray_samples=rs).detach().numpy()


def _aerial_cameras(n_side, seed, extent=40.0, height=12.0):
    """A small serpentine aerial capture (the same kind of scene bench.py uses): c2w [n,3,4] f32, OpenGL convention."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-extent, extent, n_side)
    c2w = []
    for iy, y in enumerate(xs):
        for x in (xs if iy % 2 == 0 else xs[::-1]):
            pos = np.array([x, y, height]) + rng.normal(0, 0.3, 3)
            a, b, c = rng.uniform(0, 2 * np.pi), np.radians(rng.uniform(-15, 15)), np.radians(rng.uniform(-15, 15))
            rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
            rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
            ry = np.array([[np.cos(c), 0, np.sin(c)], [0, 1, 0], [-np.sin(c), 0, np.cos(c)]])
            c2w.append(np.concatenate([rz @ rx @ ry, pos[:, None]], axis=1))
    return np.asarray(c2w, np.float32)


def gen_persoctree(out):
    """a3 warp + f1 octree build, pinned from the reference's pure-PyTorch prototype gfnerf/persoctree.py ("useful as a
    readable restatement of ConstructTrans"): distance_summary (:234-246), PCA (:183-194), get_visi_cams (:248-290),
    construct_trans (:86-181) on three camera sets with the random first pick re-drawn from the same seed, and -- captured from
    INSIDE that call by wrapping self.PCA / torch.linalg.inv for its duration -- the projected values transed_pts [n,12] (:155),
    PCA's (L, V) and the warp Jacobian jac = V . dv_dxyz [n,3,3] (:160-168) at the probe points; plus the warp of a 32^3 grid
    exactly as vis_transed_pts computes and prints it (:334-371; its .obj output is parsed back).
    The prototype re-aims the six virtual cameras with a look-at (utils.adjust_cameras_lookat, up = +z) where the C++ build
    uses the minimal rotation (Eigen::AngleAxisf, PersSampler.cpp:718-746): the z rows / offsets of w2xz, dis_summary, the camera
    selection, PCA and the step normalisation are common to both and are what the tests compare."""
    import tempfile

    from gfnerf import persoctree as po
    from nerfstudio.cameras.cameras import Cameras

    W, H, F = 1152, 864, 1000.0
    oct_ = po.PersOctree.__new__(po.PersOctree)  # __init__ would build a whole tree and write .obj files into the cwd
    oct_.device = torch.device("cpu")

    # ---- distance_summary
    g = torch.Generator().manual_seed(3)
    dis_sets = [torch.rand(n, 1, generator=g) * 30 + 5 for n in (1, 2, 6, 7, 40, 200)] + [torch.full((9, 1), 12.5)]
    out["po_ds_sizes"] = np.asarray([d.numel() for d in dis_sets], np.int64)
    out["po_ds_dis"] = np.concatenate([d.view(-1).numpy() for d in dis_sets])
    out["po_ds_summary"] = np.asarray([oct_.distance_summary(d) for d in dis_sets], np.float64)
    out["po_ds_empty"] = np.asarray(oct_.distance_summary(torch.zeros(0, 1)), np.float64)

    # ---- PCA on random [n,12] values
    pts12 = torch.randn(4096, 12, generator=g) * torch.linspace(3.0, 0.2, 12)[None, :] + torch.randn(1, 12, generator=g)
    pts12 = pts12 @ torch.linalg.qr(torch.randn(12, 12, generator=g))[0]
    L, V = oct_.PCA(pts12)
    out["po_pca_pts"] = pts12.numpy(); out["po_pca_L"] = L.numpy(); out["po_pca_V"] = V.numpy()

    # ---- construct_trans on three camera sets
    c2w_all = _aerial_cameras(7, seed=21)
    n = c2w_all.shape[0]
    cams_all = Cameras(fx=torch.full((n,), F), fy=torch.full((n,), F), cx=torch.full((n,), W / 2), cy=torch.full((n,), H / 2),
                       camera_to_worlds=torch.from_numpy(c2w_all), width=W, height=H)
    out["po_c2w"] = c2w_all
    out["po_intr"] = np.asarray([F, F, W / 2, H / 2], np.float32)
    cases = [(np.arange(n), (0.0, 0.0, 0.0), 16.0, 101), (np.asarray([0, 1, 2, 7, 8, 9, 14, 15, 16, 21, 22]), (-30.0, -28.0, 1.0), 8.0, 102),
             (np.asarray([3, 10, 17, 24, 31, 38, 45]), (5.0, 2.0, -2.0), 4.0, 103)]
    for ci, (idx, center, side, seed) in enumerate(cases):
        visible = cams_all[torch.from_numpy(idx)]
        cen = torch.tensor(center, dtype=torch.float32).view(3, 1)
        gp = torch.Generator().manual_seed(seed)
        rand_pts = (torch.rand((4096, 3), generator=gp) - 0.5) * side + cen.permute(1, 0)
        cap = {}
        orig_pca, orig_inv = oct_.PCA, torch.linalg.inv

        def pca_spy(pts, _cap=cap, _orig=orig_pca):
            _cap["transed"] = pts.clone()
            Lr, Vr = _orig(pts)
            _cap["L"], _cap["V"] = Lr.clone(), Vr.clone()
            return Lr, Vr

        def inv_spy(x, _cap=cap, _orig=orig_inv):
            _cap.setdefault("inv_inputs", []).append(x.clone())
            return _orig(x)

        oct_.PCA = pca_spy
        torch.linalg.inv = inv_spy
        try:
            torch.manual_seed(seed)
            first = torch.randint(len(idx), (1,), dtype=torch.int32).item()  # the draw construct_trans is about to make (:107)
            torch.manual_seed(seed)
            ti = oct_.construct_trans(visible_cameras=visible, rand_pts=rand_pts, center=cen)
        finally:
            oct_.PCA = orig_pca
            torch.linalg.inv = orig_inv
        jac = cap["inv_inputs"][-1]  # the last inverse of the call is jac_warp2world = inv(V . dv_dxyz) (:169-170)
        assert jac.shape == (4096, 3, 3)
        out[f"po_ct{ci}_cams"] = idx.astype(np.int64); out[f"po_ct{ci}_center"] = np.asarray(center, np.float32)
        out[f"po_ct{ci}_side"] = np.asarray(side, np.float32); out[f"po_ct{ci}_first"] = np.asarray(first, np.int64)
        out[f"po_ct{ci}_pts"] = rand_pts.numpy()
        out[f"po_ct{ci}_w2xz"] = ti.w2xz.numpy(); out[f"po_ct{ci}_weight"] = ti.weight.numpy()
        out[f"po_ct{ci}_dis_summary"] = np.asarray(ti.dis_summary, np.float64)
        out[f"po_ct{ci}_transed"] = cap["transed"].numpy(); out[f"po_ct{ci}_L"] = cap["L"].numpy(); out[f"po_ct{ci}_V"] = cap["V"].numpy()
        out[f"po_ct{ci}_jac"] = jac.numpy()
        # the warp of a 32^3 grid as vis_transed_pts evaluates it (:334-371): run it, read its .obj back
        node = po.TreeNode(center=cen, side_len=side, pers_trans=ti)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "w.obj")
            oct_.vis_transed_pts(node, path)
            warped = np.asarray([[float(v) for v in ln.split()[1:]] for ln in open(path)], np.float64)
        assert warped.shape == (32 ** 3, 3)
        out[f"po_ct{ci}_grid_warped"] = warped[::7].astype(np.float32)  # every 7th point of the (deterministic) grid

    # ---- get_visi_cams (:248-290) on five boxes.  It casts one ray per 100th pixel of the image, GetVisiCams of the C++ build a
    # 128 x res_h grid (PersSampler.cpp:51-65): the two can only be compared as a sandwich.  `lower` = the prototype as it is
    # (every camera it reports does see the box); `upper` = the prototype run on cameras whose image is extended by a margin and
    # whose principal point is shifted in 25-pixel steps -- the union over the 16 shifts samples every 25th pixel of a slightly
    # larger image, on the box grown by 5 %: a camera outside `upper` sees nothing of the box.
    oct_.cameras = cams_all
    bounds = torch.zeros((n, 2)); bounds[:, 0] = 0.01; bounds[:, 1] = 512.0
    boxes = [((0.0, 0.0, 0.0), 64.0), ((-20.0, 10.0, -4.0), 16.0), ((30.0, -30.0, 2.0), 8.0), ((0.0, 0.0, -200.0), 32.0), ((3.0, 4.0, 40.0), 8.0)]
    lower, upper = [], []
    for center, side in boxes:
        cen = torch.tensor(center, dtype=torch.float32).view(3, 1)
        oct_.cameras = cams_all
        lower.append(oct_.get_visi_cams(side, cen, bounds).numpy())
        up = np.zeros(n, bool)
        for dx in (0, 25, 50, 75):
            for dy in (0, 25, 50, 75):
                oct_.cameras = Cameras(fx=torch.full((n,), F), fy=torch.full((n,), F), cx=torch.full((n,), W / 2 + 100 + dx),
                                       cy=torch.full((n,), H / 2 + 100 + dy), camera_to_worlds=torch.from_numpy(c2w_all), width=W + 300,
                                       height=H + 300)
                up |= oct_.get_visi_cams(side * 1.05, cen, bounds).numpy()
        upper.append(up)
    out["po_visi_boxes"] = np.asarray([list(c) + [s] for c, s in boxes], np.float32)
    out["po_visi_lower"] = np.stack(lower)  # [box, camera] bool
    out["po_visi_upper"] = np.stack(upper)
    out["po_visi_bounds"] = bounds.numpy()


def main():
    assert os.path.isdir(REF), "reference tree not mounted; fixtures can only be regenerated in the build container"
    _install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(1)
    out = {}
    for fn in (gen_mlp, gen_composite, gen_activations, gen_sh, gen_losses, gen_adam, gen_rays, gen_scheduler, gen_s3im, gen_pixel_samplers, gen_torch_field):
        fn(out)
    path = os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")
    po_out = {}
    gen_persoctree(po_out)
    path = os.path.join(HERE, "persoctree_vectors.npz")
    np.savez_compressed(path, **po_out)
    print(f"wrote {path}: {len(po_out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()

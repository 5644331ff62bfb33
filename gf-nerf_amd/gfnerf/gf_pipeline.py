"""GFNerfPipeline (gfnerf/gf_pipeline.py:94-299) reduced to what sits on the hot path: it wires a source of ray
batches to the model and returns (model_outputs, loss_dict, metrics_dict) like get_train_loss_dict (:146-192),
including the block-stage error-map write-back (:180-185).  Image loading from disk is out of scope (SURVEY 2d OUT
rows); any object with `next_train(step) -> (RayBundle, batch[, ray_indices])` can stand in: `SyntheticDataManager`
draws seeded synthetic rays, `ImageStackDataManager` is next_train of base_datamanager.py:923-948 over an image stack
resident in HBM (pixel samplers + ray generator of gfnerf_amd.data)."""
import numpy as np
import torch

from .. import data, scene
from .nerfacto import GFNeRFModel, GFNeRFModelConfig
from .perssampler import _Cameras
from .rays import RayBundle


class SyntheticDataManager:
    def __init__(self, c2w, rays_per_batch=4096, device="cuda:0", seed=0):
        self.c2w, self.R, self.dev, self.seed = c2w, rays_per_batch, device, seed

    def next_train(self, step):
        o, d, cam = scene.make_rays(self.c2w, self.R, seed=self.seed + step)
        gt = np.random.default_rng(self.seed + 7919 * (step + 1)).random((self.R, 3)).astype(np.float32)
        t = lambda a: torch.from_numpy(a).to(self.dev)  # noqa: E731
        steps = torch.full((self.R, 1), step, dtype=torch.int64, device=self.dev)
        return RayBundle(origins=t(o), directions=t(d), camera_indices=t(cam)[:, None], rel_camera_indices=t(cam)[:, None],
                         steps=steps), {"image": t(gt)}


class ImageStackDataManager:
    """train_image_dataloader (cached stack) + train_pixel_sampler + train_ray_generator of the reference's datamanager.
    images f32 or u8 [n,H,W,3] on the device; c2w [n_cams,3,4]; intr [n_cams,4] = (fx, fy, cx, cy); error_maps [n,H,W(,1)]
    switches to the ErrorPixelSampler (the block stage's data policy, base_datamanager.py:797-830)."""

    def __init__(self, images, c2w, intr, rays_per_batch=4096, image_idx=None, error_maps=None, generator=None):
        dev = images.device
        n = images.shape[0]
        idx = torch.arange(n, dtype=torch.int32, device=dev) if image_idx is None else image_idx.to(torch.int32).contiguous()
        self.error_map = data.ErrorMap(error_maps) if error_maps is not None else None
        self.image_batch = dict(image=images.contiguous(), image_idx=idx, rel_camera_idx=idx.clone(), error_map=self.error_map)
        cls = data.ErrorPixelSampler if error_maps is not None else data.PixelSampler
        self.train_pixel_sampler = cls(rays_per_batch, generator=generator)
        self.train_ray_generator = data.RayGenerator(c2w, intr)
        self.train_image_dataloader = self  # the object the pipeline calls _update_error_map on

    def next_train(self, step):
        rays, batch, ray_indices = data.next_train(self.image_batch, self.train_pixel_sampler, self.train_ray_generator)
        R = ray_indices.shape[0]
        steps = torch.full((R, 1), step, dtype=torch.int64, device=ray_indices.device)
        bundle = RayBundle(origins=rays["origins"], directions=rays["directions"], camera_indices=rays["camera_indices"][:, None],
                           rel_camera_indices=rays["rel_camera_indices"][:, None], steps=steps, pixel_area=rays["pixel_area"][:, None],
                           lookat_directions=rays["lookat"])
        return bundle, batch, ray_indices

    def update_error_map(self, ray_indices, pred_rgb, gt_rgb):
        """_update_error_map(ray_indices, |gt - pred|.sum(-1)) (dataloaders.py:140-142) in one launch."""
        if self.error_map is not None:
            self.error_map.update(ray_indices, pred_rgb, gt_rgb)


class GFNerfPipeline(torch.nn.Module):
    def __init__(self, model_config: GFNeRFModelConfig, c2w, intrinsics, device="cuda:0", test_mode="val", world_size=1,
                 local_rank=0, base_dir="", rays_per_batch=4096, datamanager=None):
        super().__init__()
        n = c2w.shape[0]
        cameras = _Cameras(torch.from_numpy(np.ascontiguousarray(c2w)), torch.from_numpy(np.ascontiguousarray(intrinsics)))
        bounds = torch.zeros((n, 2))
        bounds[:, 0] = 0.01                          # gf_pipeline.py:116-119
        bounds[:, 1] = 512
        self.datamanager = datamanager or SyntheticDataManager(c2w, rays_per_batch, device, seed=1000 * local_rank)
        self._model = GFNeRFModel(model_config, cameras, bounds, num_train_data=n, device=device, base_dir=base_dir)
        self.world_size, self.local_rank = world_size, local_rank

    @property
    def model(self):
        return self._model

    def get_train_loss_dict(self, step: int):
        nxt = self.datamanager.next_train(step)
        ray_bundle, batch, ray_indices = nxt if len(nxt) == 3 else (nxt[0], nxt[1], None)
        model_outputs = self.model(ray_bundle)
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        # update error map in block stage (:180-185)
        if self.model.field.cur_stage == "block_stage" and hasattr(self.datamanager, "train_image_dataloader") and ray_indices is not None:
            self.datamanager.update_error_map(ray_indices, model_outputs["rgb"].detach().contiguous(), batch["image"])
        return model_outputs, loss_dict, metrics_dict

    def get_param_groups(self):
        return self.model.get_param_groups()

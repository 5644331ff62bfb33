"""GFNerfPipeline (gfnerf/gf_pipeline.py:94-299) reduced to what sits on the hot path: it wires a source of ray
batches to the model and returns (model_outputs, loss_dict, metrics_dict) like get_train_loss_dict (:146-192).  The
reference's datamanager (image loading, pixel sampling, error maps) is out of scope (SURVEY 2d OUT rows); any object
with `next_train(step) -> (RayBundle, batch)` can stand in -- `SyntheticDataManager` draws seeded synthetic rays."""
import numpy as np
import torch

from .. import scene
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


class GFNerfPipeline(torch.nn.Module):
    def __init__(self, model_config: GFNeRFModelConfig, c2w, intrinsics, device="cuda:0", test_mode="val", world_size=1,
                 local_rank=0, base_dir="", rays_per_batch=4096):
        super().__init__()
        n = c2w.shape[0]
        cameras = _Cameras(torch.from_numpy(np.ascontiguousarray(c2w)), torch.from_numpy(np.ascontiguousarray(intrinsics)))
        bounds = torch.zeros((n, 2))
        bounds[:, 0] = 0.01                          # gf_pipeline.py:116-119
        bounds[:, 1] = 512
        self.datamanager = SyntheticDataManager(c2w, rays_per_batch, device, seed=1000 * local_rank)
        self._model = GFNeRFModel(model_config, cameras, bounds, num_train_data=n, device=device, base_dir=base_dir)
        self.world_size, self.local_rank = world_size, local_rank

    @property
    def model(self):
        return self._model

    def get_train_loss_dict(self, step: int):
        ray_bundle, batch = self.datamanager.next_train(step)
        model_outputs = self.model(ray_bundle)
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        return model_outputs, loss_dict, metrics_dict

    def get_param_groups(self):
        return self.model.get_param_groups()

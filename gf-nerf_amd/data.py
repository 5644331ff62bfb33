"""Batch assembly on the device: the callers immediately upstream of the hot path.

Mirrors, name for name, the classes the reference's data manager strings together in `next_train`
(nerfstudio/data/datamanagers/base_datamanager.py:923-948):

    image_batch = next(iter_train_image_dataloader)      # cached stack of images (+ error maps) -- here resident in HBM
    batch = train_pixel_sampler.sample(image_batch)      # PixelSampler / ErrorPixelSampler  (data/pixel_samplers.py)
    ray_bundle = train_ray_generator(batch["indices"])   # RayGenerator  (model_components/ray_generators.py:41-59)
    ... pipeline: _update_error_map(ray_indices, |gt - pred|.sum(-1))   (gfnerf/gf_pipeline.py:180-185)

The reference gathers pixels on the host and draws indices with Python's `random` and `torch.multinomial`; here the
image stack, the error map and its CDF live on the GPU, every step is a handful of small launches on the current
stream, and nothing synchronises.  Random streams cannot (and need not) match the reference's: the draws are torch
device generators; the arithmetic from random numbers to indices / colours / rays is what the parity tests pin.
"""
import torch

from . import ops


class PixelSampler:
    """nerfstudio/data/pixel_samplers.py:26-209 for a stacked image batch (`image_batch["image"]` a tensor)."""

    def __init__(self, num_rays_per_batch: int, keep_full_image: bool = False, generator=None, **kwargs) -> None:
        self.kwargs = kwargs
        self.num_rays_per_batch = num_rays_per_batch
        self.keep_full_image = keep_full_image
        self.generator = generator

    def set_num_rays_per_batch(self, num_rays_per_batch: int):
        self.num_rays_per_batch = num_rays_per_batch

    def sample_method(self, batch_size, num_images, image_height, image_width, mask=None, device="cuda"):
        """-> lin i64 [batch_size] (pixel (c, y, x) = unravel(lin)).  The masked branch is dead in the reference
        (it keys on "mask_bak", :93) and is not provided."""
        if mask is not None:
            raise NotImplementedError("masked pixel sampling is unreachable in the reference (pixel_samplers.py:93)")
        u = torch.rand((batch_size, 3), device=device, generator=self.generator)
        return ops.pixel_uniform(u, num_images, image_height, image_width)

    def _collate(self, batch, lin, keep_full_image):
        img = batch["image"]
        out = ops.pixel_collate(lin, img, batch.get("image_idx"), batch.get("rel_camera_idx"))
        collated = {"image": out["image"], "indices": out["indices"], "rel_camera_indices": out["rel_camera_indices"],
                    "slot": out["slot"], "coords": out["coords"], "lin": lin}
        for key, value in batch.items():  # any other per-pixel entry: value[c, y, x] (:97-99)
            value = value.map if isinstance(value, ErrorMap) else value
            if key in ("image", "image_idx", "rel_camera_idx") or value is None or not isinstance(value, torch.Tensor):
                continue
            collated[key] = value.reshape(img.shape[0] * img.shape[1] * img.shape[2], -1)[lin]
        if keep_full_image:
            collated["full_image"] = img
        return collated

    def collate_image_dataset_batch(self, batch, num_rays_per_batch: int, keep_full_image: bool = False):
        n, H, W, _ = batch["image"].shape
        lin = self.sample_method(num_rays_per_batch, n, H, W, device=batch["image"].device)
        return self._collate(batch, lin, keep_full_image)

    def sample(self, image_batch):
        if not isinstance(image_batch["image"], torch.Tensor):
            raise ValueError("image_batch['image'] must be a torch.Tensor resident on the GPU")
        return self.collate_image_dataset_batch(image_batch, self.num_rays_per_batch, keep_full_image=self.keep_full_image)


class ErrorMap:
    """The cached batch's `error_map` [n_images,H,W,1] plus the two-level CDF the weighted sampler reads."""

    def __init__(self, error_map: torch.Tensor):
        if error_map.dim() == 3:
            error_map = error_map[..., None]
        self.map = error_map.contiguous().float()
        self.block_sums = ops.errmap_block_sums(self.map)

    def refresh(self):
        """Partial sums from scratch (after the map was written by anything but `update`)."""
        ops.errmap_block_sums(self.map, out=self.block_sums)

    def update(self, ray_indices, pred_rgb, gt_rgb):
        """CacheDataloader._update_error_map (dataloaders.py:140-142) with the error of gf_pipeline.py:181-184.  Like the
        reference, column 0 of ray_indices is used as the position in the cached stack."""
        ops.errmap_update(self.map, self.block_sums, ray_indices.to(torch.int32).contiguous(), pred_rgb, gt_rgb)


class ErrorPixelSampler(PixelSampler):
    """nerfstudio/data/pixel_samplers.py:594-844: 20 % of a batch follows the error map, the rest is uniform."""

    def __init__(self, num_rays_per_batch: int, keep_full_image: bool = False, generator=None, **kwargs) -> None:
        super().__init__(num_rays_per_batch, keep_full_image, generator, **kwargs)
        self.weighted_choice_ratio = 0.2
        self._ws = None
        self.last_status = None

    def sample_method(self, batch_size, num_images, image_height, image_width, error_map=None, device="cuda"):
        """error_map: an ErrorMap (or a raw tensor, whose partial sums are then rebuilt on every call)."""
        if error_map is None:
            raise ValueError("ErrorPixelSampler needs the batch's error_map")
        em = error_map if isinstance(error_map, ErrorMap) else ErrorMap(error_map)
        n_weighted = int(batch_size * self.weighted_choice_ratio)  # :699
        bits = torch.randint(-2 ** 63, 2 ** 63 - 1, (batch_size,), dtype=torch.int64, device=em.map.device,
                             generator=self.generator)
        lin, self.last_status, self._ws = ops.errmap_sample(em.map, em.block_sums, bits, n_weighted, workspace=self._ws)
        return lin

    def collate_image_dataset_batch(self, batch, num_rays_per_batch: int, keep_full_image: bool = False):
        n, H, W, _ = batch["image"].shape
        em = batch["error_map"]
        lin = self.sample_method(num_rays_per_batch, n, H, W, error_map=em, device=batch["image"].device)
        return self._collate(batch, lin, keep_full_image)


class RayGenerator:
    """nerfstudio/model_components/ray_generators.py:41-59 without a pose optimiser (the GF-NeRF config disables camera
    optimisation): c2w f32 [n_cams,3,4], intr f32 [n_cams,4] = (fx, fy, cx, cy) device tensors."""

    def __init__(self, c2w, intr):
        self.c2w, self.intr = c2w.contiguous(), intr.contiguous()

    def __call__(self, ray_indices, coords=None):
        """ray_indices [n,3] = (camera, row, col) -> dict(origins, directions, lookat, pixel_area, dir_norm, camera_indices)."""
        idx = ray_indices.to(torch.int32)
        cam = idx[:, 0].contiguous()
        if coords is None:
            coords = (idx[:, 1:3].to(torch.float32) + 0.5).contiguous()  # image_coords[y, x] (cameras.py get_image_coords)
        rays = ops.generate_rays(self.c2w, self.intr, cam_idx=cam, coords=coords)
        rays["camera_indices"] = cam
        return rays


def next_train(image_batch, pixel_sampler, ray_generator):
    """VanillaDataManager.next_train (base_datamanager.py:923-948) -> (rays, batch, ray_indices)."""
    batch = pixel_sampler.sample(image_batch)
    rays = ray_generator(batch["indices"], coords=batch["coords"])
    rays["rel_camera_indices"] = batch["rel_camera_indices"]
    return rays, batch, batch["indices"]

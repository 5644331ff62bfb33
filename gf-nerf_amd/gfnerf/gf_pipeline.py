"""GFNerfPipeline (gfnerf/gf_pipeline.py:94-299): the reference's constructor -- `GFNerfPipeline(config, device, test_mode,
world_size, local_rank, base_dir)` with `config.datamanager.setup(device=, test_mode=, world_size=, local_rank=)` and
`config.model.setup(scene_box=, num_train_data=, metadata=, cameras=, bounds=, base_dir=)` (:94-138) -- and its methods
(get_train_loss_dict with the block-stage error-map write-back :146-192, get_eval_loss_dict, get_eval_image_metrics_and_images,
get_average_eval_image_metrics, load_pipeline, get_training_callbacks, get_param_groups).

The datamanager is the host's (GFNerfDataManager, nerfstudio/data/datamanagers/base_datamanager.py:589-992; image loading from
disk is out of scope).  Two datamanagers that implement the part of its interface the pipeline touches live here, for
runs that keep everything in HBM and for the tests: `SyntheticDataManager` (seeded synthetic rays) and `ImageStackDataManager`
(next_train of :923-948 over an image stack resident in HBM with the pixel samplers + ray generator of gfnerf_amd.data, the
ErrorPixelSampler data policy of the block stage, and `setup_train_split_oct`: each block trains on its own camera cluster).

Multi-GPU: the reference wraps the model in DDP (:136-138), which never all-reduces the hash table (it is not a registered
parameter: hash_3d_anchored.py:26,73-78).  Here nothing is wrapped.  Global stage, world_size > 1: the engine's TrainSync
MAX-reduces the octree statistics and averages every gradient inside the backward (gfnerf_amd.dist), so the Trainer's own
optimizer.step() sees the reduced gradients.  Focal stage: nothing is exchanged -- every rank trains a block of its own
(GFNeRFModel.stage_of: block (period * world + rank) mod n_split) on that block's camera cluster and writes that block's
checkpoint alone (GFNeRFField.save_table: private file, os.replace)."""
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Type

import numpy as np
import torch
import torch.distributed as dist

from .. import data, scene
from ._host import InstantiateConfig, ModelConfig, Pipeline, TrainingCallback, TrainingCallbackAttributes
from .nerfacto import GFNeRFModelConfig
from .perssampler import _Cameras
from .rays import RayBundle


def module_wrapper(ddp_or_model):
    """gf_pipeline.py:56-62: the model behind a DDP wrapper, or the model itself (this build never wraps)."""
    return ddp_or_model.module if hasattr(ddp_or_model, "module") and not hasattr(ddp_or_model, "get_outputs") else ddp_or_model


# ------------------------------------------------------------------------------------------------ datamanagers
class _TrainDataset:
    """What the pipeline reads off datamanager.train_dataset: scene_box, metadata, len()."""

    def __init__(self, n, scene_box=None, metadata=None):
        self._n, self.scene_box, self.metadata = n, scene_box, metadata or {}

    def __len__(self):
        return self._n


class _DataManagerBase:
    def __init__(self, c2w, intrinsics, device):
        self.device = torch.device(device)
        c2w_t = torch.as_tensor(np.ascontiguousarray(c2w) if isinstance(c2w, np.ndarray) else c2w).float()
        K = torch.as_tensor(np.ascontiguousarray(intrinsics) if isinstance(intrinsics, np.ndarray) else intrinsics).float()
        self.train_dataparser_outputs = SimpleNamespace(cameras=_Cameras(c2w_t, K))
        self.train_dataset = _TrainDataset(c2w_t.shape[0], scene_box=SimpleNamespace(aabb=torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])))
        self.eval_dataset = self.train_dataset
        self.split_version = 0     # bumped whenever setup_train_split_oct changes what next_train draws from
        self._split_idx = None

    def to(self, device):
        self.device = torch.device(device)
        return self

    def get_training_callbacks(self, training_callback_attributes) -> List[TrainingCallback]:
        return []

    def get_param_groups(self) -> Dict[str, List[torch.nn.Parameter]]:
        return {}


class SyntheticDataManager(_DataManagerBase):
    def __init__(self, c2w, intrinsics=None, rays_per_batch=4096, device="cuda:0", seed=0):
        if intrinsics is None:
            intrinsics = np.repeat(np.array([[[scene.FOCAL, 0, scene.WIDTH / 2], [0, scene.FOCAL, scene.HEIGHT / 2], [0, 0, 1]]], np.float32), c2w.shape[0], 0)
        super().__init__(c2w, intrinsics, device)
        self.c2w, self.R, self.seed = np.asarray(c2w), rays_per_batch, seed
        self.cams = None  # setup_train_split_oct: the active block's camera cluster

    def setup_train_split_oct(self, cameras_labels, split_idx, sample_tmp_dir=None):
        """base_datamanager.py setup_train_split_oct: from now on rays are drawn from the cameras of cluster `split_idx` only.
        (The update_datamanager callback calls this after EVERY iteration, nerfacto.py:491-519: unchanged split, nothing to do.)"""
        if self._split_idx == int(split_idx):
            return
        self.cams = np.nonzero(cameras_labels.reshape(-1).cpu().numpy() == int(split_idx))[0]
        self._split_idx = int(split_idx)
        self.split_version += 1

    def next_train(self, step):
        o, d, cam = scene.make_rays(self.c2w, self.R, seed=self.seed + step, cams=self.cams)
        gt = np.random.default_rng(self.seed + 7919 * (step + 1)).random((self.R, 3)).astype(np.float32)
        t = lambda a: torch.from_numpy(a).to(self.device)  # noqa: E731
        steps = torch.full((self.R, 1), step, dtype=torch.int64, device=self.device)
        bundle = RayBundle(origins=t(o), directions=t(d), camera_indices=t(cam)[:, None], rel_camera_indices=t(cam)[:, None], steps=steps)
        bundle._step_host = (steps, int(step))  # the step as a host integer too (GFNeRFModel._bundle: no read-back of `steps`)
        return bundle, {"image": t(gt)}, None

    def next_eval(self, step):
        bundle, batch, _ = self.next_train(step)
        bundle.steps = None
        return bundle, batch


class ImageStackDataManager(_DataManagerBase):
    """train_image_dataloader (cached stack) + train_pixel_sampler + train_ray_generator of the reference's datamanager.
    images f32 or u8 [n,H,W,3] on the device; c2w [n_cams,3,4]; intr [n_cams,4] = (fx, fy, cx, cy); error_maps [n,H,W(,1)]
    switches to the ErrorPixelSampler (the block stage's data policy, base_datamanager.py:797-830)."""

    def __init__(self, images, c2w, intr, rays_per_batch=4096, image_idx=None, error_maps=None, generator=None):
        dev = images.device
        K = torch.zeros((c2w.shape[0], 3, 3), dtype=torch.float32)
        ic = intr.detach().cpu()
        K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = ic[:, 0], ic[:, 1], ic[:, 2], ic[:, 3], 1.0
        super().__init__(c2w.detach().cpu(), K, dev)
        n = images.shape[0]
        idx = torch.arange(n, dtype=torch.int32, device=dev) if image_idx is None else image_idx.to(torch.int32).contiguous()
        self.rays_per_batch, self.generator = rays_per_batch, generator
        self._all = dict(image=images.contiguous(), image_idx=idx, error_maps=error_maps)
        self.train_ray_generator = data.RayGenerator(c2w, intr)
        self.train_image_dataloader = self  # the object the pipeline calls _update_error_map on
        self._select(None)

    def _select(self, keep):
        """(Re)build the cached batch over the images `keep` (positions in the stack; None = all)."""
        a = self._all
        sel = slice(None) if keep is None else keep
        em = a["error_maps"]
        self.error_map = data.ErrorMap(em[sel].clone() if keep is not None else em) if em is not None else None
        idx = a["image_idx"][sel].contiguous()
        self.image_batch = dict(image=a["image"][sel].contiguous(), image_idx=idx, rel_camera_idx=idx.clone(), error_map=self.error_map)
        cls = data.ErrorPixelSampler if em is not None else data.PixelSampler
        self.train_pixel_sampler = cls(self.rays_per_batch, generator=self.generator)

    def setup_train_split_oct(self, cameras_labels, split_idx, sample_tmp_dir=None):
        """GFNerfDataManager.setup_train_split_oct: the cached batch becomes the images of camera cluster `split_idx` (their error
        maps with them) -- every focal block trains on its own cameras (perssampler.py:216-242)."""
        if self._split_idx == int(split_idx):  # called after every iteration (nerfacto.py:491-519): unchanged split, nothing to do
            return
        labels = cameras_labels.reshape(-1).to(self._all["image_idx"].device)
        keep = torch.nonzero(labels[self._all["image_idx"].long()] == int(split_idx)).reshape(-1)
        assert keep.numel() > 0, f"camera cluster {split_idx} holds no image of the stack"
        self._select(keep)
        self._split_idx = int(split_idx)
        self.split_version += 1

    def set_error_maps(self, maps):
        """render_init_error_maps (nerfacto.py:354-430) hands over one |gt - pred| map per training image."""
        self._all["error_maps"] = maps[..., None] if maps.dim() == 3 else maps
        self._select(None)
        self._split_idx = None
        self.split_version += 1

    def train_images_for_error_maps(self, down_scale_factor=8, step=None):
        """(gt image [H,W,3] in [0,1], RayBundle of every `down_scale_factor`-th pixel [h,w,.]) per training image (:383-391)."""
        imgs, idx = self._all["image"], self._all["image_idx"]
        H, W = imgs.shape[1:3]
        ys = torch.arange(0, H, down_scale_factor, device=imgs.device)
        xs = torch.arange(0, W, down_scale_factor, device=imgs.device)
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        for k in range(imgs.shape[0]):
            cam = int(idx[k])
            ind = torch.stack([torch.full_like(yy, cam), yy, xx], -1).reshape(-1, 3)
            rays = self.train_ray_generator(ind)
            shp = (ys.numel(), xs.numel())
            cam_t = torch.full(shp + (1,), cam, dtype=torch.int32, device=imgs.device)
            gt = imgs[k].float() / 255.0 if imgs.dtype == torch.uint8 else imgs[k]
            steps = None if step is None else torch.full(shp + (1,), int(step), dtype=torch.int64, device=imgs.device)
            yield gt, RayBundle(origins=rays["origins"].view(*shp, 3), directions=rays["directions"].view(*shp, 3), camera_indices=cam_t,
                                rel_camera_indices=cam_t, steps=steps)

    def next_train(self, step):
        rays, batch, ray_indices = data.next_train(self.image_batch, self.train_pixel_sampler, self.train_ray_generator)
        R = ray_indices.shape[0]
        steps = torch.full((R, 1), step, dtype=torch.int64, device=ray_indices.device)
        bundle = RayBundle(origins=rays["origins"], directions=rays["directions"], camera_indices=rays["camera_indices"][:, None],
                           rel_camera_indices=rays["rel_camera_indices"][:, None], steps=steps, pixel_area=rays["pixel_area"][:, None],
                           lookat_directions=rays["lookat"])
        bundle._step_host = (steps, int(step))
        return bundle, batch, ray_indices

    def next_eval(self, step):
        bundle, batch, _ = self.next_train(step)
        bundle.steps = None
        return bundle, batch

    def next_eval_image(self, step):
        k = step % self._all["image"].shape[0]
        for i, (gt, bundle) in enumerate(self.train_images_for_error_maps(down_scale_factor=1)):
            if i == k:
                return k, bundle, {"image": gt}

    @property
    def fixed_indices_eval_dataloader(self):
        return [(bundle, {"image": gt}) for gt, bundle in self.train_images_for_error_maps(down_scale_factor=1)]

    def error_map_sink(self, ray_indices, slot=None):
        """What the fused loss launch needs to do update_error_map's work itself (GFNeRFModel.set_error_map_sink): (error map, its block sums, the rays'
        (c, y, x) rows, their positions in the cached stack), or None without an error map."""
        if self.error_map is None:
            return None
        idx = ray_indices if ray_indices.dtype == torch.int32 and ray_indices.is_contiguous() else ray_indices.to(torch.int32).contiguous()
        return (self.error_map.map, self.error_map.block_sums, idx, slot)

    def update_error_map(self, ray_indices, pred_rgb, gt_rgb):
        """_update_error_map(ray_indices, |gt - pred|.sum(-1)) (dataloaders.py:140-142; the error of gf_pipeline.py:181-184)
        fused into one launch."""
        if self.error_map is not None:
            slot_indices = ray_indices
            self.error_map.update(slot_indices, pred_rgb, gt_rgb)


@dataclass
class HbmDataManagerConfig(InstantiateConfig):
    """Config of the two HBM-resident datamanagers above, with the host's `setup(device=, test_mode=, world_size=, local_rank=)`
    protocol (gf_pipeline.py:105-107).  c2w [n,3,4] / intrinsics [n,3,3] numpy; images (optional) a device tensor [n,H,W,3]."""
    _target: Type = None
    c2w: Any = None
    intrinsics: Any = None
    images: Any = None
    image_idx: Any = None       # camera index of every image of the stack (default: image k is camera k)
    error_maps: Any = None
    train_num_rays_per_batch: int = 4096
    seed: int = 0

    def setup(self, **kwargs) -> Any:
        dev = kwargs.get("device", "cuda:0")
        rank = int(kwargs.get("local_rank", 0))
        if self.images is None:
            return SyntheticDataManager(self.c2w, self.intrinsics, self.train_num_rays_per_batch, dev, seed=self.seed + 1000 * rank)
        K = np.asarray(self.intrinsics)
        intr = torch.from_numpy(np.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], -1).astype(np.float32)).to(dev)
        gen = torch.Generator(device=dev).manual_seed(self.seed + 1000 * rank)
        idx = None if self.image_idx is None else torch.as_tensor(np.asarray(self.image_idx)).to(dev)
        return ImageStackDataManager(self.images, torch.from_numpy(np.ascontiguousarray(self.c2w)).to(dev), intr, self.train_num_rays_per_batch,
                                     image_idx=idx, error_maps=self.error_maps, generator=gen)


# ------------------------------------------------------------------------------------------------ the pipeline
@dataclass
class GFNerfPipelineConfig(InstantiateConfig):
    """gf_pipeline.py:65-74."""
    _target: Type = field(default_factory=lambda: GFNerfPipeline)
    datamanager: Any = field(default_factory=HbmDataManagerConfig)
    model: ModelConfig = field(default_factory=GFNeRFModelConfig)
    prefetch_next_batch: bool = True
    """(this build) get_train_loss_dict(step) also draws batch step + 1 -- after this step's forward and error-map write-back, so the
    draw sees exactly the state it would see one call later -- and starts its ray sampler on the engine's second stream, beside this
    step's backward.  False: every batch is drawn and sampled when its own step begins, like the reference (:146-148)."""


def _record_stream(obj, stream, _depth=0):
    """record_stream on every device tensor inside a batch / ray bundle / tuple (tensors made on one stream, consumed on another)."""
    if obj is None or _depth > 4:
        return
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream, _depth + 1)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream, _depth + 1)
    elif hasattr(obj, "__dict__"):
        for v in vars(obj).values():
            _record_stream(v, stream, _depth + 1)


class GFNerfPipeline(Pipeline):
    def __init__(self, config: GFNerfPipelineConfig, device: str, test_mode: str = "val", world_size: int = 1, local_rank: int = 0,
                 base_dir: str = ""):
        super().__init__()
        self.config = config
        self.test_mode = test_mode
        self.datamanager = config.datamanager.setup(device=device, test_mode=test_mode, world_size=world_size, local_rank=local_rank)
        self.datamanager.to(device)
        assert self.datamanager.train_dataset is not None, "Missing input dataset"
        cameras = self.datamanager.train_dataparser_outputs.cameras
        bounds = torch.zeros((len(cameras), 2))
        bounds[:, 0] = 0.01                          # :116-119 (for visibility field)
        bounds[:, 1] = 512
        bounds = bounds.to(device)
        cameras = cameras.to(device)
        self._model = config.model.setup(scene_box=self.datamanager.train_dataset.scene_box, num_train_data=len(self.datamanager.train_dataset),
                                         metadata=self.datamanager.train_dataset.metadata, cameras=cameras, bounds=bounds, base_dir=base_dir,
                                         device=device)
        self.model.to(device)
        self.world_size, self.local_rank = world_size, local_rank
        self._next_batch = None   # (step, datamanager split version, next_train(step) result) drawn one call early
        self._rays_per_batch = None  # rays of the last training batch (the next S3IM index vector is drawn for as many)
        if world_size > 1:
            # no DDP wrapper (see the module docstring): the engine exchanges octree statistics and gradients itself
            assert self.model.hot.sync.world == world_size, "torch.distributed must be initialised before the pipeline is built (scripts/train.py:124)"
            dist.barrier()

    @property
    def device(self):
        return self.model.device

    def get_train_loss_dict(self, step: int):
        """:146-192."""
        hot = getattr(self.model, "hot", None)
        if hot is not None:
            hot.run_deferred()  # a draw nobody has run yet (no fused optimiser stepped since)
        pre, self._next_batch = self._next_batch, None
        if pre is not None and pre[0] == step and pre[1] == getattr(self.datamanager, "split_version", 0):
            nxt = pre[2]
        else:  # nothing drawn ahead, or the callbacks changed the camera split since (update_datamanager, nerfacto.py:491-519)
            nxt = self.datamanager.next_train(step)
        ray_bundle, batch, ray_indices = nxt if len(nxt) == 3 else (nxt[0], nxt[1], None)
        self._rays_per_batch = int(ray_bundle.origins.shape[0]) if ray_bundle.origins.dim() == 2 else int(ray_bundle.origins.numel() // 3)
        model_outputs = self.model(ray_bundle)
        # update error map in block stage (:180-185).  A datamanager that keeps its error map in HBM (error_map_sink) has the write-back done by the
        # model's fused loss launch -- the same |gt - pred|.sum(-1) into the same pixels, in the launch that computes the losses from them
        update_map = (self.model.field.cur_stage == "block_stage" and hasattr(self.datamanager, "train_image_dataloader") and ray_indices is not None)
        loader = self.datamanager.train_image_dataloader if update_map else None
        sink = None
        if update_map and hasattr(loader, "error_map_sink") and hasattr(self.model, "set_error_map_sink") and self.model.training:
            sink = loader.error_map_sink(ray_indices, batch.get("slot"))
            self.model.set_error_map_sink(sink)
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        if sink is not None and getattr(self.model, "_errmap_sink", None) is not None:
            self.model.set_error_map_sink(None)  # (the fused node did not run -- another loss configuration: the write-back is done here after all)
            sink = None
        if update_map and sink is None:
            pred_color, gt_color = model_outputs["rgb"].detach().contiguous(), batch["image"]
            if hasattr(loader, "update_error_map"):   # |gt - pred|.sum(-1) and the write-back in one launch
                slots = torch.cat([batch["slot"][:, None], ray_indices[:, 1:3]], 1) if "slot" in batch else ray_indices
                loader.update_error_map(slots, pred_color, gt_color)
            else:
                loader._update_error_map(ray_indices, torch.sum(torch.abs(gt_color - pred_color), dim=-1))
        if getattr(self.config, "prefetch_next_batch", False) and self.model.training and hasattr(self.model, "prefetch_samples"):
            self._draw_ahead(step + 1)
        return model_outputs, loss_dict, metrics_dict

    def _draw_ahead(self, step):
        """Batch `step` drawn now -- as far as the device is concerned: what the draw sees (error map, random streams, sampler state) is
        what it sees here, its kernels and the ray sampler behind them run on the engine's second stream from this point of the main
        stream on, beside the backward.  The HOST part (a few dozen small launches) is put off until the backward and the optimiser
        step are enqueued (HotPath.defer): between the forward and the backward the host is what the GPU waits for."""
        hot = self.model.hot
        here = torch.cuda.Event()
        here.record()

        def draw():
            main, side = torch.cuda.current_stream(), hot.side_stream()
            side.wait_event(here)
            with torch.cuda.stream(side):
                nxt = self.datamanager.next_train(step)
                self._next_batch = (step, getattr(self.datamanager, "split_version", 0), nxt)
                self.model.prefetch_samples(nxt[0])
                if self._rays_per_batch and hasattr(self.model, "draw_s3im_index"):
                    # the next iteration's S3IM index vector: a sort and eight small launches off the main stream -- and BEHIND the sampler, which is
                    # what the next forward waits for
                    self.model.draw_s3im_index(self._rays_per_batch)
            _record_stream(nxt, main)  # allocated on the second stream, used by the next iteration on this one
            _record_stream(getattr(self.model, "_prefetched", None), main)

        # (global stage: the backward is followed by the weight-gradient kernels, whose LDS the sampler's traversal cannot share -- started
        # late it ends late: 2.23 against 2.04 ms per iteration; there the host draws at once)
        # (a forward that was one graph replay leaves the host ahead of the device: nothing to gain, and the sampler would start late)
        if hot.fused_optimizers and self.model.field.cur_stage == "block_stage" and not hot.last_forward_replayed:
            hot.defer(draw)
        else:
            draw()

    def forward(self):
        raise NotImplementedError

    def get_eval_loss_dict(self, step: int):
        """:199-213."""
        self.eval()
        ray_bundle, batch = self.datamanager.next_eval(step)
        model_outputs = self.model(ray_bundle)
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        self.train()
        return model_outputs, loss_dict, metrics_dict

    def get_eval_image_metrics_and_images(self, step: int):
        """:215-233."""
        self.eval()
        image_idx, camera_ray_bundle, batch = self.datamanager.next_eval_image(step)
        outputs = self.model.get_outputs_for_camera_ray_bundle(camera_ray_bundle)
        metrics_dict, images_dict = self.model.get_image_metrics_and_images(outputs, batch)
        assert "image_idx" not in metrics_dict
        metrics_dict["image_idx"] = image_idx
        assert "num_rays" not in metrics_dict
        metrics_dict["num_rays"] = len(camera_ray_bundle)
        self.train()
        return metrics_dict, images_dict

    def get_average_eval_image_metrics(self, step: Optional[int] = None):
        """:234-268: the metrics of every eval image, averaged (+ rays/s and frames/s of the full-image renders)."""
        from time import time

        self.eval()
        metrics_dict_list = []
        for camera_ray_bundle, batch in self.datamanager.fixed_indices_eval_dataloader:
            inner_start = time()
            height, width = camera_ray_bundle.shape
            outputs = self.model.get_outputs_for_camera_ray_bundle(camera_ray_bundle)
            metrics_dict, _ = self.model.get_image_metrics_and_images(outputs, batch)
            torch.cuda.synchronize()
            metrics_dict["num_rays_per_sec"] = height * width / (time() - inner_start)
            metrics_dict["fps"] = metrics_dict["num_rays_per_sec"] / (height * width)
            metrics_dict_list.append(metrics_dict)
        keys = set.intersection(*(set(m) for m in metrics_dict_list))
        out = {key: float(torch.mean(torch.tensor([m[key] for m in metrics_dict_list]))) for key in keys}
        self.train()
        return out

    def load_pipeline(self, loaded_state: Dict[str, Any], step: int) -> None:
        """:270-280."""
        state = {key.replace("module.", ""): value for key, value in loaded_state.items()}
        self._model.update_to_step(step)
        self.load_state_dict(state, strict=True)

    def get_training_callbacks(self, training_callback_attributes: TrainingCallbackAttributes) -> List[TrainingCallback]:
        datamanager_callbacks = self.datamanager.get_training_callbacks(training_callback_attributes)
        model_callbacks = self.model.get_training_callbacks(training_callback_attributes)
        return datamanager_callbacks + model_callbacks

    def get_param_groups(self) -> Dict[str, List[torch.nn.Parameter]]:
        return {**self.datamanager.get_param_groups(), **self.model.get_param_groups()}

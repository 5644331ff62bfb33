"""The host framework the plugin sits in.

This package replaces the `gfnerf/` package of the reference tree; everything around it -- the Trainer, the Optimizers, the
datamanager, the config machinery, the method registry -- is the reference's (modified) Nerfstudio and stays what it is.
When that tree is importable, the names below ARE its classes: `MethodSpecification` is
nerfstudio.plugins.types.MethodSpecification (so `isinstance(spec, MethodSpecification)` at plugins/registry.py:43 holds),
`Model` is nerfstudio.models.base_model.Model, and so on.  When it is not (the GPU test boxes carry no Nerfstudio),
minimal stand-ins with the same constructor and method signatures are defined here so that the plugin surface can be
built, driven and tested on its own; tests/test_api_surface.py holds both the stand-ins and the plugin classes to the
names recorded from the reference sources (tests/golden/api_signatures.json).

Host classes by origin:
  configs/base_config.py:51-58      InstantiateConfig(.setup(**kwargs) -> self._target(self, **kwargs))
  models/base_model.py:36-220       ModelConfig, Model
  fields/base_field.py:40-129       Field
  pipelines/base_pipeline.py        Pipeline
  plugins/types.py:23-33            MethodSpecification
  engine/trainer.py                 TrainerConfig (the fields gfnerf/config.py sets)
  engine/optimizers.py:38-193       AdamOptimizerConfig, Optimizers (with the GF add_optimizer / delete_optimizer)
  engine/callbacks.py:25-125        TrainingCallback, TrainingCallbackAttributes, TrainingCallbackLocation
"""
from collections import defaultdict
from dataclasses import dataclass, field
from enum import Enum, auto
from inspect import signature
from typing import Any, Callable, Dict, List, Optional, Tuple, Type

import torch
from torch import nn

try:  # the reference's Nerfstudio tree around us
    from nerfstudio.field_components.field_heads import FieldHeadNames
    from nerfstudio.configs.base_config import InstantiateConfig
    from nerfstudio.engine.callbacks import TrainingCallback, TrainingCallbackAttributes, TrainingCallbackLocation
    from nerfstudio.engine.optimizers import AdamOptimizerConfig, Optimizers
    from nerfstudio.engine.trainer import TrainerConfig
    from nerfstudio.fields.base_field import Field
    from nerfstudio.models.base_model import Model, ModelConfig
    from nerfstudio.pipelines.base_pipeline import Pipeline
    from nerfstudio.plugins.types import MethodSpecification

    HOST = "nerfstudio"
except ImportError:
    HOST = None

if HOST is None:

    class FieldHeadNames(str, Enum):
        """field_components/field_heads.py: the two heads this path produces (the values are the reference's strings)."""
        RGB = "rgb"
        DENSITY = "density"

        def __hash__(self):
            return hash(self.value)

    @dataclass
    class InstantiateConfig:
        _target: Type = None

        def setup(self, **kwargs) -> Any:
            return self._target(self, **kwargs)

    @dataclass
    class ModelConfig(InstantiateConfig):
        _target: Type = None
        enable_collider: bool = True
        collider_params: Optional[Dict[str, float]] = field(default_factory=lambda: {"near_plane": 2.0, "far_plane": 6.0})
        loss_coefficients: Dict[str, float] = field(default_factory=lambda: {"rgb_loss_coarse": 1.0, "rgb_loss_fine": 1.0})
        eval_num_rays_per_chunk: int = 4096

    class TrainingCallbackLocation(Enum):
        BEFORE_TRAIN_ITERATION = auto()
        AFTER_TRAIN_ITERATION = auto()

    @dataclass
    class TrainingCallbackAttributes:
        optimizers: Any
        grad_scaler: Any
        pipeline: Any

    class TrainingCallback:
        def __init__(self, where_to_run: List[TrainingCallbackLocation], func: Callable, update_every_num_iters: Optional[int] = None,
                     iters: Optional[Tuple[int, ...]] = None, args: Optional[List] = None, kwargs: Optional[Dict] = None):
            assert "step" in signature(func).parameters.keys(), f"'step: int' must be an argument in the callback function 'func': {func.__name__}"
            self.where_to_run, self.func = where_to_run, func
            self.update_every_num_iters, self.iters = update_every_num_iters, iters
            self.args = args if args is not None else []
            self.kwargs = kwargs if kwargs is not None else {}

        def run_callback(self, step: int) -> None:
            if self.update_every_num_iters is not None:
                if step % self.update_every_num_iters == 0:
                    self.func(*self.args, **self.kwargs, step=step)
            elif self.iters is not None and step in self.iters:
                self.func(*self.args, **self.kwargs, step=step)

        def run_callback_at_location(self, step: int, location: TrainingCallbackLocation) -> None:
            if location in self.where_to_run:
                self.run_callback(step=step)

    class Model(nn.Module):
        config: ModelConfig

        def __init__(self, config: ModelConfig, scene_box, num_train_data: int, **kwargs) -> None:
            super().__init__()
            self.config = config
            self.scene_box = scene_box
            self.render_aabb = None
            self.num_train_data = num_train_data
            self.kwargs = kwargs
            self.collider = None
            self.populate_modules()
            self.callbacks = None
            self.device_indicator_param = nn.Parameter(torch.empty(0))

        @property
        def device(self):
            return self.device_indicator_param.device

        def get_training_callbacks(self, training_callback_attributes: TrainingCallbackAttributes) -> List[TrainingCallback]:
            return []

        def populate_modules(self):
            """The host's default builds a NearFarCollider when config.enable_collider; a collider only writes nears / fars
            into the bundle, which the GF sampler ignores (PersSampler_cuda.cu:326-327)."""

        def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
            raise NotImplementedError

        def get_outputs(self, ray_bundle) -> Dict[str, torch.Tensor]:
            raise NotImplementedError

        def forward(self, ray_bundle) -> Dict[str, torch.Tensor]:
            if self.collider is not None:
                ray_bundle = self.collider(ray_bundle)
            return self.get_outputs(ray_bundle)

        def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
            return {}

        def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
            raise NotImplementedError

        @torch.no_grad()
        def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle) -> Dict[str, torch.Tensor]:
            """base_model.py:163-186: chunks of config.eval_num_rays_per_chunk rays through forward, concatenated to [H,W,.]."""
            num_rays_per_chunk = self.config.eval_num_rays_per_chunk
            image_height, image_width = camera_ray_bundle.origins.shape[:2]
            num_rays = len(camera_ray_bundle)
            outputs_lists = defaultdict(list)
            for i in range(0, num_rays, num_rays_per_chunk):
                ray_bundle = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, i + num_rays_per_chunk)
                for name, output in self.forward(ray_bundle=ray_bundle).items():
                    outputs_lists[name].append(output)
            return {name: torch.cat(lst).view(image_height, image_width, -1) for name, lst in outputs_lists.items() if torch.is_tensor(lst[0])}

        def get_image_metrics_and_images(self, outputs, batch):
            raise NotImplementedError

        def load_model(self, loaded_state: Dict[str, Any]) -> None:
            state = {key.replace("module.", ""): value for key, value in loaded_state["model"].items()}
            self.load_state_dict(state)

        def update_to_step(self, step: int) -> None:
            pass

    class Field(nn.Module):
        def __init__(self) -> None:
            super().__init__()
            self._sample_locations = None
            self._density_before_activation = None

        def density_fn(self, positions):
            raise NotImplementedError("density_fn serves the density-grid samplers of other model families; GF-NeRF samples with PersSampler")

        def get_density(self, ray_samples):
            raise NotImplementedError

        def get_normals(self):
            raise NotImplementedError("normals are not predicted on this path (predict_normals=False, gfnerf/config.py:106)")

        def get_outputs(self, ray_samples, density_embedding=None):
            raise NotImplementedError

        def forward(self, ray_samples, compute_normals: bool = False):
            density, density_embedding = self.get_density(ray_samples)
            field_outputs = self.get_outputs(ray_samples, density_embedding=density_embedding)
            field_outputs[FieldHeadNames.DENSITY] = density
            return field_outputs

    class Pipeline(nn.Module):
        datamanager: Any
        _model: Any

        @property
        def model(self):
            return self._model.module if hasattr(self._model, "module") and not isinstance(self._model, Model) else self._model

        @property
        def device(self):
            return self.model.device

        def load_state_dict(self, state_dict, strict: bool = True):
            model_state = {key[len("_model."):]: value for key, value in state_dict.items() if key.startswith("_model.")}
            pipeline_state = {key: value for key, value in state_dict.items() if not key.startswith("_model.")}
            self._model.load_state_dict(model_state, strict=strict)
            super().load_state_dict(pipeline_state, strict=False)

        def get_train_loss_dict(self, step: int):
            raise NotImplementedError

        def get_eval_loss_dict(self, step: int):
            raise NotImplementedError

        def get_eval_image_metrics_and_images(self, step: int):
            raise NotImplementedError

        def get_average_eval_image_metrics(self, step: Optional[int] = None):
            raise NotImplementedError

        def load_pipeline(self, loaded_state: Dict[str, Any], step: int) -> None:
            raise NotImplementedError

        def get_training_callbacks(self, training_callback_attributes: TrainingCallbackAttributes) -> List[TrainingCallback]:
            raise NotImplementedError

        def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
            raise NotImplementedError

    @dataclass
    class AdamOptimizerConfig(InstantiateConfig):
        _target: Type = torch.optim.Adam
        lr: float = 0.0005
        eps: float = 1e-08
        max_norm: Optional[float] = None
        weight_decay: float = 0

        def setup(self, params) -> Any:
            kwargs = {k: v for k, v in vars(self).items() if k not in ("_target", "max_norm")}
            return self._target(params, **kwargs)

    class Optimizers:
        """engine/optimizers.py:73-193 with the GF edits: a group missing from `config` takes config['fields'] (:86-87),
        delete_optimizer drops a group's optimizer, scheduler and parameters (:161-168), add_optimizer (re)creates them,
        optionally over new `params` at a new initial learning rate (:170-193)."""

        def __init__(self, config: Dict[str, Any], param_groups: Dict[str, List[nn.Parameter]]) -> None:
            self.config, self.param_groups = config, param_groups
            self.optimizers, self.schedulers, self.parameters = {}, {}, {}
            for name in list(param_groups):
                self._create(name, param_groups[name], None)

        def _create(self, name, params, lr_init):
            if name not in self.config:
                self.config[name] = self.config["fields"]
            if lr_init is None:
                lr_init = self.config[name]["optimizer"].lr
            else:
                self.config[name]["optimizer"].lr = lr_init
            self.optimizers[name] = self.config[name]["optimizer"].setup(params=params)
            self.parameters[name] = params
            if self.config[name]["scheduler"]:
                self.schedulers[name] = self.config[name]["scheduler"].setup().get_scheduler(optimizer=self.optimizers[name], lr_init=lr_init)

        def optimizer_step(self, param_group_name: str) -> None:
            self.optimizers[param_group_name].step()

        def scheduler_step(self, param_group_name: str) -> None:
            if param_group_name in self.schedulers:
                self.schedulers[param_group_name].step()

        def zero_grad_all(self) -> None:
            for optimizer in self.optimizers.values():
                optimizer.zero_grad()

        def optimizer_step_all(self) -> None:
            for name, optimizer in self.optimizers.items():
                max_norm = self.config[name]["optimizer"].max_norm
                if max_norm is not None:
                    torch.nn.utils.clip_grad_norm_(self.parameters[name], max_norm)
                optimizer.step()

        def optimizer_scaler_step_all(self, grad_scaler) -> None:
            for name, optimizer in self.optimizers.items():
                max_norm = self.config[name]["optimizer"].max_norm
                if max_norm is not None:
                    grad_scaler.unscale_(optimizer)
                    torch.nn.utils.clip_grad_norm_(self.parameters[name], max_norm)
                grad_scaler.step(optimizer)

        def scheduler_step_all(self, step: int) -> None:
            for scheduler in self.schedulers.values():
                scheduler.step()

        def delete_optimizer(self, param_group_name):
            if param_group_name in self.optimizers:
                del self.optimizers[param_group_name]
                self.schedulers.pop(param_group_name, None)
                del self.parameters[param_group_name]
                del self.param_groups[param_group_name]

        def add_optimizer(self, param_group_name_target, params=None, lr_init=None):
            if param_group_name_target in self.optimizers:
                return
            if params is not None and param_group_name_target not in self.param_groups:
                self.param_groups[param_group_name_target] = params
            if param_group_name_target in self.param_groups:
                self._create(param_group_name_target, self.param_groups[param_group_name_target], lr_init)

        def load_optimizers(self, loaded_state: Dict[str, Any]) -> None:
            for k, v in loaded_state.items():
                self.optimizers[k].load_state_dict(v)

    @dataclass
    class TrainerConfig:
        """The TrainerConfig fields gfnerf/config.py:52-148 sets (engine/trainer.py)."""
        method_name: Optional[str] = None
        steps_per_eval_batch: int = 500
        steps_per_save: int = 1000
        max_num_iterations: int = 1000000
        mixed_precision: bool = False
        pipeline: Any = None
        optimizers: Dict[str, Any] = field(default_factory=dict)
        viewer: Any = None
        vis: str = "wandb"

    @dataclass
    class MethodSpecification:
        config: TrainerConfig
        description: str

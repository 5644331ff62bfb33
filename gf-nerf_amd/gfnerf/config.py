"""gf-nerf method registration (gfnerf/config.py:43-148): `gfnerf_config` is the object the entry point
`nerfstudio.method_configs: gf-nerf = gfnerf.config:gfnerf_config` (pyproject.toml:174-176) hands to
`nerfstudio.plugins.registry.discover_methods`, which reads `spec.config.method_name`, `spec.config` and
`spec.description` (plugins/registry.py, plugins/types.py:23-33).  The reference's Nerfstudio tree is not installed in
this image, so `MethodSpecification` / `TrainerConfig` / the pipeline and datamanager configs are small stand-ins with
the same field names and the reference's values; `_target` of the pipeline and model configs are this package's mirror
classes (GFNerfPipeline, GFNeRFModel), the optimiser / scheduler entries carry the reference's numbers.  With the host
Nerfstudio tree present, INTEGRATION.md shows the three lines that put the real config classes around the same values."""
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple

from .nerfacto import GFNeRFModel, GFNeRFModelConfig
from .schedulers import GFNerfExponentialDecaySchedulerConfig

N_BLOCKS = 10
N_SPLIT_DATASET = 10
N_ACTIVE_BLOCK = 1
N_DATASET_CIRCLES = 1           # each sub-dataset is trained N_DATASET_CIRCLES times
STEPS_PERSSAMPLER_INIT = 30000  # global stage: steps during which UpdateOctNodes runs; -1 disables the stage
STEPS_PER_SPLIT_DATASET = 10000
SCALE_FACTOR = 10.0
use_appearance_embedding = True


@dataclass
class AdamOptimizerConfig:
    """engine/optimizers.py AdamOptimizerConfig: the values HotPath's Adam kernel takes."""
    lr: float = 1e-2
    eps: float = 1e-15
    weight_decay: float = 0.0


@dataclass
class GFNerfDataManagerConfig:
    """The fields of GFNerfDataManagerConfig (gfnerf/config.py:62-89) the batch-assembly path reads."""
    n_split_dataset: int = N_SPLIT_DATASET
    steps_per_split_dataset: int = STEPS_PER_SPLIT_DATASET
    steps_perssampler_init: int = STEPS_PERSSAMPLER_INIT
    scale_factor: float = SCALE_FACTOR      # dataparser.scale_factor
    patch_size: int = 1
    eval_image_indices: Tuple[int, ...] = (0,)
    train_num_rays_per_batch: int = 2048 * 4
    eval_num_rays_per_batch: int = 2048
    train_num_images_to_sample_from: int = 500
    train_num_times_to_repeat_images: int = 1000
    camera_optimizer_mode: str = "off"


@dataclass
class GFNerfPipelineConfig:
    """GFNerfPipelineConfig (gf_pipeline.py:65-74): `_target` is the pipeline class, `datamanager` / `model` its parts."""
    datamanager: GFNerfDataManagerConfig = field(default_factory=GFNerfDataManagerConfig)
    model: GFNeRFModelConfig = field(default_factory=GFNeRFModelConfig)
    _target: Any = None

    def __post_init__(self):
        if self._target is None:
            from .gf_pipeline import GFNerfPipeline

            self._target = GFNerfPipeline


@dataclass
class TrainerConfig:
    """The TrainerConfig fields gfnerf/config.py sets (engine/trainer.py)."""
    method_name: str = "gf-nerf"
    steps_per_eval_batch: int = 1000
    steps_per_save: int = 2000
    max_num_iterations: int = STEPS_PERSSAMPLER_INIT + N_DATASET_CIRCLES * STEPS_PER_SPLIT_DATASET * N_SPLIT_DATASET
    mixed_precision: bool = False
    pipeline: GFNerfPipelineConfig = field(default_factory=GFNerfPipelineConfig)
    optimizers: Dict[str, Dict[str, Any]] = field(default_factory=dict)
    vis: str = "viewer"


@dataclass
class MethodSpecification:
    """plugins/types.py:23-33."""
    config: TrainerConfig
    description: str


def _model_config() -> GFNeRFModelConfig:
    c = GFNeRFModelConfig(n_blocks=N_BLOCKS, n_split_dataset=N_SPLIT_DATASET, steps_per_split_dataset=STEPS_PER_SPLIT_DATASET,
                          steps_perssampler_init=STEPS_PERSSAMPLER_INIT, scale_factor=SCALE_FACTOR,
                          use_appearance_embedding=use_appearance_embedding, use_ch_loss=True, log2_hashmap_size=21,
                          hidden_dim=128, hidden_dim_color=128)
    c._target = GFNeRFModel  # GFNeRFModelConfig._target (nerfacto.py:75-79)
    return c


gfnerf_config = MethodSpecification(
    config=TrainerConfig(
        pipeline=GFNerfPipelineConfig(datamanager=GFNerfDataManagerConfig(), model=_model_config()),
        optimizers={
            "fields": {
                "optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15),
                "scheduler": GFNerfExponentialDecaySchedulerConfig(
                    lr_final=1e-4, max_steps=STEPS_PERSSAMPLER_INIT, n_split_dataset=N_SPLIT_DATASET,
                    n_dataset_circles=N_DATASET_CIRCLES, steps_per_split_dataset=STEPS_PER_SPLIT_DATASET,
                    steps_perssampler_init=STEPS_PERSSAMPLER_INIT),
            },
        },
    ),
    description="Nerfacto with depth supervision for multi cameras",
)


def discover_methods(entry_points) -> Tuple[Dict[str, TrainerConfig], Dict[str, str]]:
    """plugins/registry.py discover_methods over the given entry points (group nerfstudio.method_configs): load each,
    keep the MethodSpecification-shaped ones, -> (configs by method name, descriptions)."""
    methods: Dict[str, TrainerConfig] = {}
    descriptions: Dict[str, str] = {}
    for ep in entry_points:
        spec = ep.load()
        if not (hasattr(spec, "config") and hasattr(spec, "description") and hasattr(spec.config, "method_name")):
            continue  # the reference prints a warning and skips (registry.py)
        methods[spec.config.method_name] = spec.config
        descriptions[spec.config.method_name] = spec.description
    return methods, descriptions


def method_specification() -> Optional[MethodSpecification]:
    """The spec registered under the entry point (kept for callers of the earlier dict form)."""
    return gfnerf_config

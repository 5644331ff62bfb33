"""gf-nerf method registration (gfnerf/config.py:43-148): `gfnerf_config` is the object the entry point
`nerfstudio.method_configs: gf-nerf = gfnerf.config:gfnerf_config` (pyproject.toml:174-176) hands to
`nerfstudio.plugins.registry.discover_methods`, which keeps it only if `isinstance(spec, MethodSpecification)`
(plugins/registry.py:43) and then reads `spec.config.method_name`, `spec.config` and `spec.description`.

`MethodSpecification`, `TrainerConfig`, `AdamOptimizerConfig` come from `_host`: inside the reference's Nerfstudio tree they
are its own classes (so the isinstance check holds and `ns-train gf-nerf` resolves to this package's pipeline and model),
and the datamanager / scheduler / viewer configs are the tree's too; on a machine without Nerfstudio they are `_host`'s
stand-ins and the datamanager is the HBM-resident one of gf_pipeline.py.  Either way the VALUES are the reference's."""
from . import _host
from ._host import AdamOptimizerConfig, MethodSpecification, TrainerConfig
from .optim import GfnAdamOptimizerConfig
from .gf_pipeline import GFNerfPipelineConfig, HbmDataManagerConfig
from .nerfacto import GFNeRFModelConfig

N_BLOCKS = 10
N_SPLIT_DATASET = 10
N_ACTIVE_BLOCK = 1
N_DATASET_CIRCLES = 1           # each sub-dataset is trained N_DATASET_CIRCLES times
STEPS_PERSSAMPLER_INIT = 30000  # global stage: steps during which UpdateOctNodes runs; -1 disables the stage
STEPS_PER_SPLIT_DATASET = 10000
SCALE_FACTOR = 10.0
use_appearance_embedding = True


def _model_config() -> GFNeRFModelConfig:
    """gfnerf/config.py:88-127."""
    return GFNeRFModelConfig(
        n_blocks=N_BLOCKS, n_split_dataset=N_SPLIT_DATASET, steps_per_split_dataset=STEPS_PER_SPLIT_DATASET, n_activate_block=N_ACTIVE_BLOCK,
        steps_perssampler_init=STEPS_PERSSAMPLER_INIT, scale_factor=SCALE_FACTOR,
        s3im_loss_mult=1.0, s3im_kernel_size=4, s3im_stride=4, s3im_repeat_time=10, s3im_patch_height=32,
        use_appearance_embedding=use_appearance_embedding, mono_depth_loss_mult=0.00, semantic_loss_weight=0.0, patch_size=1,
        eval_num_rays_per_chunk=2048, predict_normals=False, use_normal_loss=False, use_depth_loss=False, use_ch_loss=True,
        use_perceptual_loss=False, disable_scene_contraction=False, use_semantics=False, num_semantic_classes=2, use_mask=False,
        log2_hashmap_size=21, num_levels=16, far_plane=50, num_nerf_samples_per_ray=256, num_proposal_samples_per_ray=(512, 256),
        hidden_dim=128, hidden_dim_color=128, hidden_dim_transient=128, max_res=4096, proposal_weights_anneal_max_num_iters=5000,
        background_color="black")


def _datamanager_config():
    """gfnerf/config.py:62-87: the host's GFNerfDataManagerConfig with the reference's values when the tree is there."""
    if _host.HOST == "nerfstudio":
        from nerfstudio.cameras.camera_optimizers import CameraOptimizerConfig
        from nerfstudio.data.datamanagers.base_datamanager import GFNerfDataManagerConfig
        from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig

        try:  # the reference's dataparser stays where it is (host-side I/O, SURVEY 2a OUT); stock Nerfstudio's otherwise
            from gfnerf.ori_dataparser import NerfstudioDataParserConfig
        except ImportError:
            from nerfstudio.data.dataparsers.nerfstudio_dataparser import NerfstudioDataParserConfig
        return GFNerfDataManagerConfig(
            n_split_dataset=N_SPLIT_DATASET, steps_per_split_dataset=STEPS_PER_SPLIT_DATASET, steps_perssampler_init=STEPS_PERSSAMPLER_INIT,
            dataparser=NerfstudioDataParserConfig(scale_factor=SCALE_FACTOR, scene_scale=1.0, downscale_factor=1, orientation_method="vertical",
                                                  train_split_fraction=1.0),
            patch_size=1, eval_image_indices=(0,), train_num_rays_per_batch=2048 * 4, eval_num_rays_per_batch=2048,
            train_num_images_to_sample_from=500, train_num_times_to_repeat_images=1000, semantic_sample_weights=None,
            camera_optimizer=CameraOptimizerConfig(mode="off", optimizer=AdamOptimizerConfig(lr=6e-4, eps=1e-8, weight_decay=1e-2),
                                                   scheduler=ExponentialDecaySchedulerConfig(lr_final=6e-6, max_steps=30000)))
    return HbmDataManagerConfig(train_num_rays_per_batch=2048 * 4)


def _scheduler_config():
    """gfnerf/config.py:131-137."""
    if _host.HOST == "nerfstudio":
        from nerfstudio.engine.schedulers import GFNerfExponentialDecaySchedulerConfig
    else:
        from .schedulers import GFNerfExponentialDecaySchedulerConfig
    return GFNerfExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=STEPS_PERSSAMPLER_INIT, n_split_dataset=N_SPLIT_DATASET,
                                                 n_dataset_circles=N_DATASET_CIRCLES, steps_per_split_dataset=STEPS_PER_SPLIT_DATASET,
                                                 steps_perssampler_init=STEPS_PERSSAMPLER_INIT)


def _viewer_config():
    if _host.HOST == "nerfstudio":
        from nerfstudio.configs.base_config import ViewerConfig

        return ViewerConfig(num_rays_per_chunk=1 << 8)
    return None


gfnerf_config = MethodSpecification(
    TrainerConfig(
        method_name="gf-nerf",
        steps_per_eval_batch=1000,
        steps_per_save=2000,
        max_num_iterations=STEPS_PERSSAMPLER_INIT + N_DATASET_CIRCLES * STEPS_PER_SPLIT_DATASET * N_SPLIT_DATASET,
        mixed_precision=False,
        pipeline=GFNerfPipelineConfig(datamanager=_datamanager_config(), model=_model_config()),
        # torch.optim.Adam's update on the HIP Adam kernels, reading the gradients where the HIP backward leaves them (optim.py)
        optimizers={"fields": {"optimizer": GfnAdamOptimizerConfig(lr=1e-2, eps=1e-15), "scheduler": _scheduler_config()}},
        **({"viewer": _viewer_config()} if _viewer_config() is not None else {}),
        vis="viewer",
    ),
    description="Nerfacto with depth supervision for multi cameras",
)

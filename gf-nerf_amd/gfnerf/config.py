"""gf-nerf method constants (gfnerf/config.py:43-148).  Without nerfstudio in the image the MethodSpecification /
TrainerConfig objects cannot be instantiated; `gfnerf_config` is therefore a plain dict with the same values, and
`method_specification()` builds the real objects when nerfstudio is importable (entry point
`nerfstudio.method_configs: gf-nerf = gfnerf.config:gfnerf_config`, pyproject.toml:174-176)."""
N_BLOCKS = 10
N_SPLIT_DATASET = 10
N_ACTIVE_BLOCK = 1
N_DATASET_CIRCLES = 1
STEPS_PERSSAMPLER_INIT = 30000
STEPS_PER_SPLIT_DATASET = 10000
SCALE_FACTOR = 10.0
use_appearance_embedding = True

gfnerf_config = dict(
    method_name="gf-nerf",
    max_num_iterations=STEPS_PERSSAMPLER_INIT + N_DATASET_CIRCLES * STEPS_PER_SPLIT_DATASET * N_SPLIT_DATASET,
    mixed_precision=False,
    train_num_rays_per_batch=2048 * 4, eval_num_rays_per_batch=2048, eval_num_rays_per_chunk=2048,
    model=dict(n_blocks=N_BLOCKS, n_split_dataset=N_SPLIT_DATASET, steps_per_split_dataset=STEPS_PER_SPLIT_DATASET,
               n_activate_block=N_ACTIVE_BLOCK, steps_perssampler_init=STEPS_PERSSAMPLER_INIT, scale_factor=SCALE_FACTOR,
               use_appearance_embedding=use_appearance_embedding, use_ch_loss=True, s3im_loss_mult=1.0, log2_hashmap_size=21,
               num_levels=16, hidden_dim=128, hidden_dim_color=128, background_color="black"),
    optimizers=dict(fields=dict(optimizer=dict(lr=1e-2, eps=1e-15), scheduler=dict(lr_final=1e-4, max_steps=STEPS_PERSSAMPLER_INIT))),
    description="Nerfacto with depth supervision for multi cameras",
)


def method_specification():
    """The reference's MethodSpecification, when nerfstudio (the reference's modified copy) is importable."""
    from nerfstudio.plugins.types import MethodSpecification  # noqa: F401  (raises ImportError in this image)

    raise NotImplementedError("wire gfnerf_config into TrainerConfig/GFNerfPipelineConfig of the host nerfstudio tree; see INTEGRATION.md")

"""Host-side mirror of the reference's `gfnerf` package for the per-ray hot path: same class names, method
names, argument meaning and state-dict keys as gfnerf/{hash_3d_anchored,perssampler,mlp,nerfacto_field,nerfacto}.py,
implemented on the C ABI (gfnerf_amd.ops) instead of torch.classes.my_classes.* / tinycudann."""

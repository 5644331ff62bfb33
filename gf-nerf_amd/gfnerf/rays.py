"""Minimal stand-ins for the Nerfstudio dataclasses the hot path touches (this image has no nerfstudio):
RayBundle, WarpedSamples, RaySamples (nerfstudio/cameras/rays.py:108-200, GF-NeRF's modified copy).  Field names
match, so code written against the reference's objects reads the same."""
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class RayBundle:
    origins: torch.Tensor                      # [R,3]
    directions: torch.Tensor                   # [R,3]
    camera_indices: Optional[torch.Tensor] = None      # [R,1]
    rel_camera_indices: Optional[torch.Tensor] = None  # [R,1]
    steps: Optional[torch.Tensor] = None               # [R,1]
    pixel_area: Optional[torch.Tensor] = None
    lookat_directions: Optional[torch.Tensor] = None

    @property
    def shape(self):
        return tuple(self.origins.shape[:-1])

    def __len__(self):
        n = 1
        for d in self.origins.shape[:-1]:
            n *= int(d)
        return n

    def _map(self, fn):
        return RayBundle(**{k: (fn(v) if isinstance(v, torch.Tensor) else v) for k, v in self.__dict__.items()})

    def flatten(self):
        """[..., C] -> [N, C] for every tensor field (TensorDataclass.flatten)."""
        return self._map(lambda v: v.reshape(-1, v.shape[-1]))

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int):
        """rays.py RayBundle.get_row_major_sliced_ray_bundle: rays [start, end) of the flattened (row-major) image."""
        return self.flatten()._map(lambda v: v[start_idx:end_idx])


@dataclass
class WarpedSamples:  # rays.py:108-117 -- dense [R,1024,.] like PersSampler.GetSamples returns them
    sampled_world_pts: torch.Tensor
    sampled_pts: torch.Tensor
    sampled_dirs: torch.Tensor
    sampled_dists: torch.Tensor
    sampled_t: torch.Tensor
    sampled_anchors: torch.Tensor
    pts_idx_start_end: torch.Tensor
    first_oct_dis: torch.Tensor


@dataclass
class RaySamples:
    f2samples: Optional[WarpedSamples] = None
    camera_indices: Optional[torch.Tensor] = None
    rel_camera_indices: Optional[torch.Tensor] = None
    deltas: Optional[torch.Tensor] = None
    cur_step: int = -1
    cur_split_dataset_idx: int = -1
    packed: Optional[dict] = None  # the packed sampler output the HIP path actually consumes

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

    def __len__(self):
        return self.origins.shape[0]


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

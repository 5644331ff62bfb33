"""Shared test helpers: small synthetic scene + packed/dense conversions."""
import functools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


@functools.lru_cache(maxsize=None)
def small_scene(n_side=6, max_depth=9, seed=7):
    """A few hundred nodes; builds in a few seconds with the product's host C++ builder."""
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import scene

    return scene.build_scene(n_side=n_side, max_depth=max_depth, seed=seed)


def rays_for(sc, n_rays, seed=0):
    from gfnerf_amd import scene

    return scene.make_rays(sc["c2w"], n_rays, seed)


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False

"""gfnerf_amd -- MI355X-native (gfx950) implementation of the GF-NeRF per-ray hot path.

Layout
  csrc/        hand-written HIP kernels + the C ABI declared in include/gfnerf_amd.h
  _lib.py      ctypes loader for libgfnerf_amd.so (fails loudly when it is missing)
  ops.py       torch-tensor front-end of the C ABI (device pointers + current stream)
  gfnerf/      host-side mirror of the reference's plugin surface (Hash3DAnchored,
               PersSampler, MLPNetwork, GFNeRFField, GFNeRFModel, GFNerfPipeline)
  scene.py     synthetic Rubble-scale scene used by tests and bench.py

There is no CPU fallback: every op needs the HIP extension and a gfx950 device.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (does not load the .so until first use)

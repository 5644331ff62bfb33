#!/usr/bin/env python3
"""Fixture for gfnerf_amd.procedural (the analytic terrain bench.py trains on to report PSNR): 256 rays of the synthetic capture,
their ground-truth colours computed HERE in numpy float64 from the closed form and composited by the oracle's restatement of
get_weights_f2nerf + RGBRenderer (oracle.composite_fwd; nerfstudio/cameras/rays.py:178-200, renderers.py:69-138).
    python tests/golden/make_procedural.py   ->  tests/golden/procedural_gt.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfnerf_amd  # noqa: E402,F401
import oracle as O  # noqa: E402
from gfnerf_amd import scene  # noqa: E402

Z_TOP, Z_BOTTOM, N_QUAD = 4.0, -4.0, 384


def height(x, y):
    return 1.5 * np.sin(0.35 * x) * np.cos(0.27 * y) + 0.8 * np.sin(0.9 * x + 1.3) * np.sin(0.7 * y + 0.4)


TEX_WAVELENGTHS = (1.6, 0.8, 0.4, 0.2, 0.1)
DETAIL = 0.5  # the amplitude the second set of fixture colours is made with


def texture(x, y, z, c):
    v, norm = 0.0, 0.0
    for k, lam in enumerate(TEX_WAVELENGTHS):
        f = 2.0 * np.pi / lam
        th, ph, ps = 0.7 + 2.399963 * (3 * k + c), 1.3 * (k + 1) * (c + 1), 0.9 * (k + 2) * (c + 1)
        a = np.sqrt(lam / TEX_WAVELENGTHS[0])
        v = v + a * np.sin(f * (x * np.cos(th) + y * np.sin(th)) + ph) * np.cos(0.5 * f * z + ps)
        norm += a
    return v / norm


def field(xyz, detail=0.0):
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    h = height(x, y)
    sigma = 6.0 / (1.0 + np.exp(-4.0 * (h - z)))
    rgb = np.stack([0.5 + 0.5 * np.sin(0.8 * x + 0.3 * h), 0.5 + 0.5 * np.sin(0.6 * y + 1.0 + 0.5 * h), 0.5 + 0.5 * np.cos(0.4 * (x + y)) * np.cos(1.1 * h)], -1)
    if detail > 0:
        rgb = np.clip(rgb + detail * np.stack([texture(x, y, z, c) for c in range(3)], -1), 0.0, 1.0)
    return sigma, rgb


def main():
    c2w, _, _ = scene.make_cameras(n_side=40, seed=1234)
    o, d, cam = scene.make_rays(c2w, 256, seed=424242)
    o64, d64 = o.astype(np.float64), d.astype(np.float64)
    dn = d64 / np.linalg.norm(d64, axis=-1, keepdims=True)
    dz = np.minimum(dn[:, 2], -1e-3)
    t0 = np.maximum((Z_TOP - o64[:, 2]) / dz, 0.0)
    t1 = (Z_BOTTOM - o64[:, 2]) / dz
    step = (t1 - t0) / N_QUAD
    t = t0[:, None] + step[:, None] * (np.arange(N_QUAD) + 0.5)[None, :]
    xyz = o64[:, None, :] + t[..., None] * dn[:, None, :]
    sigma, rgb = field(xyz)
    R = o.shape[0]
    off = (np.arange(R + 1) * N_QUAD).astype(np.int64)
    comp = O.composite_fwd(off, sigma.reshape(-1).astype(np.float32), np.repeat(step, N_QUAD).astype(np.float32), t.reshape(-1).astype(np.float32),
                           rgb.reshape(-1, 3).astype(np.float32))
    _, rgb_d = field(xyz, DETAIL)
    comp_d = O.composite_fwd(off, sigma.reshape(-1).astype(np.float32), np.repeat(step, N_QUAD).astype(np.float32), t.reshape(-1).astype(np.float32),
                             rgb_d.reshape(-1, 3).astype(np.float32))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "procedural_gt.npz")
    np.savez_compressed(out, rays_o=o, rays_d=d, rgb=np.asarray(comp["rgb"], np.float32), acc=np.asarray(comp["acc"], np.float32).reshape(-1),
                        detail=np.float32(DETAIL), rgb_detail=np.asarray(comp_d["rgb"], np.float32), rgb_detail_probe=rgb_d[:4, ::64].astype(np.float32),
                        sigma_probe=sigma[:4, ::64].astype(np.float32), rgb_probe=rgb[:4, ::64].astype(np.float32), xyz_probe=xyz[:4, ::64].astype(np.float32))
    print(out, comp["rgb"].shape, float(np.asarray(comp["acc"]).min()), float(np.asarray(comp["acc"]).max()))


if __name__ == "__main__":
    main()

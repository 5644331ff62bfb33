"""Synthetic "Rubble-scale" scene for tests and bench.py.

An aerial capture: a serpentine grid of pinhole cameras above a ground plane, looking
down with random tilt/yaw, world units already multiplied by the reference's
scale_factor=10 (gfnerf/config.py:49) so the extent fits bbox_side_len = 2^(10-1) = 512
(gfnerf/nerfacto.py:223 bbox_levels=10, PersSampler.cpp:921).  Everything is seeded numpy;
no dataset and no reference code is involved.
"""
import math
import os

import numpy as np

WIDTH, HEIGHT = 1152, 864
FOCAL = 1000.0
BBOX_SIDE = 512.0
BOUNDS = (0.01, 512.0)  # gf_pipeline.py:117-119
SCALE_FACTOR = 10.0     # config.py:49


def _rot(axis, ang):
    c, s = math.cos(ang), math.sin(ang)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)
    if axis == 1:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float64)


def make_cameras(n_side=20, extent=40.0, height=12.0, tilt_deg=15.0, seed=1234):
    """-> c2w f32 [n,3,4] (OpenGL: camera looks along its -z), intri f32 [n,3,3], bounds f32 [n,2]."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-extent, extent, n_side)
    c2w = []
    for iy, y in enumerate(xs):
        row = xs if iy % 2 == 0 else xs[::-1]  # serpentine flight path
        for x in row:
            pos = np.array([x, y, height]) + rng.normal(0, 0.3, 3)
            R = _rot(2, rng.uniform(0, 2 * math.pi)) @ _rot(0, math.radians(rng.uniform(-tilt_deg, tilt_deg))) @ _rot(
                1, math.radians(rng.uniform(-tilt_deg, tilt_deg)))
            c2w.append(np.concatenate([R, pos[:, None]], axis=1))
    c2w = np.asarray(c2w, np.float32)
    n = c2w.shape[0]
    K = np.array([[FOCAL, 0, WIDTH / 2], [0, FOCAL, HEIGHT / 2], [0, 0, 1]], np.float32)
    intri = np.repeat(K[None], n, 0)
    bounds = np.repeat(np.array([BOUNDS], np.float32), n, 0)
    return c2w, intri, bounds


def make_rays(c2w, n_rays, seed=0, cams=None):
    """Uniformly random (camera, pixel) rays, nerfstudio pinhole convention
    (cameras.py generate_rays: d = R [(x-cx)/fx, -(y-cy)/fy, -1], normalised).
    cams: draw only from these camera indices (a focal block's camera cluster); the returned camera ids are global."""
    rng = np.random.default_rng(seed)
    if cams is None:
        cam = rng.integers(0, c2w.shape[0], n_rays)
    else:
        cams = np.asarray(cams, np.int64)
        cam = cams[rng.integers(0, cams.size, n_rays)]
    px = rng.uniform(0, WIDTH, n_rays)
    py = rng.uniform(0, HEIGHT, n_rays)
    d_cam = np.stack([(px - WIDTH / 2) / FOCAL, -(py - HEIGHT / 2) / FOCAL, -np.ones(n_rays)], -1)
    d = np.einsum("nij,nj->ni", c2w[cam, :, :3].astype(np.float64), d_cam)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = c2w[cam, :, 3]
    return o.astype(np.float32), d.astype(np.float32), cam.astype(np.int32)


def make_noise(n_rays, train, fineness, seed=1):
    """rays_noise of PersSampler_cuda.cu:380-389: ones (eval) or U[0.5,1.5) (train), times fineness."""
    n = 1024 + n_rays + 10
    if train:
        v = np.random.default_rng(seed).uniform(0.5, 1.5, n)
    else:
        v = np.ones(n)
    return (v * fineness).astype(np.float32)


def make_primes(n_volumes, seed=2):
    """prim_pool i32 [16, V, 3]: primes in [2^28, 2^30) (Hash3DAnchored.cpp:39-55), drawn from a
    seeded window of a segmented sieve instead of the reference's rejection sampling."""
    need = 16 * n_volumes * 3
    rng = np.random.default_rng(seed)
    lo_all, hi_all = 1 << 28, 1 << 30
    span = max(1 << 22, int(need * 24))
    span = min(span, hi_all - lo_all)
    lo = int(rng.integers(lo_all, hi_all - span + 1)) | 1
    sieve = np.ones(span // 2, dtype=bool)  # odd numbers lo, lo+2, ...
    limit = int(math.isqrt(lo + span)) + 1
    small = np.ones(limit + 1, dtype=bool)
    small[:2] = False
    for p in range(2, int(math.isqrt(limit)) + 1):
        if small[p]:
            small[p * p:: p] = False
    for p in np.nonzero(small)[0][1:]:  # skip 2
        p = int(p)
        start = ((lo + p - 1) // p) * p
        if start % 2 == 0:
            start += p
        sieve[(start - lo) // 2:: p] = False
    primes = lo + 2 * np.nonzero(sieve)[0]
    assert primes.size >= need, "prime window too small"
    pick = rng.choice(primes.size, need, replace=False)
    return primes[pick].astype(np.int32).reshape(16, n_volumes, 3)


def make_feat_pool(log2_T, seed=2, scale=1e-2):
    """feat_pool f32 [16*T, 2] ~ U(-scale, scale) (Hash3DAnchored::Reset, Hash3DAnchored.cpp:171-174)."""
    rng = np.random.default_rng(seed)
    return rng.uniform(-scale, scale, (16 * (1 << log2_T), 2)).astype(np.float32)


def linear_init(out_dim, in_dim, rng):
    """torch.nn.Linear default init: U(-1/sqrt(in), 1/sqrt(in)) for weight and bias."""
    k = 1.0 / math.sqrt(in_dim)
    return rng.uniform(-k, k, (out_dim, in_dim)).astype(np.float32), rng.uniform(-k, k, out_dim).astype(np.float32)


def make_mlps(seed=4, hidden=128):
    """base_network 32->128->16 and mlp_head 63->128->128->3 (nerfacto_field.py:174-179,217-227)."""
    rng = np.random.default_rng(seed)
    base = [linear_init(hidden, 32, rng), linear_init(16, hidden, rng)]
    head = [linear_init(hidden, 63, rng), linear_init(hidden, hidden, rng), linear_init(3, hidden, rng)]
    return base, head


def build_scene(n_side=20, max_depth=16, seed=1234, cache_dir=None, threads=0):
    """Cameras + octree (built by the product's host C++ builder).  Returns a dict of numpy arrays.
    With cache_dir the blobs are cached on disk (the build is deterministic in its arguments)."""
    from . import ops

    c2w, intri, bounds = make_cameras(n_side=n_side, seed=seed)
    key = f"scene_v2_s{n_side}_d{max_depth}_seed{seed}.npz"  # v2: per-leaf random streams
    path = os.path.join(cache_dir, key) if cache_dir else None
    nodes = None
    if path and os.path.exists(path):
        try:
            z = np.load(path)
            nodes, transes = z["nodes"], z["transes"]
        except Exception:  # a torn or foreign file: rebuild
            nodes = None
    if nodes is None:
        nodes, transes = ops.octree_build(c2w, intri, bounds, max_depth=max_depth, bbox_side_len=BBOX_SIDE, seed=seed,
                                          threads=threads)
        if path:  # several ranks of one node may get here at once: write privately, publish atomically
            os.makedirs(cache_dir, exist_ok=True)
            tmp = f"{path}.{os.getpid()}.tmp.npz"
            np.savez(tmp, nodes=nodes, transes=transes)
            os.replace(tmp, path)
    return dict(c2w=c2w, intri=intri, bounds=bounds, nodes=nodes, transes=transes,
                n_nodes=nodes.size // 128, n_trans=transes.size // 576)

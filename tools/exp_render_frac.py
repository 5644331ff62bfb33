"""Scratch: what fraction of a render tile's samples the fused forward evaluates (early termination at T < 1e-4), and the
samples per ray of the eval-mode march -- the head-room a depth-chunked level-major renderer would have."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd
from gfnerf_amd import hotpath, scene, ops
cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene")
sc = scene.build_scene(n_side=20, max_depth=16, seed=1234, cache_dir=cache)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=21), seed=0, n_blocks=1)
hp.blocks[0].uniform_(-1e-3, 1e-3); hp.refresh_shadows()
H, W = 1080, 1920
K = sc["intri"]; zoom = W / 1152.0
c2w = torch.from_numpy(sc["c2w"]).cuda()
intr = torch.from_numpy(np.stack([K[:, 0, 0] * zoom, K[:, 1, 1] * zoom, np.full(K.shape[0], W / 2), np.full(K.shape[0], H / 2)], -1).astype(np.float32)).cuda()
fineness = 6.0448
for cam in (0, 37, 74):
    ys, xs = torch.meshgrid(torch.arange(400, 464, device="cuda"), torch.arange(800, 928, device="cuda"), indexing="ij")
    coords = torch.stack([ys.reshape(-1), xs.reshape(-1)], 1).float() + 0.5
    ci = torch.full((coords.shape[0],), cam, dtype=torch.int32, device="cuda")
    g = ops.generate_rays(c2w, intr, ci, coords)
    o, d = g["origins"], g["directions"]
    noise = torch.from_numpy(scene.make_noise(o.shape[0], False, fineness)).cuda()
    out = hp.render(o, d, ci, noise, block=0)
    s = out["samples"]; n = int(s["pts_offsets"][-1].item())
    ev = int((out["field"]["sigma"][:n] > 0).sum().item())
    cnt = np.diff(s["pts_offsets"].cpu().numpy())
    w = out["weights"][:n]
    print(f"cam {cam}: rays {o.shape[0]}, samples/ray mean {cnt.mean():.1f} max {cnt.max()}, evaluated {ev / n:.3f}, weights>0 {(w > 0).float().mean().item():.3f}, acc {out['accumulation'].mean().item():.4f}")

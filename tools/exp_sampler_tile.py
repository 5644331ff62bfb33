"""Scratch: the sampler alone on a render tile (384 x 360 pixels = 138,240 rays): what do trace / march / fill cost when nothing
else is on the chip?  Run under rocprofv3 --kernel-trace --stats."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd  # noqa: F401
from gfnerf_amd import hotpath, ops, scene
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=21), seed=0, n_blocks=1)
H, W = 1080, 1920
K = sc["intri"]; zoom = W / 1152.0
c2w = torch.from_numpy(sc["c2w"]).cuda()
intr = torch.from_numpy(np.stack([K[:, 0, 0] * zoom, K[:, 1, 1] * zoom, np.full(K.shape[0], W / 2), np.full(K.shape[0], H / 2)], -1).astype(np.float32)).cuda()
rays = ops.generate_rays(c2w, intr, cam0=37, first_pixel=0, width=W, n=H * W, want_extras=False)
O = rays["origins"].view(H, W, 3)[360:720, 768:1152].reshape(-1, 3).contiguous()
D = rays["directions"].view(H, W, 3)[360:720, 768:1152].reshape(-1, 3).contiguous()
R = O.shape[0]
noise = torch.full((1024 + R + 10,), 7.5886, dtype=torch.float32, device="cuda")
for _ in range(6):
    s = hp.sample(O, D, noise)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    s = hp.sample(O, D, noise)
e1.record(); torch.cuda.synchronize()
print(f"{R} rays, {int(s['pts_offsets'][-1])} samples: sampler alone {e0.elapsed_time(e1) / 5:.3f} ms per tile")

"""Scratch: how many DISTINCT table cells does a group of 32 samples touch per level in a render tile -- (A) 32 consecutive
samples of one ray (what fused_fwd_kernel's waves hold today) against (B) the k-th samples of an 8 x 4 patch of neighbouring
pixels' rays?  Lanes of one gather instruction that hit the same cell of the same volume hit the same 8 slots: one line fill
instead of many.  Render tile = 128 x 64 pixels of a 1920 x 1080 frame, bench scene."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd  # noqa: F401
from gfnerf_amd import hotpath, ops, scene, _lib

n_side = int(os.environ.get("N_SIDE", "40"))
sc = scene.build_scene(n_side=n_side, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=21), seed=0, n_blocks=1)
H, W = 1080, 1920
K = sc["intri"]
zoom = W / 1152.0
c2w = torch.from_numpy(sc["c2w"]).cuda()
intr = torch.from_numpy(np.stack([K[:, 0, 0] * zoom, K[:, 1, 1] * zoom, np.full(K.shape[0], W / 2), np.full(K.shape[0], H / 2)], -1).astype(np.float32)).cuda()
fineness = float(os.environ.get("FINENESS", "7.5886"))
scales = np.zeros(16, dtype=np.float32)
_lib.lib().gfn_hash_level_scales_host(scales.ctypes.data)
for cam, (ty, tx) in ((37, (4, 7)), (74, (8, 3)), (111, (12, 11))):
    rays = ops.generate_rays(c2w, intr, cam0=cam, first_pixel=0, width=W, n=H * W, want_extras=False)
    O = rays["origins"].view(H, W, 3)[ty * 64:(ty + 1) * 64, tx * 128:(tx + 1) * 128].reshape(-1, 3).contiguous()
    D = rays["directions"].view(H, W, 3)[ty * 64:(ty + 1) * 64, tx * 128:(tx + 1) * 128].reshape(-1, 3).contiguous()
    R = O.shape[0]
    noise = torch.full((1024 + R + 10,), fineness, dtype=torch.float32, device="cuda")
    s = hp.sample(O, D, noise)
    off = s["pts_offsets"].long()
    n = int(off[-1].item())
    cnt = (off[1:] - off[:-1])
    ray_id = torch.repeat_interleave(torch.arange(R, device="cuda"), cnt)
    k = torch.arange(n, device="cuda") - off[ray_id]
    x = s["warped_pts"][:n]
    vol = s["anchors"][:n, 0].long()
    py, px = ray_id // 128, ray_id % 128
    patch = (py // 4) * 16 + (px // 8)                      # 8 x 4 pixel patches of the 128 x 64 tile
    gA = ray_id * 64 + k // 32                              # (A) ray, 32-sample chunk
    gB = patch * 2048 + k                                   # (B) patch, sample index
    gC = (py * 4 + px // 32) * 2048 + k                     # (C) 32 x 1 strip of a row, sample index
    print(f"camera {cam} tile ({ty},{tx}): {R} rays, {n} samples ({n / R:.1f} per ray)")
    tot = {"A": 0, "B": 0, "C": 0}
    for level in range(16):
        c = torch.floor(x * float(scales[level])).clamp_(min=0).long()  # the sampler's warped_pts are in table space already
        key = ((vol * 2048 + c[:, 0]) * 2048 + c[:, 1]) * 2048 + c[:, 2]
        row = [f"  level {level:2d} (res {scales[level]:7.1f})"]
        for name, g in (("A", gA), ("B", gB), ("C", gC)):
            pairs = torch.stack([g, key], 1)
            distinct = torch.unique(pairs, dim=0).shape[0]
            groups = torch.unique(g).shape[0]
            tot[name] += distinct
            row.append(f"{name}: {distinct / n:5.3f} cells/sample ({distinct / groups:5.1f} per group of {n / groups:4.1f})")
        print("  ".join(row))
    print("  all levels, distinct cells per sample:", {k_: round(v / n / 16, 3) for k_, v in tot.items()})

# lockstep cost of the patch kernel: a wave runs until the LAST of its 32 rays has ended or become opaque
hp.table.mul_(100.0); hp.blocks[0].normal_(0, 0.01); hp.refresh_shadows()
cam_t = torch.full((R,), 37, dtype=torch.int32, device="cuda")
out = hp.render(O, D, cam_t, noise, block=0, early_T=1e-4)
w = out["weights"][:n]
alive = torch.zeros(R, device="cuda").index_add_(0, ray_id, (w > 0).float())
for name, pid in (("8x4", patch), ("32x1", py * 4 + px // 32), ("4x8", (py // 8) * 32 + px // 4)):
    npatch = int(pid.max().item()) + 1
    # per-patch max over its rays (scatter-reduce through sorting: ray -> patch is constant per ray)
    ray_patch = torch.zeros(R, dtype=torch.long, device="cuda").index_copy_(0, ray_id, pid)
    mx = torch.zeros(npatch, device="cuda").scatter_reduce_(0, ray_patch, alive, reduce="amax")
    print(f"patch {name}: mean alive samples per ray {alive.mean().item():.1f} of {n / R:.1f}; mean over patches of the longest ray {mx.mean().item():.1f} "
          f"-> lockstep efficiency {alive.mean().item() / mx.mean().item():.2f}")

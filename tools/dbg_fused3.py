import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import gfnerf_amd
from gfnerf_amd import hotpath, scene, ops
from helpers import small_scene, rays_for
import oracle as O
sc = small_scene()
cfg = hotpath.HotPathConfig(log2_T=14, early_T=0.0, grad_f16=False, keep_features=True)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
rng = np.random.default_rng(11)
hp.blocks[0].copy_(torch.from_numpy(rng.uniform(-1e-2, 1e-2, tuple(hp.blocks[0].shape)).astype(np.float32)))
hp.table.mul_(100.0); hp.layers[1][0].mul_(4.0); hp.refresh_shadows()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
R = 384
o, d, cam = rays_for(sc, R, seed=3)
noise = t(scene.make_noise(R, True, 5.0, seed=2))
gt = rng.random((R, 3)).astype(np.float32)
# what the fused-decoder test does first
for prec in ("x3", "f32"):
    hp.cfg.mlp_precision = prec
    s = hp.sample(t(o), t(d), noise)
    fu = hp._field_forward(s, t(cam), block=0, fused=True, save=True)
    un = hp._field_forward(s, t(cam), block=0, fused=False)
hp.cfg.mlp_precision = "x3"
s = hp.sample(t(o), t(d), noise)
N = int(s["pts_offsets"][-1])
P = hp.export_params()
f, comp = hp._forward_fused(s, t(cam), None, 0.0, save=True)
sep = hp._field_forward(s, t(cam), block=None, fused=True, save=True)
a, b = f["feat"].cpu().numpy()[:N], sep["feat"].cpu().numpy()[:N]
x = s["warped_pts"].cpu().numpy()[:N]; an = s["anchors"].cpu().numpy()[:N, 0].astype(np.int64)
ref = O.hash_fwd(P["table"], P["prim"], P["bias"], x, an, P["prim"].shape[1], 14)
bad = np.argwhere(a != b)
print("mismatches", len(bad), "fused-vs-oracle", (a != ref).sum(), "separate-vs-oracle", (b != ref).sum())
for i, j in bad[:6]:
    print(i, j, "fused", a[i, j], "sep", b[i, j], "oracle", ref[i, j], "vol", an[i], "x", x[i])
t16 = hp.table16.clone()
hp.refresh_shadows()
print("table16 changed by refresh:", int((t16 != hp.table16).sum()))
outs = []
for k in range(4):
    f, comp = hp._forward_fused(s, t(cam), None, 0.0, save=True)
    outs.append(f["feat"].cpu().numpy()[:N].copy())
    bad = np.argwhere(outs[-1] != ref)
    print("run", k, "mismatch vs oracle", len(bad), [(int(i), int(j)) for i, j in bad[:4]])
ray_id = s["ray_id"].cpu().numpy()[:N]; po = s["pts_offsets"].cpu().numpy()
for i, j in bad[:3]:
    r = ray_id[i]; print("sample", i, "ray", r, "local idx", i - po[r], "ray len", po[r + 1] - po[r], "level", j // 2, "vol", an[i], "n_vol", P["prim"].shape[1])
refb = O.hash_fwd(P["blocks"][0], P["block_prims"][0], P["bias"], x, an, P["prim"].shape[1], 14)
for k in range(6):
    f, comp = hp._forward_fused(s, t(cam), None, 0.0, save=True)
    a = f["feat"].cpu().numpy()[:N]
    sg = f["sigma"].cpu().numpy()[:N]; sg2 = sep["sigma"].cpu().numpy()[:N]
    bad = np.argwhere(a != ref)
    if len(bad) == 0:
        continue
    print("run", k, "feat mismatches", len(bad), "sigma mismatches", int((sg != sg2).sum()))
    for i, j in bad[:6:2]:
        hits = [(d_, jj) for d_ in range(-64, 65) for jj in range(32) if 0 <= i + d_ < N and a[i, j] == ref[i + d_, jj]]
        print("  ", i, j, a[i, j], "ref", ref[i, j], "dual?", a[i, j] == np.float32(ref[i, j] + refb[i, j]), "matches ref at (dsample, col):", hits[:5])
    break

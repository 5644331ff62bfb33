"""Round 6 report ("as a report only"): the inference decoder with the ACTIVATION operands of some of its layers as one f16 number each
(GFN_RENDER_X2=<layer mask>, csrc/x3_device.h X2_*: 1 base layer 1 (hash features), 2 base layer 2, 4 head layer 1 (geo | SH), 8 head layer 2, 16 head
layer 3; 31 = all -- two MFMAs per product sum instead of three, no lo' halves to make) against the split-f16 x3 form (0) -- full-HD frames of the
random-init scene and of engines trained on the procedural terrain, THE SAME ENGINE rendered every way in one process (the switch is read per call):
ms/Mpix of each form and max / mean |rgb_mask - rgb_x3| over the frame (the x3 frame is within 2e-6 of the oracle on the pixels
tests/test_gpu_bench_workload.py::test_bench_scene_full_frame_tile_render checks, so this IS the distance to the oracle to 2e-6), and whether accumulation
and depth moved at all (masks 8, 16, 24 touch nothing the density depends on).  The march's fineness is the one that gives 192 samples per ray on the
engine at hand, as in bench.py.    python tools/exp_render_x2.py [masks, comma separated] [trained seeds, comma separated]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd  # noqa: F401
from gfnerf_amd import hotpath, ops, procedural, render as rmod, scene

MASKS = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "24,28,31,8,16,4,2,1").split(",")]
SEEDS = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")]
SWITCH = "GFN_RENDER_X2"
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
H, W = 1080, 1920
c2w = torch.from_numpy(sc["c2w"]).cuda()
K = sc["intri"]
zoom = W / 1152.0
intr = torch.from_numpy(np.stack([K[:, 0, 0] * zoom, K[:, 1, 1] * zoom, np.full(K.shape[0], W / 2), np.full(K.shape[0], H / 2)], -1).astype(np.float32)).cuda()


def fineness_for(hp, target=192.0, n=4096):
    o, d, _ = procedural.device_rays(c2w, n, torch.Generator(device="cuda").manual_seed(77))
    lo, hi = 0.5, 64.0
    for _ in range(16):
        fin = (lo * hi) ** 0.5
        mean = hp.sample(o, d, ops.sampler_noise(n, fin, train=False, device="cuda"))["pts_offsets"][-1].item() / n
        if abs(mean - target) / target < 0.01:
            break
        lo, hi = (fin, hi) if mean > target else (lo, fin)
    return fin, mean


def engines():
    fog = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=21), seed=0, n_blocks=1)
    fog.blocks[0].uniform_(-1e-3, 1e-3, generator=torch.Generator(device="cuda").manual_seed(100)); fog.refresh_shadows()
    yield "fog", fog
    del fog
    for sd in SEEDS:
        r = procedural.train_and_eval(sc, seed=sd, steps_global=3000, steps_focal=1000, detail=procedural.STUDY_DETAIL, lr_final=procedural.STUDY_LR_FINAL, keep_engine=True,
                                      **procedural.LEGS["fast"])
        yield f"trained seed {sd} ({r['psnr']:.2f} dB)", r["engine"]
        del r


for name, hp in engines():
    fin, mean = fineness_for(hp)
    tr = rmod.TileRenderer(hp, c2w, intr, block=0, fineness=fin, colour_x2=False)
    res = {}
    for rep in (0, 1):  # (every form twice, interleaved: the second round is what is reported)
        for x in [0] + MASKS:
            os.environ[SWITCH] = str(x)
            tr.graph = None  # (a captured tile graph has its kernel baked in: captured again under the switch)
            out = tr.render_camera(37, H, W)
            img = {k: out[k].clone() for k in ("rgb", "accumulation", "depth")}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for cam in (74, 111, 148):
                tr.render_camera(cam, H, W)
            torch.cuda.synchronize()
            res[x] = ((time.perf_counter() - t0) * 1e3 / (3 * H * W / 1e6), img)
    os.environ.pop(SWITCH, None)
    print(f"{name}: fineness {fin:.3f}, {mean:.0f} samples per ray, mean rgb {float(res[0][1]['rgb'].mean()):.3f}; x3 (mask 0): {res[0][0]:.2f} ms/Mpix", flush=True)
    for x in MASKS:
        d = (res[x][1]["rgb"] - res[0][1]["rgb"]).abs().float()
        same = all(torch.equal(res[x][1][k], res[0][1][k]) for k in ("accumulation", "depth"))
        print(f"  mask {x:2d}: {res[x][0]:6.2f} ms/Mpix ({res[x][0] / res[0][0] - 1:+.1%})  |rgb - rgb_x3| max {float(d.max()):.2e} mean {float(d.mean()):.2e} "
              f"p99.9 {float(torch.quantile(d.flatten()[::7], 0.999)):.2e}  margin to 1e-3: x{1e-3 / max(float(d.max()), 1e-12):.1f}  accumulation and depth {'the same bits' if same else 'MOVED'}",
              flush=True)
    del tr, res
    torch.cuda.empty_cache()

"""Round 6 probe: the focal step's Adam on the block table (HBM-bound, 1 GB) BESIDE the lookup of the frozen global table for the next batch (bound by the
L2 -> L1 gather rate) -- what would a step gain that pipelined the two across its boundary?  Bench workload (1,600 cameras, 4096 x 192, T = 2^21):
each alone, back to back on one stream, and on two streams (the lookup on all CUs and capped).    python tools/exp_adam_beside_lookup.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd  # noqa: F401
from gfnerf_amd import hotpath, ops, scene

sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
cfg = hotpath.HotPathConfig(log2_T=21)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
hp.blocks[0].uniform_(-1e-3, 1e-3); hp.refresh_shadows()
t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
R = 4096
o, d, cam = scene.make_rays(sc["c2w"], R, seed=0)
s = hp.sample(t(o), t(d), t(scene.make_noise(R, True, 7.5886, seed=1)))
n = int(s["pts_offsets"][-1].item())
print("samples", n, flush=True)
V, T = hp.n_volumes, cfg.log2_T
feat = [None]


def lookup(cus=0):
    feat[0] = ops.hash_fwd_lm(hp.table16, hp.prim, hp.bias_arg, s["warped_pts"], s["anchors"], V, T, n_dev=s["n_dev"], halves_from=cfg.lm_halves_from,
                              split_log2=cfg.lm_split_log2, planar=True, out=feat[0], opts=ops.launch_opts(decoder_cus=cus) if cus else None)


p = hp.blocks[0]
g = torch.zeros(p.shape, dtype=torch.float16, device="cuda")
g.view(-1)[: 17 << T].normal_(0, 1e-2)
m, v = torch.zeros_like(p), torch.zeros_like(p)
tick = [0]


def adam():
    tick[0] += 1
    ops.adam_step_table(p, g, m, v, tick[0], lr=1e-4, eps=1e-15, shadow_f16=hp.blocks16[0], n=17 << T)


side = torch.cuda.Stream()


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def both(cus=0):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        adam()
    lookup(cus)
    main.wait_stream(side)


def both_adam_first_on_main(cus=0):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    adam()
    with torch.cuda.stream(side):
        lookup(cus)
    main.wait_stream(side)


print(f"lookup alone {timed(lookup):.3f} ms   adam alone {timed(adam):.3f} ms   back to back {timed(lambda: (adam(), lookup())):.3f} ms", flush=True)
for cus in (0, 240, 224, 192):
    print(f"two streams, lookup on {cus or 256} CUs: {timed(lambda: both(cus)):.3f} ms   (adam enqueued first: {timed(lambda: both_adam_first_on_main(cus)):.3f})", flush=True)

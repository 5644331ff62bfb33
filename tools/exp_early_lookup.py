"""Round 5 feasibility probe: would the NEXT step's lookup of the FROZEN global table (focal stage) overlap with this step's backward + accumulation + Adam
if it ran on the second stream behind the prefetched sampler?  The four kernels lean on different parts of the chip (the lookup on L2 -> L1 gathers, the
backward on scattered stores, Adam on HBM, the march on the vector ALU).  Times, on the bench's focal batch (4096 x 192, T = 2^21, random init):
  tail alone (backward + accumulate + Adam) | lookup alone | sampler alone | tail beside sampler (the product) | tail beside sampler + lookup | the three sums
    python tools/exp_early_lookup.py [cus_for_tail]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import hotpath, ops, scene, _lib  # noqa: E402

R, FIN = 4096, 7.5886
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
cfg = hotpath.HotPathConfig(log2_T=21)
hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=1)
hp.blocks[0].uniform_(-1e-3, 1e-3)
hp.refresh_shadows()
hp._adam_state("block0", hp.blocks[0])
hp._record_buffers()
t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
batches = []
for k in range(2):
    o, d, cam = scene.make_rays(sc["c2w"], R, seed=k)
    gt = np.random.default_rng(7 + k).random((R, 3)).astype(np.float32)
    batches.append(tuple(t(a) for a in (o, d, cam, gt)))
noise = t(scene.make_noise(R, True, FIN, seed=1))
side = ops.concurrent_stream(role="sampler")
main = torch.cuda.current_stream()
# a few plain steps so that K is planned and every buffer exists
for i in range(6):
    o, d, cam, gt = batches[i % 2]
    hp.train_step(o, d, cam, gt, noise, stage="focal", block=0)
torch.cuda.synchronize()
o, d, cam, gt = batches[0]
comp, ctx = hp.forward_train(o, d, cam, noise, "focal", 0, want_depth=False)
loss, g = ops.charbonnier(comp["rgb"], gt)
hp._join_aux()
s = ctx["samples"]
n = int(s["pts_offsets"][-1])
st = hp._phase_state
print("samples", n, "K", st.k.tolist(), "list A", st.stats.tolist(), "forward", hp.last_forward, flush=True)
sa = st.samples()  # phase A's list of the forward above
feat2 = torch.empty((16, s["ts"].shape[0], 2), dtype=torch.float32, device="cuda")
o2, d2 = batches[1][0], batches[1][1]
samp_out = [None]


def tail():
    hp.backward(ctx, g)
    hp.optimizer_step("focal", 0)


def lookup():
    ops.hash_fwd_lm(hp.table16, hp.prim, hp.bias_arg, s["warped_pts"], s["anchors"], hp.n_volumes, cfg.log2_T, out=feat2, n_dev=s["n_dev"],
                    halves_from=cfg.lm_halves_from, split_log2=cfg.lm_split_log2, planar=True, sample_list=sa)


def sampler():
    samp_out[0] = hp.sample(o2, d2, noise, out=samp_out[0])


def timed(fn_main, fn_side=None, reps=12, cus=0):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if fn_side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fn_side()
        # (the cap is the engine's policy for a backward with a prefetch pending -- HotPath._cu_cap -- not a process-wide setting since ABI 107)
        hp.cfg.prefetch_cus = hp.cfg.prefetch_cus_dense_eager = cus or 256
        hp._pre = (None, None, None, False, {}) if cus else None
        if fn_main is not None:
            fn_main()
        hp._pre = None
        if fn_side is not None:
            main.wait_stream(side)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


cus = int(sys.argv[1]) if len(sys.argv) > 1 else 160
sampler(); torch.cuda.synchronize()
r = dict(tail=timed(tail), lookup=timed(lookup), sampler=timed(sampler),
         tail_capped=timed(tail, cus=cus),
         tail_beside_sampler=timed(tail, sampler, cus=cus),
         tail_beside_lookup=timed(tail, lookup),
         tail_beside_lookup_capped=timed(tail, lookup, cus=cus),
         tail_beside_sampler_and_lookup=timed(tail, lambda: (sampler(), lookup()), cus=cus),
         tail_beside_sampler_and_lookup_uncapped=timed(tail, lambda: (sampler(), lookup())),
         sampler_then_lookup_alone=timed(None, lambda: (sampler(), lookup())))
for k, v in r.items():
    print(f"{k:42s} {v:.3f} ms", flush=True)
print(f"now: tail beside sampler + lookup afterwards = {r['tail_beside_sampler'] + r['lookup']:.3f} ms; early lookup = {r['tail_beside_sampler_and_lookup']:.3f} ms "
      f"(cap {cus}), {r['tail_beside_sampler_and_lookup_uncapped']:.3f} (no cap)")

#!/usr/bin/env python3
"""bench.py -- train rays/s of the GF-NeRF hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--stage focal|global]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full training pass of the hot path over one batch of 4096 synthetic rays:
PersSampler trace+march -> Hash3DAnchored (global + active focal table) -> base MLP -> density /
SH / appearance head input -> head MLP -> compositing -> Charbonnier + S3IM loss -> full backward ->
Adam.  Workload (BASELINE.json configs[2]): "GF-NeRF global+focal, 4 sub-encoder blocks on 1 GPU,
Rubble-scale synthetic cameras, 4096 rays x 192 samples".  With N > 1 (configs[3]) every rank
owns one focal block and its own rays (weak scaling); the reference freezes the global table
and the decoder in this stage, so there is no data-path collective; `--stage
global` exercises the ray-data-parallel global stage with an RCCL all-reduce of the gradients.

Prints ONE JSON line on rank 0 (see README / DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_RAYS = 4096
TARGET_SAMPLES = {"focal": 192, "global": 128}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--stage", choices=["focal", "global"], default="focal")
    ap.add_argument("--n-side", type=int, default=20, help="cameras per side of the synthetic capture grid")
    ap.add_argument("--log2-T", type=int, default=21)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-times", action="store_true", help="print per-stage event timings to stderr (adds event overhead)")
    ap.add_argument("--cpu-rays", type=int, default=2048, help="rays in the bounded cpu_baseline sample")
    ap.add_argument("--early-T", type=float, default=1e-4, help="early-termination transmittance (0 = none, the reference)")
    ap.add_argument("--no-oct-update", action="store_true", help="global stage: skip the per-step UpdateOctNodes (K5/K6)")
    ap.add_argument("--render-frames", type=int, default=2,
                    help="also time N full 1920x1080 frames through the hipGraph tile renderer (BASELINE configs[4]); 0 = skip")
    ap.add_argument("--lm-halves-from", type=int, default=6, help="level-major lookup: levels >= this are gathered one 4 MB address half at a time")
    ap.add_argument("--lm-global-after", type=int, default=None, help="global stage: level-major forward from this optimiser step on (-1: never; default: HotPathConfig)")
    ap.add_argument("--render-lm", action="store_true", help="render: level-major lookups + decoder + compositing kernels instead of the fused forward")
    ap.add_argument("--record-capacity", type=int, default=0, help="gradient record buffer in records (0: HotPathConfig default)")
    ap.add_argument("--async-adam", action="store_true",
                    help="focal stage: Adam of step i on a third stream beside the frozen global table's lookup of step i + 1 (about "
                         "-1.5 percent per step; off by default so that the roofline kernel is timed running alone)")
    ap.add_argument("--lm-global-forward", action="store_true", help="global stage: level-major lookup + decoder + compositing instead of the fused forward kernel")
    ap.add_argument("--atomic-backward", action="store_true", help="backward issues its table adds as global atomics instead of records accumulated in LDS")
    ap.add_argument("--unfused-backward", action="store_true", help="decoder backward and hash scatter as two kernels")
    ap.add_argument("--no-prefetch", action="store_true", help="do not run the next batch's sampler beside this step's backward")
    ap.add_argument("--prefetch-cus", type=int, default=192, help="CUs left to the fused backward while the prefetched sampler runs beside it")
    ap.add_argument("--render-pipeline", type=int, default=8,
                    help="tiles per hipGraph replay: their samplers run beside the fused forwards on a second graph branch (1 = one tile per replay)")
    ap.add_argument("--render-decoder-cus", type=int, default=252, help="CUs the fused forward may occupy while a sampler runs beside it")
    ap.add_argument("--render-tile-rays", type=int, default=8192, help="rays per render tile")
    ap.add_argument("--render-tile-width", type=int, default=128, help="render tiles are blocks of this width (0 = raster runs of 4096 pixels)")
    ap.add_argument("--mlp-precision", choices=["x3", "f32"], default="x3",
                    help="fused decoder arithmetic: x3 = split-f16 MFMA at f32 accuracy (default), f32 = exact f32 MFMA")
    ap.add_argument("--share-block-primes", action="store_true",
                    help="block tables reuse the global table's primes (enables the interleaved one-gather-per-corner image); "
                         "default: own primes per table like the reference")
    return ap.parse_args()


def pick_fineness(hp, o, d, target, scene, torch):
    """Bisect the ray-march fineness so the mean valid samples per ray hits the config's budget."""
    lo, hi = 0.5, 64.0
    R = o.shape[0]
    best = None
    for _ in range(14):
        mid = (lo * hi) ** 0.5
        noise = torch.from_numpy(scene.make_noise(R, True, mid, seed=1)).cuda()
        s = hp.sample(o, d, noise)
        mean = s["pts_offsets"][-1].item() / R
        best = (mid, mean)
        if abs(mean - target) / target < 0.005:
            break
        if mean > target:
            lo = mid
        else:
            hi = mid
    return best


def lm_issued_gathers(samples, n_live, halves_from, torch):
    """Gathers one lm_lookup_kernel launch issues for this batch: 8 per (sample, level), except that below level `halves_from`
    only the first lane of a run of consecutive samples in one cell of one volume gathers when a wave holds <= 40 such runs
    (csrc/lm_lookup.hip).  Waves are taken as aligned groups of 64 samples (the kernel's start at its workgroup's run)."""
    pts = samples["warped_pts"][:n_live]
    vol = samples["anchors"][:n_live, 0]
    total = 0
    lane0 = (torch.arange(n_live, device=pts.device) % 64) == 0
    pad = (-n_live) % 64
    for level in range(16):
        if level >= halves_from:
            total += 8 * n_live
            continue
        cell = torch.floor(pts * float(np.float32(2.0 ** (7.0 * level / 15.0 + 3.0)))).to(torch.int32)
        head = torch.ones(n_live, dtype=torch.bool, device=pts.device)
        head[1:] = (vol[1:] != vol[:-1]) | (cell[1:] != cell[:-1]).any(1)
        head |= lane0
        per_wave = torch.nn.functional.pad(head.to(torch.int32), (0, pad)).view(-1, 64).sum(1)
        shared = torch.repeat_interleave(per_wave <= 40, 64)[:n_live]
        total += 8 * int(torch.where(shared, head, torch.ones_like(head)).sum().item())
    return total


def cpu_baseline(P, batch, log2_T, fineness, n_rays, block, scene):
    """The oracle (a CPU port of the same path) on a bounded slice of the same batch, all host threads
    for the OpenMP-parallel pieces.  fwd + bwd, no optimiser step."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pipeline as opipe  # oracle: test infrastructure, used here only as the reported CPU baseline

    o, d, cam, gt = (a[:n_rays] for a in batch)
    noise = scene.make_noise(n_rays, True, fineness, seed=1)
    t0 = time.perf_counter()
    loss, grads, aux = opipe.loss_and_grads(P, o, d, cam, gt, noise, log2_T, block=block)
    dt = time.perf_counter() - t0
    n_samples = int(aux["samples"]["pts_offsets"][-1])
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()  # what OpenMP may run on
    return dict(value=n_rays / dt, unit="rays/s", cores=cores, kind="port",
                sample=f"{n_rays} rays ({n_samples} samples) of the same batch, oracle fwd+bwd (no Adam), {dt:.1f} s; "
                       "trace/march/hash-fwd/linear-fwd/composite run OpenMP on all cores, the backward pieces are scalar")


def cpu_reference_field():
    """BASELINE configs[0]: the reference's pure-PyTorch CPU field path (TorchNerfactoField, torch HashEncoding 2^19, SH, MLPs,
    cumsum compositing) at 1024 rays x 48 samples, through oracle/torch_field.py -- the restatement pinned to the reference's
    own classes by tests/golden (the reference cannot travel to the GPU box).  5 warm-up + 20 timed iterations."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_field as tf

    import torch

    prev = torch.get_num_threads()
    torch.set_num_threads(min(prev, 16))  # a one-GPU box's CPU share; 128 oversubscribed threads ran this 10x slower
    try:
        r = tf.time_config0(n_rays=1024, n_samples=48, warmup=5, iters=20, log2_hashmap_size=19)
    finally:
        torch.set_num_threads(prev)
    return dict(workload="Nerfstudio nerfacto field (torch HashEncoding 16 levels 16..1024, 2^19) on synthetic Lego-like rays, 1024 rays x 48 samples",
                fwd_ms=r["fwd_ms"], fwd_bwd_ms=r["fwd_bwd_ms"], train_rays_per_sec=1024 / (r["fwd_bwd_ms"] * 1e-3),
                infer_rays_per_sec=1024 / (r["fwd_ms"] * 1e-3), threads=r["threads"], kind="port of the reference's torch path, pinned by golden vectors")


def render_bench(hp, sc, args, fineness, torch):
    """BASELINE configs[4]: full-res camera-path pass, 1920x1080, hipGraph-captured per-tile inference (eval-mode ray
    march: noise = ones * fineness; same fineness as the training batches).  Untimed: graph capture + one warm-up frame."""
    from gfnerf_amd import render as rmod

    H, W = 1080, 1920
    c2w = torch.from_numpy(sc["c2w"]).cuda()
    K = sc["intri"]
    zoom = W / 1152.0  # the synthetic capture's 1152-wide intrinsics scaled to the output width
    intr = torch.from_numpy(np.stack([K[:, 0, 0] * zoom, K[:, 1, 1] * zoom, np.full(K.shape[0], W / 2), np.full(K.shape[0], H / 2)], -1)
                            .astype(np.float32)).cuda()
    block = 0 if (args.stage == "focal" and hp.blocks) else None
    hp.cfg.level_major_render = bool(args.render_lm)
    tr = rmod.TileRenderer(hp, c2w, intr, tile_rays=args.render_tile_rays, block=block, fineness=fineness, use_graph=True, tile_width=args.render_tile_width,
                           pipeline=args.render_pipeline, decoder_cus=args.render_decoder_cus)
    n_cams = c2w.shape[0]
    tr.render_camera(0, H, W)  # capture + warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(args.render_frames):
        img = tr.render_camera((37 * (f + 1)) % n_cams, H, W)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mpix = args.render_frames * H * W / 1e6
    return dict(metric="render_ms_per_mpix", value=dt * 1e3 / mpix, unit="ms/Mpix", frames=args.render_frames, frame=f"{W}x{H}",
                ms_per_frame=dt * 1e3 / args.render_frames, rays_per_sec=args.render_frames * H * W / dt, tile_rays=args.render_tile_rays, hipgraph=True, tiles_per_replay=args.render_pipeline,
                tile=(f"{args.render_tile_width}x{args.render_tile_rays // args.render_tile_width} pixel blocks" if args.render_tile_width else "raster runs"),
                table="global + focal block 0" if block is not None else "global", fineness=fineness,
                mean_accumulation=float(img["accumulation"].mean().item()))


def batch_assembly_bench(sc, torch, iters=50, n_images=64, H=864, W=1152):
    """SURVEY 8(f) row 3 beside the step it feeds: ErrorPixelSampler draw (20 % by error map, 80 % uniform, without
    replacement) + pixel gather from a u8 image stack resident in HBM + ray generation + the error-map write-back, per
    4096-ray batch.  Reported beside the headline number, not folded into it (the metric's batches are resident)."""
    from gfnerf_amd import data

    dev = torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator(device=dev).manual_seed(11)
    images = torch.randint(0, 256, (n_images, H, W, 3), dtype=torch.uint8, device=dev, generator=gen)
    em = data.ErrorMap(torch.rand((n_images, H, W, 1), device=dev, generator=gen))
    c2w = torch.from_numpy(sc["c2w"][:n_images]).to(dev)
    K = sc["intri"][:n_images]
    intr = torch.from_numpy(np.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], -1).astype(np.float32)).to(dev)
    idx = torch.arange(n_images, dtype=torch.int32, device=dev)
    batch = dict(image=images, image_idx=idx, rel_camera_idx=idx, error_map=em)
    sampler, rg = data.ErrorPixelSampler(N_RAYS, generator=gen), data.RayGenerator(c2w, intr)
    pred = torch.rand((N_RAYS, 3), device=dev, generator=gen)

    def one():
        rays, b, ray_idx = data.next_train(batch, sampler, rg)
        em.update(torch.cat([b["slot"][:, None], ray_idx[:, 1:3]], 1), pred, b["image"])

    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return dict(ms_per_batch=ms, rays=N_RAYS, image_stack=f"{n_images} x {W}x{H} u8 in HBM", sampler="ErrorPixelSampler 20% weighted / 80% uniform",
                includes="index draw, pixel gather, ray generation, error-map write-back + CDF upkeep", duplicates_left=int(sampler.last_status[1].item()))


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import gfnerf_amd
    from gfnerf_amd import _lib, hotpath, scene

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (not used by the driver): several ranks on ONE card need the gloo backend (RCCL refuses duplicate GPUs)
    backend = os.environ.get("GFN_BENCH_BACKEND", "nccl")
    if "GFN_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["GFN_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    _lib.require_device()  # the HIP extension + a gfx950 device, or fail loudly

    # ---- synthetic scene (untimed): cameras + octree from the host C++ builder, cached per box
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene")
    t_build = time.perf_counter()
    if rank == 0:
        sc = scene.build_scene(n_side=args.n_side, max_depth=16, seed=1234, cache_dir=cache)
    if world > 1:
        dist.barrier()
    if rank != 0:
        sc = scene.build_scene(n_side=args.n_side, max_depth=16, seed=1234, cache_dir=cache)
    t_build = time.perf_counter() - t_build
    n_cams = sc["c2w"].shape[0]
    n_blocks = 4 if (args.stage == "focal" and world == 1) else (1 if args.stage == "focal" else 0)
    cfg = hotpath.HotPathConfig(log2_T=args.log2_T, early_T=args.early_T, share_block_primes=args.share_block_primes,
                                mlp_precision=args.mlp_precision, prefetch_cus=args.prefetch_cus, prefetch_cus_global=args.prefetch_cus if args.prefetch_cus != 192 else 224,
                                fused_backward=not args.unfused_backward, lds_accumulate=not args.atomic_backward, async_adam=args.async_adam, **({'record_capacity': args.record_capacity} if args.record_capacity else {}), level_major_global=args.lm_global_forward, **({'level_major_global_after': args.lm_global_after} if args.lm_global_after is not None else {}), lm_halves_from=args.lm_halves_from)
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=n_cams, cfg=cfg, seed=0, n_blocks=n_blocks,
                         device=f"cuda:{local_rank if world > 1 else 0}")
    for i, b in enumerate(hp.blocks):  # mid-training state: non-zero residual tables
        g = torch.Generator(device="cuda").manual_seed(100 + i + 16 * rank)
        b.uniform_(-1e-3, 1e-3, generator=g)
    hp.refresh_shadows()
    # optimiser state and the gradient-record buffers exist before the first step (allocation is set-up, not training)
    if args.stage == "focal":
        for i, b in enumerate(hp.blocks):
            hp._adam_state(f"block{i}", b)
    else:
        hp._adam_state("table", hp.table)
    hp._record_buffers()

    # ---- batches resident in HBM before the timed region (4 rotated batches per rank)
    batches_np, batches = [], []
    for k in range(4):
        o, d, cam = scene.make_rays(sc["c2w"], N_RAYS, seed=1000 * rank + k)
        gt = np.random.default_rng(7 + k).random((N_RAYS, 3)).astype(np.float32)
        batches_np.append((o, d, cam, gt))
        batches.append(tuple(torch.from_numpy(a).cuda() for a in (o, d, cam, gt)))
    target = TARGET_SAMPLES[args.stage]
    fineness, mean_samples = pick_fineness(hp, batches[0][0], batches[0][1], target, scene, torch)
    # The march noise (PersSampler_cuda.cu:380-389) and S3IM's arange + 9 permutations (losses.py:780-789) are drawn on the
    # device EVERY step, inside the timed region like the reference does: step i draws them for step i + 1 on a third stream
    # (they depend on nothing), beside the step's own kernels.
    from gfnerf_amd import ops as _ops
    draw_stream = torch.cuda.Stream()
    drawn = {}

    def draw(i):
        main = torch.cuda.current_stream()
        with torch.cuda.stream(draw_stream):
            nz = _ops.sampler_noise(N_RAYS, fineness, train=True, device="cuda")
            s3 = _ops.s3im_index(N_RAYS, cfg.s3im_repeat_time, device="cuda")
        for t_ in (nz, s3):
            t_.record_stream(main)
            if hp._side is not None:
                t_.record_stream(hp._side)
        drawn[i] = (nz, s3)
        main.wait_stream(draw_stream)

    allreduce = None
    if world > 1 and args.stage == "global":
        def allreduce(grads):  # ray data parallelism: sum of per-rank mean losses -> average
            for g in grads:
                dist.all_reduce(g)
                g.mul_(1.0 / world)

    def step(i):
        o, d, cam, gt = batches[i % 4]
        block = (i % n_blocks) if args.stage == "focal" else None
        nxt = batches[(i + 1) % 4]
        if i not in drawn:
            draw(i)
        draw(i + 1)
        noise, s3 = drawn.pop(i)
        pre = None if args.no_prefetch else (nxt[0], nxt[1], drawn[i + 1][0])  # the data loader knows the next batch
        # global stage: UpdateOctNodes K5/K6 every step, right after the forward like GFNeRFModel.get_outputs (nerfacto.py:598-616)
        out = hp.train_step(o, d, cam, gt, noise, stage=args.stage, block=block, allreduce=allreduce, s3im_index=s3,
                            prefetch=pre, update_oct=args.stage == "global" and not args.no_oct_update)
        return out

    # two priming steps before the W warm-up steps: code objects are loaded, workspaces and the caching allocator's blocks
    # exist (set-up; reported as config.priming_steps)
    PRIMING = 2
    for i in range(PRIMING):
        step(i)
    for i in range(args.warmup):
        step(PRIMING + i)
    # dominant kernel: timed live with events on the launch stream (torch's current stream)
    hp.timed_ops = {"*"} if args.stage_times else {"fused_fwd", "bwd_scatter", "grad_reduce", "hash_fwd", "hash_bwd", "field_fwd", "field_bwd"}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, aux = step(PRIMING + args.warmup + i)
    host_enqueue = time.perf_counter() - t0  # when the host had launched everything: (well) below `elapsed` = the GPU is the bound
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    n_live = aux["samples"]["pts_offsets"][-1].item()

    if rank == 0 and args.stage_times:
        tot = 0.0
        for name, ev in hp.timings.items():
            t = sum(a.elapsed_time(b) for a, b in ev) / args.steps
            tot += t
            print(f"[stage] {name:16s} {t:8.3f} ms/step ({len(ev) // args.steps} launches)", file=sys.stderr)
        print(f"[stage] {'sum':16s} {tot:8.3f} ms/step", file=sys.stderr)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = N_RAYS * world * args.steps / elapsed
        # per-kernel event timings over the timed region; the dominant one carries the roofline.
        # Algorithmic bytes per sample (DESIGN.md "Kernels"): hash lookup = 16 levels x 8 corners x 4 B f16x2 per
        # table (512 B/table) + 16 B point/anchor + 128 B features written; hash scatter = 16 x 8 x 4 B
        # packed-f16 atomic adds + 128 B gradient read + 16 B point/anchor.
        n_tables = 2 if args.stage == "focal" else 1
        # fused forward: only samples in front of each ray's termination point are looked up (sigma > 0 marks them); per such
        # sample 512 B of gathers per table + 16 B point/anchor + 8 B dt/t in, sigma 4 + rgb 12 + h0 4 + masks 48 + weight/alpha 8 out
        n_eval = int((aux["field"]["sigma"][:n_live] > 0).sum().item())
        alg = {"fused_fwd": n_eval * (512 * n_tables + 24 + 76), "hash_fwd": n_live * (512 * n_tables + 16 + 128),
               "hash_bwd": n_live * (16 * 8 * 2 * (2 if cfg.grad_f16 else 4) + 128 + 16),
               "bwd_scatter": n_eval * (16 * 8 * 2 * (2 if cfg.grad_f16 else 4) + 16 + 16 + 56)}
        kern = {}
        for name, ev in hp.timings.items():
            kern[name] = float(sum(a.elapsed_time(b) for a, b in ev)) / args.steps  # per step, however many regions a step times
        dom = max((k for k in kern if k in alg), key=lambda k: kern[k])
        kern_ms = kern[dom]
        alg_bytes = alg[dom]
        kernel_name, launches = dom + "_kernel", 1
        lm = getattr(hp, "last_forward", "") == "lm" and dom == "hash_fwd"
        if lm:
            # level-major lookup: one launch of lm_lookup_kernel per table inside the timed "hash_fwd" region.  Per launch and
            # sample: 512 B gathered + 16 B point/anchor + 128 B of level-major features written (+ the second table's launch
            # reads the first one's 128 B back to add to them: 64 B averaged over the two launches)
            kernel_name, launches, dom = "lm_lookup_kernel", n_tables, "lm_lookup"
            kern_ms = kern_ms / launches
            alg_bytes = n_live * (512 + 16 + 128 + 64)
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # measured separately with rocprofv3 --pmc (see profiles/README.md)
        traffic, req = None, None
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            traffic = pmc.get(f"{dom}:{args.stage}")
            # the PMC pass was taken on the default configuration only (own primes per table, early_T 1e-4, T = 2^21, 400 cameras)
            default_cfg = (not args.share_block_primes) and args.early_T == 1e-4 and args.log2_T == 21 and args.n_side == 20
            if not default_cfg:
                traffic = None
            n_req = pmc.get(f"{dom}:{args.stage}:fetch_requests") if default_cfg else None
            if lm:  # the level-major kernel is bound by how fast a CU's address unit walks 64 divergent lanes, not by the fabric
                n_g = lm_issued_gathers(aux["samples"], n_live, cfg.lm_halves_from, torch)
                req = dict(unit="G gathers/s (4-byte, L2-resident range)", achieved=n_g / (kern_ms * 1e-3) / 1e9, peak=pmc["l2_gather_ceiling_per_s"] / 1e9,
                           frac=n_g / (kern_ms * 1e-3) / pmc["l2_gather_ceiling_per_s"], fabric_requests_per_launch=n_req,
                           nominal_gathers_per_launch=n_live * 128, issued_gathers_per_launch=n_g,
                           source="gathers the kernel issues per launch (16 levels x 8 corners per sample, minus the corners that runs of samples in one "
                                  "cell share below level lm_halves_from: counted from this batch) over the live kernel time; peak = tools/gather_bench.hip on a 4 MB table")
            elif n_req:  # what actually bounds a random 4-byte gather: 64-byte L2-miss requests per second
                req = dict(unit="G L2-miss requests/s", achieved=n_req / (kern_ms * 1e-3) / 1e9, peak=pmc["gather_request_ceiling_per_s"] / 1e9,
                           frac=n_req / (kern_ms * 1e-3) / pmc["gather_request_ceiling_per_s"],
                           source="FETCH_SIZE/64 (profiles/pmc_traffic.json) over the live kernel time; peak = tools/gather_bench.hip")
        roofline = dict(bound="hbm", kernel=kernel_name, launches_per_step=launches, achieved=alg_bytes / (kern_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=traffic, avg_kernel_ms=kern_ms,
                        alg_bytes_per_launch=alg_bytes, kernel_ms={k: round(v, 4) for k, v in kern.items()}, request_rate=req)
        w = aux["comp"]["weights"][:n_live]
        alive_frac = float((w > 0).float().mean().item()) if w is not None else None
        out = dict(metric="train_rays_per_sec", value=value, unit="rays/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=ms, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   samples_per_sec=value * n_live / N_RAYS,
                   config=dict(workload=("GF-NeRF global+focal train step, 4 focal blocks resident (1 active/step), "
                                         if args.stage == "focal" and world == 1 else
                                         "GF-NeRF focal stage, 1 block per GPU, " if args.stage == "focal" else
                                         "GF-NeRF global stage train step (ray data parallel), ") +
                                        f"4096 rays x {target} samples, Rubble-scale synthetic cameras",
                               rays_per_batch=N_RAYS, mean_samples_per_ray=n_live / N_RAYS, log2_table_size=args.log2_T,
                               n_cameras=n_cams, n_tree_nodes=int(sc["n_nodes"]), n_warp_transforms=int(sc["n_trans"]),
                               fineness=fineness, stage=args.stage, early_T=cfg.early_T, alive_sample_frac=alive_frac,
                               evaluated_sample_frac=n_eval / max(n_live, 1), fused_forward=bool(cfg.fused_forward),
                               block_primes="shared with global table" if cfg.share_block_primes else "own per table (reference)",
                               mlp=("split-f16 x3 MFMA 32x32x16 (f32 accuracy)" if cfg.mlp_precision == "x3" else "f32 MFMA 32x32x2"),
                               parallelism=("1 focal block per GPU, no data-path collective" if args.stage == "focal"
                                            else f"dp{world} all-reduce(table,mlp,emb grads)") if world > 1 else "single GPU",
                               scene_build_s=round(t_build, 1), priming_steps=2, host_enqueue_ms_per_step=host_enqueue / args.steps * 1e3, async_adam=bool(cfg.async_adam), per_step_draws="march noise + S3IM permutations drawn on the device every timed step", loss=float(loss.item())),
                   roofline=roofline)
        if args.render_frames > 0 and world == 1:  # the side measurements belong to the single-GPU line
            out["render"] = render_bench(hp, sc, args, fineness, torch)
            out["batch_assembly"] = batch_assembly_bench(sc, torch)
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_reference_field"] = cpu_reference_field()
            out["cpu_baseline"] = cpu_baseline(hp.export_params(), batches_np[0], args.log2_T, fineness, args.cpu_rays,
                                               0 if args.stage == "focal" else None, scene)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

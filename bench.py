#!/usr/bin/env python3
"""bench.py -- train rays/s of the GF-NeRF hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--stage focal|global]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full training pass of the hot path over one batch of 4096 synthetic rays:
PersSampler trace+march -> Hash3DAnchored (global + active focal table) -> base MLP -> density /
SH / appearance head input -> head MLP -> compositing -> Charbonnier + S3IM loss -> full backward ->
Adam.  Workload at N = 1 (BASELINE.json configs[2]): "GF-NeRF global+focal, 4 sub-encoder blocks on 1
GPU, Rubble-scale synthetic cameras (1,600), 4096 rays x 192 samples".  With N > 1 (configs[3]) the
cameras are clustered into N equal blocks, every rank owns one focal block and draws its rays only from
its block's cameras (weak scaling); the reference freezes the global table and the decoder in this
stage, so there is no data-path collective.  The line also carries `global_stage`
(configs[1], 4096 x 128, all parameters train; at N > 1 ray data parallel: octree statistics MAX-reduced,
gradients averaged by a direct all-reduce over RCCL, both inside HotPath.train_step) and `render`
(configs[4]; at N > 1 every frame goes to the rank that owns its block).

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (child
processes, before anything touches a GPU); under torchrun it is one of the ranks.

Prints ONE JSON line on rank 0 (see README / DESIGN.md "Measurement").
"""
import argparse
import functools
import tempfile
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_RAYS = 4096
PSNR_RECORD = "r06_psnr_study.json"  # the round's many-seed PSNR record under profiles/ (tools/psnr_study.py; legs plugin | reference)
TARGET_SAMPLES = {"focal": 192, "global": 128}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PRIMING = 2            # set-up steps before the W warm-up steps (code objects, workspaces, allocator blocks)
GRAPH_WARMUP = 4       # untimed replays between the capture and the timed region (launch mode graph)
CTL_DEV = "cuda"       # where the few control scalars of the job (timings, fineness) are exchanged: the device under RCCL, the host under gloo


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--stage", choices=["focal", "global"], default="focal")
    ap.add_argument("--n-side", type=int, default=40, help="cameras per side of the synthetic capture grid (SURVEY 8(d): 40 x 40 = 1,600)")
    ap.add_argument("--log2-T", type=int, default=21)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-times", action="store_true", help="print per-stage event timings to stderr (adds event overhead)")
    ap.add_argument("--cpu-rays", type=int, default=2048, help="rays in the bounded cpu_baseline sample")
    ap.add_argument("--early-T", type=float, default=1e-4, help="early-termination transmittance (0 = none, the reference)")
    ap.add_argument("--one-phase", action="store_true",
                    help="level-major forward over every sample (no termination-aware split into phases; HotPathConfig.two_phase = False)")
    ap.add_argument("--no-oct-update", action="store_true", help="global stage: skip the per-step UpdateOctNodes (K5/K6)")
    ap.add_argument("--render-frames", type=int, default=2,
                    help="also time N full 1920x1080 frames (per rank) through the hipGraph tile renderer (BASELINE configs[4]); 0 = skip")
    ap.add_argument("--global-steps", type=int, default=200,
                    help="timed steps of the `global_stage` sub-record (configs[1], 4096 x 128); 0 = skip.  Its warm-up is --global-warmup")
    ap.add_argument("--global-warmup", type=int, default=60, help="past the level-major switch at optimiser step 50")
    ap.add_argument("--sub-timeout", type=float, default=240.0, help="seconds a sub-record may take before the line is printed without it")
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph",
                    help="graph (default): the timed steps are hipGraph replays of the captured step (gfnerf_amd.stepgraph): one launch "
                         "per step, immune to a busy host; eager: every kernel enqueued from Python.  Data-parallel global-stage steps "
                         "are always eager (collectives)")
    ap.add_argument("--probe-steps", type=int, default=10, help="eager steps after a graph-timed region that time the individual kernels with HIP events")
    ap.add_argument("--graph-experiment", choices=["off", "eager", "graph"], default="off",
                    help="experiment: ONE fixed batch every step, eagerly launched or as one captured hipGraph replayed per step "
                         "(isolates launch overhead; the captured step's optimiser scalars are frozen, so not a training run)")
    ap.add_argument("--backend", choices=["auto", "nccl", "gloo"], default="auto",
                    help="torch.distributed backend at N > 1 (auto: RCCL whenever every rank has its own GPU -- a failure there is an error, "
                         "never a silent fall-back --, gloo only with --share-gpu; gloo: explicitly, for rehearsals)")
    ap.add_argument("--job-timeout", type=float, default=1500.0, help="N > 1, self-spawned ranks: seconds after which the ranks are killed and the exit code is 124")
    ap.add_argument("--psnr-seeds", type=int, default=3,
                    help="`psnr` sub-record: seeds per leg (the product's training leg against the reference-semantics leg, paired by seed); the many-seed "
                         "record is profiles/r06_psnr_study.json (tools/psnr_study.py)")
    ap.add_argument("--trained-seeds", type=str, default="0,1",
                    help="`trained_state` sub-records (N = 1): engine seeds whose procedurally TRAINED state (global stage, then 4 focal blocks in rotation) the same "
                         "steps are timed on; empty = none")
    ap.add_argument("--trained-steps", type=int, default=300, help="timed steps of a trained-state record")
    ap.add_argument("--psnr-steps", type=str, default="3000,1000",
                    help="`psnr` and `trained_state` sub-records (N = 1): global,focal training steps on the procedural scene per run "
                         "(gfnerf_amd.procedural.STUDY_STEPS); 0,0 = skip")
    ap.add_argument("--global-x3-steps", type=int, default=60,
                    help="timed steps of the global stage with --wgrad-planes x3 (hi + lo' operands of the weight gradient), reported beside the default one-plane record; 0 = skip")
    ap.add_argument("--rays8192-steps", type=int, default=100,
                    help="timed steps of the `rays8192` sub-records (the reference's default batch, gfnerf/config.py:75: focal and global step on 8192 rays, N = 1); 0 = skip")
    ap.add_argument("--plugin-steps", type=int, default=100,
                    help="timed iterations of the `plugin_step` sub-records (the step through GFNerfPipeline + Optimizers, N = 1 only); 0 = skip")
    ap.add_argument("--inject-divergence", action="store_true",
                    help="test hook (N > 1): rank 1 changes one table entry after the global-stage steps -- the run must then report "
                         "replicas_bit_identical false and end with exit code 5")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: allow more ranks than visible GPUs (ranks share cards; gloo instead of RCCL)")
    ap.add_argument("--lm-halves-from", type=int, default=6, help="level-major lookup: levels >= this are gathered one address part at a time")
    ap.add_argument("--lm-split-log2", type=int, default=0, help="level-major lookup: 2^this address parts per level (0 / 1: halves; 2: quarters)")
    ap.add_argument("--lm-global-after", type=int, default=None, help="global stage: level-major forward from this optimiser step on (-1: never; default: HotPathConfig)")
    ap.add_argument("--record-capacity", type=int, default=0, help="gradient record buffer in records (0: HotPathConfig default)")
    ap.add_argument("--fork-first", action="store_true", help="forward_backward forks the octree update / next sampler before it enqueues the loss kernels (round-2 order)")
    ap.add_argument("--segmented", action="store_true",
                    help="experiment: capture the global step in the segments of the data-parallel step (exchange points eager, identities on one rank)")
    ap.add_argument("--draws-first", action="store_true",
                    help="graph launch: start the next step's random draws at the top of the step (beside the lookups) instead of behind them")
    ap.add_argument("--async-adam", action="store_true",
                    help="focal stage: Adam of step i on a third stream beside the frozen global table's lookup of step i + 1 (about "
                         "-1.5 percent per step; off by default so that the roofline kernel is timed running alone)")
    ap.add_argument("--lm-global-forward", action="store_true", help="global stage: level-major lookup + decoder + compositing instead of the fused forward kernel")
    ap.add_argument("--atomic-backward", action="store_true", help="backward issues its table adds as global atomics instead of records accumulated in LDS")
    ap.add_argument("--unfused-backward", action="store_true", help="decoder backward and hash scatter as two kernels")
    ap.add_argument("--no-prefetch", action="store_true", help="do not run the next batch's sampler beside this step's backward")
    ap.add_argument("--prefetch-cus", type=int, default=0, help="CUs left to the fused backward while the prefetched sampler runs beside it")
    ap.add_argument("--render-pipeline", type=int, default=0,
                    help="tiles per hipGraph replay: their samplers run beside the fused forwards on a second graph branch (1 = one tile per replay)")
    ap.add_argument("--no-render-x3", action="store_true", help="skip the render sub-record's second pass with the x3 decoder form throughout (profile runs: ONE inference "
                    "kernel instantiation in the trace / counter tables)")
    ap.add_argument("--render-eager", action="store_true", help="render tiles as eager launches, one tile at a time (counter collection: rocprofv3 --pmc serialises "
                    "dispatches anyway and sees eager launches reliably); the kernels are the graph's")
    ap.add_argument("--render-decoder-cus", type=int, default=224, help="CUs the fused forward may occupy while a sampler runs beside it")
    ap.add_argument("--render-patch-width", type=int, default=8,
                    help="render: a wave walks a patch of this x 32/this neighbouring pixels' rays in lockstep (0: a wave per ray, like the training forward)")
    ap.add_argument("--render-tile-rays", type=int, default=0, help="rays per render tile (0: render.choose_tile picks a block of ~512 Ki rays that divides the frame, capped by the free device memory)")
    ap.add_argument("--render-tile-width", type=int, default=128, help="with --render-tile-rays: render tiles are blocks of this width (0 = raster runs)")
    ap.add_argument("--wgrad-planes", choices=["rne", "x3", "mixed"], default="rne",
                    help="global stage: operands of the decoder's weight gradient (HotPathConfig.wgrad_planes): rne = one f16 plane per operand, rounded to "
                         "nearest (default); x3 = hi + lo' planes (rounds 1-4); mixed = dz one plane, activations split")
    ap.add_argument("--record-merge", type=int, default=-1, help="focal backward in record form: merge runs of samples in one cell first, 0 / 1 for every launch (HotPathConfig.record_merge); -1: the engine's policy (dense batches only)")
    ap.add_argument("--wgrad-beside-accumulate", action="store_true", help="experiment (global stage): wgrad on a second stream beside grad_accumulate")
    ap.add_argument("--mlp-precision", choices=["x3", "f32"], default="x3",
                    help="fused decoder arithmetic: x3 = split-f16 MFMA at f32 accuracy (default), f32 = exact f32 MFMA")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start one child process per rank (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in its environment, like torch.distributed.run does) and wait for them.  This parent never touches a GPU
    (device_count() does not initialise HIP), so nothing that holds a GPU is ever replaced or forked."""
    import torch

    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not args.share_gpu:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible (use --share-gpu to rehearse ranks on shared cards)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(args.gpus))
        # rank 0 owns stdout (the JSON line); the others only report on stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    t_end = time.monotonic() + args.job_timeout
    try:
        while procs:
            if time.monotonic() > t_end:  # a rank that hangs (in a collective, in a rendezvous) must not hang the job
                print(f"bench.py: ranks still running after {args.job_timeout:.0f} s: killed", file=sys.stderr)
                rc = 124
                break
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:  # a rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            p.kill()
    return rc


class Deadline:
    """Safety net around a sub-record that talks to other ranks: if it has not finished after `seconds`, `on_expire` runs
    on a helper thread (rank 0 prints the line with an "error" entry for the sub-record) and the process leaves with exit code 3
    (os._exit: the main thread sits in a collective that will never return)."""

    def __init__(self, seconds, on_expire):
        self.seconds, self.on_expire, self._done = seconds, on_expire, threading.Event()

    def __enter__(self):
        def watch():
            if not self._done.wait(self.seconds):
                try:
                    self.on_expire()
                finally:
                    _OUT.flush()
                    os._exit(3)

        threading.Thread(target=watch, daemon=True).start()
        return self

    def __exit__(self, *exc):
        self._done.set()
        return False


# ------------------------------------------------------------------------------------------------ helpers
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


def lm_issued_gathers(samples, n_live, halves_from, torch, mask=None):
    """Gathers one lm_lookup_kernel launch issues for this batch: 8 per (sample, level), except that below level `halves_from`
    only the first lane of a run of consecutive samples in one cell of one volume gathers when a wave holds <= 40 such runs
    (csrc/lm_lookup.hip).  Waves are taken as aligned groups of 64 samples (the kernel's start at its workgroup's run)."""
    pts = samples["warped_pts"][:n_live]
    vol = samples["anchors"][:n_live, 0]
    want = torch.ones(n_live, dtype=torch.bool, device=pts.device) if mask is None else mask[:n_live] != 0  # phased forward: the launch's share
    total = 0
    lane0 = (torch.arange(n_live, device=pts.device) % 64) == 0
    pad = (-n_live) % 64
    for level in range(16):
        if level >= halves_from:
            total += 8 * int(want.sum().item())
            continue
        cell = torch.floor(pts * float(np.float32(2.0 ** (7.0 * level / 15.0 + 3.0)))).to(torch.int32)
        head = torch.ones(n_live, dtype=torch.bool, device=pts.device)
        head[1:] = (vol[1:] != vol[:-1]) | (cell[1:] != cell[:-1]).any(1) | ~want[:-1]
        head |= lane0
        head &= want
        per_wave = torch.nn.functional.pad(head.to(torch.int32), (0, pad)).view(-1, 64).sum(1)
        shared = torch.repeat_interleave(per_wave <= 40, 64)[:n_live]
        total += 8 * int(torch.where(shared, head, want).sum().item())
    return total


def _sha256(path):
    import hashlib

    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def _cores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()


def cpu_baseline(P, batch, log2_T, fineness, n_rays, block, scene):
    """The oracle (a CPU port of the same path) on a bounded slice of the same batch, all host threads
    for the OpenMP-parallel pieces.  fwd + bwd, no optimiser step."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pipeline as opipe  # oracle: test infrastructure, used here only as the reported CPU baseline

    o, d, cam, gt = (a[:n_rays] for a in batch)
    noise = scene.make_noise(n_rays, True, fineness, seed=1)
    t0 = time.perf_counter()
    loss, grads, aux = opipe.loss_and_grads(P, o, d, cam, gt, noise, log2_T, block=block, bwd_threads=0)
    dt = time.perf_counter() - t0
    n_samples = int(aux["samples"]["pts_offsets"][-1])
    return dict(value=n_rays / dt, unit="rays/s", cores=_cores(), kind="port",
                sample=f"{n_rays} rays ({n_samples} samples) of the same batch, oracle fwd+bwd (no Adam), {dt:.1f} s; "
                       "all stages OpenMP on all cores (trace, march, hash fwd/bwd, decoder fwd/bwd, compositing fwd/bwd)")


def cpu_baseline_global(P, batch, log2_T, fineness, scene, n_packed=1024, n_dense=48):
    """BASELINE.md section 3, row C2 (configs[1], global stage 4096 x 128) on the host cores, bounded samples of the same batch:
    `packed` = the oracle forward on the valid samples only (per-stage split); `dense1024` = the same forward run the way the
    reference executes it -- every ray padded to MAX_SAMPLE_PER_RAY = 1024 points (anchors 0, points 0, delta 0:
    PersSampler_cuda.cu:437-444) that all go through hash + both MLPs (nerfacto_field.py:443)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    import pipeline as opipe

    out = {}
    o, d, cam, _ = (a[:n_packed] for a in batch)
    noise = scene.make_noise(n_packed, True, fineness, seed=1)
    t = [time.perf_counter()]
    m = opipe.sample(P, o, d, noise); t.append(time.perf_counter())
    f = opipe.field_forward(P, m, cam, log2_T); t.append(time.perf_counter())
    O.composite_fwd(m["pts_offsets"], f["sigma"], m["dists"], m["ts"], f["rgb"]); t.append(time.perf_counter())
    n = int(m["pts_offsets"][-1])
    tot = t[-1] - t[0]
    out["packed"] = dict(value=n_packed / tot, unit="rays/s (forward)", samples_per_sec=n / tot, cores=_cores(), kind="port",
                         sample=f"{n_packed} rays ({n} samples) of the global-stage batch, oracle forward, {tot:.1f} s",
                         stage_ms=dict(sampler=(t[1] - t[0]) * 1e3, field=(t[2] - t[1]) * 1e3, composite=(t[3] - t[2]) * 1e3))
    # reference-emulating dense mode: [R,1024] points per ray, padding = zeros
    o, d, cam, _ = (a[:n_dense] for a in batch)
    noise = scene.make_noise(n_dense, True, fineness, seed=1)
    t0 = time.perf_counter()
    m = opipe.sample(P, o, d, noise)
    off = m["pts_offsets"]
    M = 1024
    cnt = np.minimum(np.diff(off), M)
    md = dict(m)
    x = np.full((n_dense * M, 3), (np.float32(0.0) + np.float32(1.5)) * (np.float32(1.0) / np.float32(3.0)), np.float32)
    anchors = np.zeros((n_dense * M, 3), m["anchors"].dtype)
    dists = np.zeros(n_dense * M, np.float32)
    ts = np.zeros(n_dense * M, np.float32)
    for r in range(n_dense):
        a, c = off[r], cnt[r]
        x[r * M:r * M + c] = m["x"][a:a + c]; anchors[r * M:r * M + c] = m["anchors"][a:a + c]
        dists[r * M:r * M + c] = m["dists"][a:a + c]; ts[r * M:r * M + c] = m["ts"][a:a + c]
    md.update(x=x, anchors=anchors, dists=dists, ts=ts, ray_id=np.repeat(np.arange(n_dense), M).astype(np.int64),
              pts_offsets=(np.arange(n_dense + 1) * M).astype(off.dtype))
    f = opipe.field_forward(P, md, cam, log2_T)
    O.composite_fwd(md["pts_offsets"], f["sigma"], md["dists"], md["ts"], f["rgb"])
    dt = time.perf_counter() - t0
    out["dense1024"] = dict(value=n_dense / dt, unit="rays/s (forward)", samples_per_sec=n_dense * M / dt, cores=_cores(), kind="port",
                            sample=f"{n_dense} rays x 1024 padded points ({int(cnt.sum())} valid), oracle forward as the reference executes it, {dt:.1f} s")
    return out


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


def render_bench(hp, sc, args, fineness, torch, cams, block, world=1, rank=0, owners=None, dist=None):
    """BASELINE configs[4]: full-res camera-path pass, 1920x1080, hipGraph-captured per-tile inference (eval-mode ray
    march: noise = ones * fineness; same fineness as the training batches).  Untimed: graph capture + one warm-up frame.
    cams: the path's cameras (global ids); with world > 1 `owners[f]` is the rank that renders frame f (the owner of the
    frame's block), every rank renders its own frames, rank 0 gathers the finished images."""
    from gfnerf_amd import render as rmod

    H, W = 1080, 1920
    c2w = torch.from_numpy(sc["c2w"]).cuda()
    K = sc["intri"]
    zoom = W / 1152.0  # the synthetic capture's 1152-wide intrinsics scaled to the output width
    intr = torch.from_numpy(np.stack([K[:, 0, 0] * zoom, K[:, 1, 1] * zoom, np.full(K.shape[0], W / 2), np.full(K.shape[0], H / 2)], -1)
                            .astype(np.float32)).cuda()
    hp.cfg.render_patch_width = int(args.render_patch_width)
    tr = rmod.TileRenderer(hp, c2w, intr, tile_rays=args.render_tile_rays or None, block=block, fineness=fineness, use_graph=not args.render_eager,
                           tile_width=args.render_tile_width if args.render_tile_rays else None, pipeline=args.render_pipeline or None,
                           decoder_cus=args.render_decoder_cus)
    mine = [f for f in range(len(cams)) if owners is None or owners[f] == rank]
    tr.render_camera(int(cams[mine[0]]) if mine else 0, H, W)  # capture + warm-up
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    local = {}
    for f in mine:
        local[f] = tr.render_camera(int(cams[f]), H, W)
    torch.cuda.synchronize()
    if world > 1:
        frames = rmod.gather_frames(local, len(cams), H, W, dst=0, owners=owners)  # the path's one collective
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=CTL_DEV)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    n_frames = len(cams)
    mpix = n_frames * H * W / 1e6
    acc = float(local[mine[-1]]["accumulation"].mean().item()) if mine else None
    roof = None
    if world == 1 and mine:
        roof = render_roofline(hp, tr, c2w, intr, int(cams[mine[-1]]), H, W, block, fineness, dt / len(mine), torch)
    fp = tr.footprint_bytes()
    # the renderer's default decoder form (HotPathConfig.render_colour_x2: the colour head's layer 2 on one f16 activation operand; density, accumulation and
    # depth are the x3 form's bits) -- and, beside it, the same frames with the split-f16 x3 form throughout and how far the two frames' rgb are apart
    x3_ms, x2_dist = None, None
    if world == 1 and mine and tr.colour_x2 and getattr(tr, "record", False) and not args.no_render_x3:
        last = {k: local[mine[-1]][k].clone() for k in ("rgb", "accumulation", "depth")}
        tile_rays, rec_slots = tr.T, tr.rec_slots
        del tr, local
        torch.cuda.empty_cache()
        tr = rmod.TileRenderer(hp, c2w, intr, tile_rays=args.render_tile_rays or None, block=block, fineness=fineness, use_graph=not args.render_eager,
                               tile_width=args.render_tile_width if args.render_tile_rays else None, pipeline=args.render_pipeline or None,
                               decoder_cus=args.render_decoder_cus, colour_x2=False)
        tr.render_camera(int(cams[mine[0]]), H, W)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for f in mine:
            f3 = tr.render_camera(int(cams[f]), H, W)
        torch.cuda.synchronize()
        x3_ms = (time.perf_counter() - t1) * 1e3 / mpix
        x2_dist = dict(max_abs_rgb=float((f3["rgb"] - last["rgb"]).abs().max().item()),
                       accumulation_and_depth_same_bits=bool(torch.equal(f3["accumulation"], last["accumulation"]) and torch.equal(f3["depth"], last["depth"])))
    return dict(metric="render_ms_per_mpix", value=dt * 1e3 / mpix, unit="ms/Mpix", frames=n_frames, frame=f"{W}x{H}", roofline=roof,
                decoder=("split-f16 x3; colour head layer 2 on one f16 activation operand (colour_x2: density / accumulation / depth are the x3 form's bits)"
                         if tr.colour_x2 or x3_ms is not None else "split-f16 x3"), x3_ms_per_mpix=x3_ms, colour_x2_vs_x3=x2_dist,
                hbm_footprint_gb=None if fp is None else fp / 1e9, record_slots_per_ray=tr.rec_slots if getattr(tr, "record", False) else None,
                record_pool_regrown=tr.pool_regrown,
                ms_per_frame=dt * 1e3 / max(len(mine), 1), rays_per_sec=n_frames * H * W / dt, tile_rays=tr.T, hipgraph=tr.use_graph,
                tiles_per_replay=tr.P,
                kernel=("fused_render_patch_kernel: a wave walks %d x %d neighbouring pixels' rays in lockstep, samples read from the counting march's records" % (args.render_patch_width, 32 // args.render_patch_width)
                        if args.render_patch_width > 0 and tr.patch_tw is not None else "fused_fwd_kernel: a wave per ray"),
                tile=(f"{tr.tw}x{tr.T // tr.tw} pixel blocks" if tr.tw else "raster runs"),
                table="global + focal block" if block is not None else "global", fineness=fineness, mean_accumulation=acc,
                routing=("single GPU" if world == 1 else f"{n_frames} frames routed to the rank owning each frame's block "
                         f"({len(mine)} on rank 0), images gathered on rank 0 inside the timed region"))


def render_roofline(hp, tr, c2w, intr, cam, H, W, block, fineness, s_per_frame, torch):
    """What a frame's inference kernel evaluates, against the rates that bound it.  The samples in front of each ray's termination
    point are counted by the wave-per-ray forward (HotPath.render without a tile width: it returns the weights) over the same frame's rays in raster
    chunks; per such sample the kernel issues 16 levels x 8 corners gathers per table and 61,952 useful decoder FLOP (x 3 MFMAs issued).  Rates are
    over the WHOLE frame time (samplers, copies and the 32 CUs left to the samplers included): lower bounds of what the kernel itself sustains;
    its own duration and counters are profiles/r04_render_kernel_stats.csv / r04_pmc_render_per_kernel.csv."""
    from gfnerf_amd import ops

    rays = ops.generate_rays(c2w, intr, cam0=cam, first_pixel=0, width=W, n=H * W, want_extras=False)
    O, D = rays["origins"], rays["directions"]
    n_tables = 2 if block is not None else 1
    chunk, n_eval, n_all = 65536, 0, 0
    for a in range(0, H * W, chunk):
        o, d = O[a:a + chunk].contiguous(), D[a:a + chunk].contiguous()
        n = o.shape[0]
        noise = torch.full((1024 + n + 10,), float(fineness), device="cuda")
        out = hp.render(o, d, torch.full((n,), cam, dtype=torch.int32, device="cuda"), noise, block=block)
        m = int(out["samples"]["pts_offsets"][-1].item())
        n_all += m
        n_eval += int((out["weights"][:m] > 0).sum().item())
        del out
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
    fresh = bool(pmc.get("source_sha256", {}).get("render")) and all(
        _sha256(os.path.join(ROOT, "gf-nerf_amd", f)) == h for f, h in pmc["source_sha256"]["render"].items())
    gathers = n_eval * 128.0 * n_tables
    ceiling = pmc.get("l2_gather_ceiling_per_s")
    return dict(evaluated_samples_per_frame=n_eval, sampled_per_frame=n_all, evaluated_frac=n_eval / max(n_all, 1),
                samples_per_sec=n_eval / s_per_frame, nominal_gathers_per_sec=gathers / s_per_frame, distinct_address_gather_ceiling_per_s=ceiling,
                gather_note="16 levels x 8 corners per table and evaluated sample; the lanes of a patch want 0.09 distinct cells per sample and level "
                            "(tools/exp_patch_sharing.py) and a gather instruction costs its DISTINCT addresses (tools/gather_share.hip), so the nominal rate "
                            "exceeds the ceiling measured for distinct addresses: the kernel is bound by instruction issue, not by the gather path",
                mfma_tflops_useful=n_eval * 61952.0 / s_per_frame / 1e12, mfma_per_tile=160 if (tr.colour_x2 and block is not None) else 192,  # (216 until round 6: the image's embedding is folded into the head's bias; colour_x2: head layer 2 issues 64 instead of 96)
                mfma_tflops_issued=n_eval * (160 if (tr.colour_x2 and block is not None) else 192) * 32768.0 / 32.0 / s_per_frame / 1e12, mfma_peak_tflops=2500.0,
                mfma_busy_frac=pmc.get("mfma_busy:render") if fresh else None, valu_busy_frac=pmc.get("valu_busy:render") if fresh else None,
                lockstep_efficiency=(n_eval / (pmc["decoded_samples_per_launch:render"] * H * W / pmc["tile_rays:render"])
                                     if fresh and pmc.get("decoded_samples_per_launch:render") and pmc.get("tile_rays:render") else None),  # useful samples /
                # samples the lockstep patches decode (the counter run's launches are tiles of pmc["tile_rays:render"] rays)
                kernel=("fused_render_patch_kernel<1, 2, true, true, %d, true>" % (8 if tr.colour_x2 else 0)) if block is not None else "fused_render_patch_kernel<0, 2, true, true>",
                kernel_ms_per_tile=(pmc["kernel_ms:render"] * tr.T / pmc["tile_rays:render"] if fresh and pmc.get("kernel_ms:render") and pmc.get("tile_rays:render") else None),
                note="rates over the whole frame time; counters from profiles/pmc_traffic.json only while csrc/fused_fwd.hip, x3_device.h, hash_device.h are the profiled ones")


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


def procedural_image_stack(sc, torch, image_idx, H, W, chunk=1 << 17):
    """Renderings [n,H,W,3] u8 of the procedural terrain (gfnerf_amd.procedural, the PSNR study's ground truth) from the cameras `image_idx` of the
    capture at H x W pixels, through the datamanager's own ray generator (pixel centres as the pipeline will draw them) -> (images, intrinsics [n_cams,3,3])."""
    from gfnerf_amd import data, procedural

    dev = torch.device("cuda", torch.cuda.current_device())
    K = np.array(sc["intri"], np.float32, copy=True)
    K[:, 0, :] *= W / 1152.0
    K[:, 1, :] *= H / 864.0
    intr = torch.from_numpy(np.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], -1).astype(np.float32)).to(dev)
    rg = data.RayGenerator(torch.from_numpy(np.ascontiguousarray(sc["c2w"])).to(dev), intr)
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    images = torch.empty((len(image_idx), H, W, 3), dtype=torch.uint8, device=dev)
    for k, cam in enumerate(image_idx):
        idx = torch.stack([torch.full_like(yy, int(cam)), yy, xx], -1).reshape(-1, 3)
        out = images[k].view(-1, 3)
        for a in range(0, H * W, chunk):
            r = rg(idx[a:a + chunk])
            out[a:a + chunk] = (procedural.gt_colors(r["origins"], r["directions"], procedural.STUDY_DETAIL).clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)
    return images, K


def plugin_step_bench(sc, torch, stage, steps=200, warmup=30, n_rays=N_RAYS, fused_optimizer=True, prefetch=True, nan_check=True, graph_forward=True, trained=None,
                      hot_cfg=None):
    """The training iteration THROUGH the plugin surface, the way the reference's Trainer drives it (engine/trainer.py:195-231,
    382-441): callbacks before -> optimizers.zero_grad_all -> pipeline.get_train_loss_dict(step) (incl. datamanager.next_train)
    -> sum(loss_dict) -> backward -> the NaN check over model.named_parameters() -> optimizers.optimizer_step_all ->
    scheduler_step_all -> callbacks after; rays/s = rays per batch / iteration wall time (trainer.py:225-231).
    The pipeline, model, field, sampler, optimizers, scheduler and callbacks are the mirror classes of gfnerf_amd.gfnerf with
    the `gf-nerf` method config's values (gfnerf/config.py) on the bench scene; the datamanager is the HBM-resident synthetic
    one (image loading is out of scope).  stage "global": steps 100.., 4096 x 128; "focal": steps 30100.., 4096 x 192, block 0.
    trained = (steps_global, steps_focal) (stage "focal"): the image stack holds renderings of the procedural terrain (the PSNR
    study's ground truth) instead of noise, the schedule is the study's (steps_global global iterations, then block 0's period), and the pipeline
    TRAINS through this very surface from step 0 -- stage switch, camera clustering, octree upkeep and all -- before the last `steps` iterations of
    block 0's period are timed: the reference's own definition of rays/s (trainer.py:225-231) on a trained scene."""
    from gfnerf_amd.gfnerf._host import AdamOptimizerConfig, Optimizers, TrainingCallbackAttributes, TrainingCallbackLocation
    from gfnerf_amd.gfnerf.gf_pipeline import GFNerfPipelineConfig, HbmDataManagerConfig
    from gfnerf_amd.gfnerf.nerfacto import GFNeRFModelConfig
    from gfnerf_amd.gfnerf.optim import GfnAdamOptimizerConfig
    from gfnerf_amd.gfnerf.schedulers import GFNerfExponentialDecaySchedulerConfig

    S_INIT, S_SPLIT, N_SPLIT = (30000, 10000, 4) if trained is None else (int(trained[0]), int(trained[1]), 4)
    if trained is not None and (stage != "focal" or S_SPLIT < steps + 100):
        raise ValueError("plugin_step_bench(trained=...): the focal stage, with a block period longer than the timed window")
    mk = dict(log2_hashmap_size=21, hidden_dim=128, hidden_dim_color=128, steps_perssampler_init=S_INIT, steps_per_split_dataset=S_SPLIT,
              n_split_dataset=N_SPLIT, n_blocks=N_SPLIT, scale_factor=10.0, use_appearance_embedding=True, use_ch_loss=True, s3im_loss_mult=1.0,
              eval_num_rays_per_chunk=2048)
    # the cached image stack of the datamanager (train_num_images_to_sample_from images resident at a time, gfnerf/config.py:77): 64
    # synthetic 1152 x 864 u8 images of cameras spread over the whole capture, in HBM; pixels are drawn, gathered and turned into
    # rays on the device (gfnerf_amd.data)
    n_img = 64
    gen = torch.Generator(device="cuda").manual_seed(11)
    image_idx = (np.arange(n_img) * (sc["c2w"].shape[0] // n_img)).astype(np.int32)
    if trained is None:
        images, intri = torch.randint(0, 256, (n_img, 864, 1152, 3), dtype=torch.uint8, device="cuda", generator=gen), sc["intri"]
    else:
        images, intri = procedural_image_stack(sc, torch, image_idx, 432, 576)
    pcfg = GFNerfPipelineConfig(datamanager=HbmDataManagerConfig(c2w=sc["c2w"], intrinsics=intri, images=images, image_idx=image_idx,
                                                                 train_num_rays_per_batch=n_rays),
                                model=GFNeRFModelConfig(**mk), prefetch_next_batch=prefetch)
    pipe = pcfg.setup(device="cuda:0", test_mode="val", world_size=1, local_rank=0, base_dir=tempfile.mkdtemp(prefix="gfn_plugin_bench_"))  # (its own: a block table checkpoint left by another run would be resumed)
    model = pipe.model
    model.hot.cfg.graph_forward = bool(graph_forward)
    for k, v in (hot_cfg or {}).items():  # experiments (tools/exp_plugin_trained.py): any HotPathConfig field
        setattr(model.hot.cfg, k, v)
    sched = GFNerfExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=S_INIT, n_split_dataset=N_SPLIT, n_dataset_circles=1,
                                                  steps_per_split_dataset=S_SPLIT, steps_perssampler_init=S_INIT)
    ocfg = (GfnAdamOptimizerConfig if fused_optimizer else AdamOptimizerConfig)(lr=1e-2, eps=1e-15)
    optimizers = Optimizers({"fields": {"optimizer": ocfg, "scheduler": sched}}, pipe.get_param_groups())
    callbacks = pipe.get_training_callbacks(TrainingCallbackAttributes(optimizers=optimizers, grad_scaler=None, pipeline=pipe))
    pipe.train()
    nat = model.persampler.sampler
    first = (100 if stage == "global" else S_INIT + 100) if trained is None else 0

    def pin_fineness(step, target):
        """The ray-march fineness that gives `target` samples per ray on the tree as it is now, held from here on (UpdateRayMarch follows the
        step: 16 -> 1 over the first 10,000 steps, PersSampler.cpp:958-967)."""
        rb = pipe.datamanager.next_train(step)[0]
        lo, hi = 0.5, 64.0
        for _ in range(14):
            f = (lo * hi) ** 0.5
            nat.ray_march_fineness_ = f
            mean = nat.sample_packed(rb.origins, rb.directions, table_space=True, want_world=False)["pts_offsets"][-1].item() / n_rays
            if abs(mean - target) / target < 0.005:
                break
            lo, hi = (f, hi) if mean > target else (lo, f)
        nat.UpdateRayMarch = lambda cur_step, _f=f: setattr(nat, "ray_march_fineness_", _f)
        return mean

    mean = pin_fineness(first, TARGET_SAMPLES[stage if trained is None else "global"])
    if trained is None:
        nat.compact_freq_, nat.sub_div_milestones_ = 1 << 30, []  # tree maintenance (ProcOctree every 1000 steps) outside the window

    marks = [] if os.environ.get("GFN_PLUGIN_EVENTS") else None  # where the device spends the iteration (events on the launch stream + the second stream's tail)

    def mark(stream=None):
        if marks is not None and marks and marks[-1] is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream) if stream is not None else e.record()
            marks[-1].append(e)

    def train_iteration(step):
        if marks is not None:
            marks.append([])
        mark()
        for cb in callbacks:
            cb.run_callback_at_location(step, location=TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
        optimizers.zero_grad_all()
        _, loss_dict, metrics_dict = pipe.get_train_loss_dict(step=step)
        mark()
        loss = functools.reduce(torch.add, loss_dict.values())  # trainer.py:391
        loss.backward()
        mark()
        is_nan = False
        if nan_check:  # trainer.py:415-423 (one host synchronisation per parameter that carries a .grad)
            for tag, value in model.named_parameters():
                if value.grad is not None and torch.isnan(value.grad).any():
                    is_nan = True
        if not is_nan:
            optimizers.optimizer_step_all()
            optimizers.scheduler_step_all(step)
        mark()
        if model.hot._side is not None:
            mark(model.hot._side)
        for cb in callbacks:
            cb.run_callback_at_location(step, location=TrainingCallbackLocation.AFTER_TRAIN_ITERATION)
        return loss

    step = first
    trained_info = None
    if trained is not None:
        # the whole run through the plugin surface: the global stage, the stage switch (camera clusters, block 0's table), block 0's period up to the
        # timed window.  Fineness re-pinned to the stage's sample budget where the tree or the stage changes; ProcOctree at its own cadence
        t_train = time.perf_counter()
        last = S_INIT + S_SPLIT - steps - 8
        while step < last:
            if step in (S_INIT // 3, 2 * S_INIT // 3):
                pin_fineness(step, TARGET_SAMPLES["global"])  # (the tree has been pruned: hold 128 samples per ray)
            if step == S_INIT + 1 or step == last - warmup:
                mean = pin_fineness(step, TARGET_SAMPLES["focal"])
            if step == last - warmup:
                nat.compact_freq_, nat.sub_div_milestones_ = 1 << 30, []  # no tree maintenance inside the timed window
            loss = train_iteration(step); step += 1
        torch.cuda.synchronize()
        trained_info = dict(train_steps=[S_INIT, step - S_INIT], train_seconds=time.perf_counter() - t_train, image_stack="64 renderings of the procedural terrain, 576 x 432",
                            loss_before_window=float(loss.detach()))
    else:
        for _ in range(warmup):
            train_iteration(step); step += 1
    torch.cuda.synchronize()
    prof = None
    if os.environ.get("GFN_PLUGIN_PROFILE"):  # where the host spends the iteration (tools/README.md)
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = train_iteration(step); step += 1
    host = time.perf_counter() - t0
    if prof is not None:
        import pstats

        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(45)
        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(60)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if marks:  # mean over the timed iterations: forward + losses | backward | optimiser | gap to the next iteration; the second stream's tail behind the optimiser
        rows = [m for m in marks[-steps:] if len(m) >= 4]
        seg = lambda a, b: float(np.mean([r[a].elapsed_time(r[b]) for r in rows]))  # noqa: E731
        nxt = float(np.mean([rows[i][3].elapsed_time(rows[i + 1][0]) for i in range(len(rows) - 1)]))
        side = float(np.mean([r[3].elapsed_time(r[4]) for r in rows if len(r) > 4])) if any(len(r) > 4 for r in rows) else None
        print(f"[plugin events] forward+losses {seg(0, 1):.3f}  backward {seg(1, 2):.3f}  optimiser {seg(2, 3):.3f}  to next iteration {nxt:.3f} ms; second stream ends "
              f"{side if side is None else round(side, 3)} ms after the optimiser", file=sys.stderr, flush=True)
    if trained_info is not None and model.hot._records is not None:  # how full the backward's record streams run on this scene (one more iteration, counted)
        hot = model.hot
        hot.record_counts_seen = torch.zeros_like(hot._records[1])
        train_iteration(step); step += 1
        torch.cuda.synchronize()
        seen = hot.record_counts_seen.to(torch.int64)
        n_streams = seen.numel()
        cap = (hot._records[0].numel() // n_streams) & ~1
        trained_info.update(records_per_step=int(seen.sum().item()), record_stream_cap=int(cap), streams_over_cap=int((seen > cap).sum().item()),
                            records_over_cap=int((seen - cap).clamp(min=0).sum().item()), busiest_stream=int(seen.max().item()),
                            table_grad_nonzero=int((hot.table_grad != 0).sum().item()), loss_after_window=float(loss.detach()))
        hot.record_counts_seen = None
    return dict(metric="train_rays_per_sec", value=n_rays / dt, unit="rays/s", ms_per_step=dt * 1e3, host_ms_per_step=host / steps * 1e3, steps=steps,
                warmup=warmup, stage=stage, mean_samples_per_ray=mean, loss=float(loss.detach()),
                optimizer="GfnAdam (HIP Adam kernels on the engine's gradient buffers)" if fused_optimizer else "torch.optim.Adam on dense f32 gradients",
                prefetch_next_batch=prefetch, nan_check=nan_check, trained=trained_info, forward=getattr(model.hot, "last_forward", None),
                surface="GFNerfPipeline.get_train_loss_dict -> backward -> Optimizers.optimizer_step_all -> scheduler_step_all (trainer.py:382-441)")


def psnr_study_record(steps_global, steps_focal):
    """The many-seed PSNR record of the round (profiles/r06_psnr_study.json, written by tools/psnr_study.py on an MI355X): reported beside this run's few
    seeds ONLY when it was made with the same protocol (gfnerf_amd.procedural.PROTOCOL, steps)."""
    from gfnerf_amd import procedural

    path = os.path.join(ROOT, "profiles", PSNR_RECORD)
    try:
        rec = json.load(open(path))
        st = rec["psnr_study"]
        if st["steps"] != [steps_global, steps_focal] or rec.get("protocol") != procedural.PROTOCOL:
            return dict(study_note=f"profiles/{PSNR_RECORD} was made with another protocol: not reported")
        return dict(study_delta=st["delta"], study_se=st["se"], study_n_seeds=st["n_seeds"], study_decoder_delta=st["decoder_delta"], study_legs=st["legs"],
                    study_source=f"profiles/{PSNR_RECORD} (tools/psnr_study.py; same protocol)")
    except (OSError, KeyError, ValueError):
        return {}


def rays8192_records(sc, args, torch, sync, steps=100):
    """The reference's DEFAULT batch -- train_num_rays_per_batch = 8192 (gfnerf/config.py:75) -- where the contract's headline is quoted on 4096 rays
    (BASELINE configs[1], configs[2]): the same two steps, random-init engines, twice the rays.  The lookup's passes and the launch sequence do not
    grow with the batch, so the plugin's default runs at a better rays/s than the contract batch: a sub-record, not the headline."""
    import copy

    a2 = copy.copy(args)
    a2.record_capacity = max(args.record_capacity, 1 << 28)  # (twice the records per stream: keep the streams from overflowing into atomics)
    out = dict(rays_per_batch=8192, source="gfnerf/config.py:75 train_num_rays_per_batch")
    for stage, n_blocks, warm in (("focal", 4, 60), ("global", 0, 60)):  # (VERDICT r05 #5a: 60 warm-up steps for both: the phase plan and the CU hints have settled)
        run = StageRun(a2, stage, sc, 1, 0, sync, torch, n_blocks=n_blocks, n_rays=8192)
        res = run.run(steps, warm)
        roof, n_live, n_eval = run.roofline(res)
        ms = res["elapsed"] / steps * 1e3
        k_sum = float(sum(roof["kernel_ms"].values())) if isinstance(roof.get("kernel_ms"), dict) else None
        out[stage] = dict(ms_per_step=ms, rays_per_sec=8192 / (ms * 1e-3), mean_samples_per_ray=n_live / 8192, evaluated_sample_frac=n_eval / max(n_live, 1),
                          launch=run.launch, kernel_ms=roof["kernel_ms"], roofline_frac=roof["frac"], steps=steps, warmup=warm,
                          recaptures_in_timed_region=res["recaptures"], host_enqueue_ms=res["host_enqueue"] / steps * 1e3,
                          kernel_ms_sum=k_sum, kernel_to_step_ratio=None if not k_sum else k_sum / ms)
        del run, res
        torch.cuda.empty_cache()
    out.update(value=out["focal"]["rays_per_sec"], ms_per_step=out["focal"]["ms_per_step"], global_rays_per_sec=out["global"]["rays_per_sec"],
               global_ms_per_step=out["global"]["ms_per_step"], metric="train_rays_per_sec", unit="rays/s")
    return out


def trained_state_records(sc, args, torch, sync, seeds, steps_global, steps_focal, render_cams=None):
    """The headline step (and the global-stage step) timed on TRAINED engines: gfnerf_amd.procedural trains the global stage on the procedural terrain
    (the PSNR study's scene and schedule), the global step is timed there (learning rate held at the schedule's end), then four focal blocks train in
    rotation and the focal step is timed -- same StageRun / StepGraph / roofline code as the random-init records, batches = rays of the capture with
    their ground-truth colours.  -> dict with one record per seed; value / ms_per_step = the SLOWEST seed's focal step (the number DESIGN.md leads with
    when it is slower than the random-init headline)."""
    from gfnerf_amd import procedural

    dev = torch.device("cuda", torch.cuda.current_device())
    c2w = torch.from_numpy(sc["c2w"]).to(dev)
    recs = []

    def timed(hp, stage, seed):
        gen = torch.Generator(device=dev).manual_seed(9000 + seed)
        batches = []
        for _ in range(4):
            o, d, cam = procedural.device_rays(c2w, N_RAYS, gen)
            batches.append((o, d, cam, procedural.gt_colors(o, d, procedural.STUDY_DETAIL)))
        hp._drop_stale_samples()
        if args.prefetch_cus:  # (the engine came with procedural's configuration: the bench's switch applies to its timed steps too)
            hp.cfg.prefetch_cus = hp.cfg.prefetch_cus_global = hp.cfg.prefetch_cus_dense = hp.cfg.prefetch_cus_dense_eager = args.prefetch_cus
        run = StageRun(args, stage, sc, 1, 0, sync, torch, engine=dict(hp=hp, batches=batches))
        res = run.run(args.trained_steps, 10)
        roof, n_live, n_eval = run.roofline(res)
        w = res["aux"]["comp"]["weights"][:n_live]
        ms = res["elapsed"] / args.trained_steps * 1e3
        hp._drop_stale_samples()
        hp.phase_override = None
        return dict(ms_per_step=ms, rays_per_sec=N_RAYS / (ms * 1e-3), mean_samples_per_ray=n_live / N_RAYS, alive_sample_frac=float((w > 0).float().mean().item()),
                    evaluated_sample_frac=n_eval / max(n_live, 1), evaluated_samples_per_sec=n_eval / (ms * 1e-3), two_phase=run.phase, forward=hp.last_forward,
                    recaptures_in_timed_region=res["recaptures"], timed_steps=args.trained_steps,
                    launch=run.launch, fineness=run.fineness, loss=res["loss"],
                    roofline={k: roof[k] for k in ("kernel", "launches_per_step", "achieved", "peak", "unit", "frac", "avg_kernel_ms", "alg_bytes_per_launch", "kernel_ms")})

    for sd in seeds:
        rec = dict(seed=sd)

        def after_global(hp, lr_end):
            hp.cfg.lr = lr_end
            rec["global"] = timed(hp, "global", sd)

        r = procedural.train_and_eval(sc, seed=sd, steps_global=steps_global, steps_focal=steps_focal, log2_T=args.log2_T, detail=procedural.STUDY_DETAIL, lr_final=procedural.STUDY_LR_FINAL, n_blocks=4,
                                      keep_engine=True, after_global=after_global, progress=None, **procedural.LEGS["fast"])
        hp = r.pop("engine")
        hp.cfg.lr_block = hp.cfg.lr_block * procedural.STUDY_LR_FINAL  # the schedule's end
        rec["focal"] = timed(hp, "focal", sd)
        rec.update(psnr_global=r["psnr_global"], psnr=r["psnr"], leaves_pruned=r["leaves_pruned"], train_steps=[steps_global, steps_focal])
        if render_cams and "render" not in [k for x in recs for k in x]:  # the first seed's trained engine renders the headline record's frames
            try:
                rr = render_bench(hp, sc, args, rec["focal"]["fineness"], torch, render_cams, 0)
                rec["render"] = {k: rr[k] for k in ("value", "unit", "frames", "ms_per_frame", "tile_rays", "tiles_per_replay", "mean_accumulation", "fineness", "hbm_footprint_gb",
                                                    "decoder", "x3_ms_per_mpix", "colour_x2_vs_x3") if k in rr}
                if rr.get("roofline"):
                    rec["render"].update(evaluated_frac=rr["roofline"]["evaluated_frac"], samples_per_ray=rr["roofline"]["sampled_per_frame"] / (1080 * 1920),
                                         evaluated_samples_per_frame=rr["roofline"]["evaluated_samples_per_frame"])
            except Exception as e:  # noqa: BLE001
                rec["render"] = dict(error=f"{type(e).__name__}: {e}")
        recs.append(rec)
        del hp, r
        torch.cuda.empty_cache()
        print(f"[bench] trained state seed {sd}: focal {rec['focal']['ms_per_step']:.3f} ms (alive {rec['focal']['alive_sample_frac']:.2f}, evaluated "
              f"{rec['focal']['evaluated_sample_frac']:.2f}), global {rec['global']['ms_per_step']:.3f} ms", file=sys.stderr)
    slow = max(recs, key=lambda x: x["focal"]["ms_per_step"])
    slow_g = max(recs, key=lambda x: x["global"]["ms_per_step"])
    rend = next((x["render"] for x in recs if "render" in x and "error" not in x["render"]), {})
    return dict(metric="train_rays_per_sec", unit="rays/s", value=slow["focal"]["rays_per_sec"], ms_per_step=slow["focal"]["ms_per_step"], seed=slow["seed"],
                render_ms_per_mpix=rend.get("value"), render_evaluated_frac=rend.get("evaluated_frac"), render_samples_per_ray=rend.get("samples_per_ray"),
                render_x3_ms_per_mpix=rend.get("x3_ms_per_mpix"), render_colour_x2_max_abs_rgb=(rend.get("colour_x2_vs_x3") or {}).get("max_abs_rgb"),
                alive_sample_frac=slow["focal"]["alive_sample_frac"], evaluated_sample_frac=slow["focal"]["evaluated_sample_frac"],
                two_phase_K=(slow["focal"]["two_phase"] or {}).get("K"), roofline_frac=slow["focal"]["roofline"]["frac"],
                global_ms_per_step=slow_g["global"]["ms_per_step"], global_rays_per_sec=slow_g["global"]["rays_per_sec"],
                workload="the headline step (4 focal blocks resident, 1 active / step, 4096 x 192) and the global step (4096 x 128) on engines TRAINED on the procedural "
                         f"terrain ({steps_global} global + {steps_focal} focal steps, texture {procedural.STUDY_DETAIL}); headline keys = the slowest seed",
                seeds=recs)


# ------------------------------------------------------------------------------------------------ one timed training run
class StageRun:
    """K timed training steps of one stage on this rank (the same code for the headline and the `global_stage` sub-record)."""

    def __init__(self, args, stage, sc, world, rank, sync, torch, cams=None, n_blocks=0, block_seed=0, engine=None, n_rays=None):
        """engine: dict(hp=HotPath, batches=[(o, d, cam, gt) device tensors] x 4) -- time the steps on an engine that exists already (the
        `trained_state` sub-records: tables, decoder, embedding and pruned octree as gfnerf_amd.procedural left them) instead of a fresh
        random-init one."""
        from gfnerf_amd import hotpath, ops, scene

        self.args, self.stage, self.world, self.rank, self.torch, self.scene = args, stage, world, rank, torch, scene
        self.R = int(n_rays or N_RAYS)  # rays per batch (the contract's 4096; the `rays8192` sub-records: the reference's default, gfnerf/config.py:75)
        self.ops = ops
        self.trained = engine is not None
        if engine is not None:
            self.hp = hp = engine["hp"]
            self.cfg, self.n_blocks = hp.cfg, (len(hp.blocks) if stage == "focal" else 0)
            hp._record_buffers()
            self.batches, self.batches_np = list(engine["batches"]), None
            self.target = TARGET_SAMPLES[stage]
            self.fineness, self.mean_samples = pick_fineness(hp, self.batches[0][0], self.batches[0][1], self.target, scene, torch)
            self.draw_stream = ops.concurrent_stream(role="draws")
            self.drawn = {}
            self.i = 0
            return
        ga = args.lm_global_after
        self.cfg = cfg = hotpath.HotPathConfig(
            log2_T=args.log2_T, early_T=args.early_T, mlp_precision=args.mlp_precision, wgrad_planes=args.wgrad_planes,
            wgrad_beside_accumulate=args.wgrad_beside_accumulate,
            two_phase=not args.one_phase,
            **({"prefetch_cus": args.prefetch_cus, "prefetch_cus_global": args.prefetch_cus, "prefetch_cus_dense": args.prefetch_cus,
                "prefetch_cus_dense_eager": args.prefetch_cus} if args.prefetch_cus else {}),
            fused_backward=not args.unfused_backward, lds_accumulate=not args.atomic_backward, async_adam=args.async_adam,
            level_major_global=args.lm_global_forward, lm_halves_from=args.lm_halves_from, lm_split_log2=args.lm_split_log2, loss_before_fork=not args.fork_first,
            **({"record_capacity": args.record_capacity} if args.record_capacity else {}),
            **({"record_merge": args.record_merge} if args.record_merge >= 0 else {}),
            **({"level_major_global_after": ga} if ga is not None else {}))
        self.n_blocks = n_blocks
        # every rank builds the SAME replica (seed 0): the data-parallel global stage keeps them bit-identical from there on
        self.hp = hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, n_blocks=n_blocks, sync=sync)
        for i, b in enumerate(hp.blocks):  # mid-training state: non-zero residual tables (a rank's block is its own)
            g = torch.Generator(device="cuda").manual_seed(100 + i + 16 * block_seed)
            b.uniform_(-1e-3, 1e-3, generator=g)
        hp.refresh_shadows()
        # optimiser state and the gradient-record buffers exist before the first step (allocation is set-up, not training)
        if stage == "focal":
            for i, b in enumerate(hp.blocks):
                hp._adam_state(f"block{i}", b)
        else:
            hp._adam_state("table", hp.table)
        hp._record_buffers()
        # batches resident in HBM before the timed region (4 rotated batches per rank); cams = this rank's camera cluster
        self.batches_np, self.batches = [], []
        for k in range(4):
            o, d, cam = scene.make_rays(sc["c2w"], self.R, seed=1000 * rank + k + (0 if stage == "focal" else 500), cams=cams)
            gt = np.random.default_rng(7 + k).random((self.R, 3)).astype(np.float32)
            self.batches_np.append((o, d, cam, gt))
            self.batches.append(tuple(torch.from_numpy(a).cuda() for a in (o, d, cam, gt)))
        self.target = TARGET_SAMPLES[stage]
        self.fineness, self.mean_samples = pick_fineness(hp, self.batches[0][0], self.batches[0][1], self.target, scene, torch)
        if world > 1:  # one fineness for the whole job (rank 0's): replicas must march alike in the global stage
            import torch.distributed as dist

            f = torch.tensor([self.fineness], dtype=torch.float64, device=CTL_DEV)
            dist.broadcast(f, src=0)
            self.fineness = float(f.item())
        # The march noise (PersSampler_cuda.cu:380-389) and S3IM's arange + 9 permutations (losses.py:780-789) are drawn on the
        # device EVERY step, inside the timed region like the reference does: step i draws them for step i + 1 on a third stream
        # (they depend on nothing), beside the step's own kernels.
        self.draw_stream = ops.concurrent_stream(role="draws")
        self.drawn = {}
        self.i = 0

    def _draw(self, i):
        torch, hp = self.torch, self.hp
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.draw_stream):
            nz = self.ops.sampler_noise(self.R, self.fineness, train=True, device="cuda")
            s3 = self.ops.s3im_index(self.R, self.cfg.s3im_repeat_time, device="cuda")
        for t_ in (nz, s3):
            t_.record_stream(main)
            if hp._side is not None:
                t_.record_stream(hp._side)
        self.drawn[i] = (nz, s3)
        main.wait_stream(self.draw_stream)

    def step(self):
        i, args = self.i, self.args
        self.i += 1
        o, d, cam, gt = self.batches[i % 4]
        block = (i % self.n_blocks) if self.stage == "focal" else None
        nxt = self.batches[(i + 1) % 4]
        if i not in self.drawn:
            self._draw(i)
        self._draw(i + 1)
        noise, s3 = self.drawn.pop(i)
        pre = None if args.no_prefetch else (nxt[0], nxt[1], self.drawn[i + 1][0])  # the data loader knows the next batch
        # global stage: UpdateOctNodes K5/K6 every step, right after the forward like GFNeRFModel.get_outputs (nerfacto.py:598-616);
        # with N > 1 ranks train_step MAX-reduces the octree statistics and averages the gradients (HotPath.sync)
        return self.hp.train_step(o, d, cam, gt, noise, stage=self.stage, block=block, s3im_index=s3, prefetch=pre,
                                  update_oct=self.stage == "global" and not args.no_oct_update)

    def run_graph_experiment(self, steps, warmup, mode):
        torch, hp = self.torch, self.hp
        o, d, cam, gt = self.batches[0]
        noise = self.ops.sampler_noise(self.R, self.fineness, train=True, device="cuda")
        s3 = self.ops.s3im_index(self.R, self.cfg.s3im_repeat_time, device="cuda")
        block = 0 if self.stage == "focal" else None

        def one():
            return hp.train_step(o, d, cam, gt, noise, stage=self.stage, block=block, s3im_index=s3, prefetch=(o, d, noise),
                                 update_oct=self.stage == "global" and not self.args.no_oct_update)

        for _ in range(PRIMING + warmup):
            loss, aux = one()
        torch.cuda.synchronize()
        g = None
        if mode == "graph":
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss, aux = one()
                torch.cuda.current_stream().wait_stream(hp._side)  # the prefetched sampler is a branch of this step's graph
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if g is not None:
                g.replay()
            else:
                loss, aux = one()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(json.dumps(dict(experiment=mode, stage=self.stage, ms_per_step=el / steps * 1e3, host_enqueue_ms_per_step=host / steps * 1e3, loss=float(loss.item()))), file=_OUT)

    def _graph(self):
        """Capture the step (gfnerf_amd.stepgraph) after the eager warm-up; None when the step cannot be captured."""
        from gfnerf_amd import stepgraph

        hp, args = self.hp, self.args
        if args.launch != "graph" or args.no_prefetch or (self.world > 1 and self.stage == "global" and args.no_oct_update):
            return None, "eager (requested)"
        try:
            hp._drop_stale_samples()  # the last eager step prefetched a sample nobody will collect
            blocks = list(range(self.n_blocks)) if self.stage == "focal" else None
            sg = stepgraph.StepGraph(hp, self.stage, self.R, self.fineness, block=blocks, update_oct=self.stage == "global" and not args.no_oct_update,
                                     draws_after_lookup=not args.draws_first,
                                     segmented=True if (args.segmented and self.stage == "global") else None)
            o, d, _, _ = self.batches[self.i % 4]
            sg.prime(o, d)
            sg.p = self.i % sg.cycle  # the block rotation goes on where the eager steps stopped
            sg.capture()
            if sg.segmented:
                n_seg = sum(isinstance(x, self.torch.cuda.CUDAGraph) for x in sg.G[0])
                return sg, (f"hipGraph replay in {n_seg} segments per step, the data-parallel exchanges (octree MAX, table gradient, decoder gradients) "
                            f"eager between them; forward {hp.last_forward}")
            return sg, (f"hipGraph replay ({sg.cycle} captured graphs: sample-buffer parity x block rotation; forward {hp.last_forward}" +
                        ("" if sg.phased is None else ", phased" if sg.phased else ", one phase: the plan at capture found no split worth its launches") + ")")
        except Exception as e:  # noqa: BLE001 -- the eager path is always there
            hp._pre = None
            return None, f"eager (capture failed: {type(e).__name__}: {e})"

    def graph_step(self, sg):
        i = self.i
        self.i += 1
        _, _, cam, gt = self.batches[i % 4]
        nxt = self.batches[(i + 1) % 4]
        return sg.step(cam, gt, nxt[0], nxt[1])

    def run(self, steps, warmup, dist=None):
        """-> dict(elapsed (max over ranks), host_enqueue, loss, aux) for `steps` timed steps after PRIMING + warmup untimed ones.
        Launch mode graph: the eager warm-up is followed by the capture and GRAPH_WARMUP untimed replays; the timed steps are
        replays; then --probe-steps eager steps time the individual kernels with HIP events (a replay hides them)."""
        torch, hp, world = self.torch, self.hp, self.world
        for _ in range(PRIMING + warmup):
            self.step()
        sg, self.launch = self._graph()
        if sg is not None:
            for _ in range(GRAPH_WARMUP):
                self.graph_step(sg)
        events = {"*"} if self.args.stage_times else {"fused_fwd", "bwd_scatter", "grad_reduce", "hash_fwd", "hash_bwd", "field_fwd", "field_bwd", "wgrad", "adam"}
        hp.timings = {}
        hp.timed_ops = set() if sg is not None else events
        captures0 = sg.captures if sg is not None else 0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, aux = self.graph_step(sg) if sg is not None else self.step()
        host_enqueue = time.perf_counter() - t0  # when the host had launched everything: (well) below `elapsed` = the GPU is the bound
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=CTL_DEV)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = tmax.item()
        loss_value = float(loss.item())
        probe = steps
        if sg is not None:  # per-kernel event timings: eager steps right after the timed region
            hp._pre = None
            hp.timed_ops = events
            hp.phase_override = False if sg.phased is False else None  # ... with the forward the replays ran (StepGraph keeps its verdict to itself)
            probe = max(self.args.probe_steps, 1)
            for _ in range(probe):
                # the stream spins for a few ms first, so that every launch of the step is queued behind it: the events then
                # bracket GPU time only (with an empty queue and a slow host they would also count the host's launch gaps)
                torch.cuda._sleep(12_000_000)
                loss, aux = self.step()
            torch.cuda.synchronize()
        hp.timed_ops = set()
        # (a re-plan of the phases inside the timed region -- StepGraph.step: K idle for 16 replays, or replan_every -- is a synchronisation + capture
        # of every graph of the cycle, ~30 ms: it belongs to the product's step and stays in the time; the count says that it happened)
        return dict(elapsed=elapsed, host_enqueue=host_enqueue, loss=loss_value, aux=aux, steps=steps, event_steps=probe,
                    recaptures=(sg.captures - captures0) if sg is not None else 0)

    def roofline(self, res):
        """Dominant kernel by event time per step + its algorithmic bytes (DESIGN.md "Kernels")."""
        torch, hp, cfg, args = self.torch, self.hp, self.cfg, self.args
        aux, steps = res["aux"], res["event_steps"]
        n_live = aux["samples"]["pts_offsets"][-1].item()
        n_tables = 2 if self.stage == "focal" else 1
        # fused forward: only samples in front of each ray's termination point are looked up (sigma > 0 marks them); per such
        # sample 512 B of gathers per table + 16 B point/anchor + 8 B dt/t in, sigma 4 + rgb 12 + h0 4 + masks 48 + weight/alpha 8 out
        n_eval = int((aux["field"]["sigma"][:n_live] > 0).sum().item())
        self.phase = None
        if getattr(hp, "last_forward", "") == "lm2" and hp._phase is not None:  # the phased forward counts what it looks up and decodes itself
            nA, nB_rays, nB_samples, K = (int(v) for v in hp._phase.stats_last.tolist())
            n_eval = nA + nB_samples
            self.phase = dict(K=K, phase_a_samples=nA, phase_b_rays=nB_rays, phase_b_samples=nB_samples, next_K=int(hp._phase.k.item()))
        # hash lookup per table and sample = 512 B gathered + 12 B point + 8 B anchor + 128 B features written = 660 B
        # (unfused); hash scatter = 16 x 8 x 4 B packed-f16 adds + point/anchor (+ 56 B of decoder state in the fused kernel)
        alg = {"fused_fwd": n_eval * (512 * n_tables + 24 + 76), "hash_fwd": n_live * 660 * n_tables,
               "hash_bwd": n_live * (16 * 8 * 2 * (2 if cfg.grad_f16 else 4) + 128 + 16),
               "bwd_scatter": n_eval * (16 * 8 * 2 * (2 if cfg.grad_f16 else 4) + 16 + 16 + 56)}
        kern = {name: float(sum(a.elapsed_time(b) for a, b in ev)) / steps for name, ev in hp.timings.items()}
        dom = max((k for k in kern if k in alg), key=lambda k: kern[k])
        kern_ms, alg_bytes = kern[dom], alg[dom]
        # the symbols as rocprofv3's kernel trace prints them (profiles/*_kernel_stats.csv)
        kernel_name, launches = {"bwd_scatter": "field_bwd_dx_x3_kernel<true, 3>" if self.stage == "global" else "field_bwd_dx_x3_kernel<false, 3>",
                                 "hash_fwd": "hash_fwd_kernel", "hash_bwd": "hash_bwd_kernel", "fused_fwd": "fused_fwd_kernel<0, true, true>"}.get(dom, dom + "_kernel"), 1
        lm = getattr(hp, "last_forward", "").startswith("lm") and dom == "hash_fwd"
        if lm:  # level-major lookup: one launch of lm_lookup_kernel per table inside the timed "hash_fwd" region
            kernel_name, launches, dom = "lm_lookup_kernel<true, false> (+ <true, true>: the second table accumulates)" if n_tables == 2 else "lm_lookup_kernel<true, false>", n_tables, "lm_lookup"
            kern_ms = kern_ms / launches
            # the samples this launch looked up (all of them, or phase A's share: the first K of every ray)
            alg_bytes = (self.phase["phase_a_samples"] if self.phase else n_live) * 660
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # measured separately with rocprofv3 --pmc (see profiles/README.md)
        traffic, req = None, None
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            # the counters describe the kernel sources they were collected on: a figure measured on another version of a kernel is not
            # reported (tools/sync_profiles.py records the hashes with the counters)
            def fresh(group):
                files = pmc.get("source_sha256", {}).get(group)
                ok = bool(files) and all(_sha256(os.path.join(ROOT, "gf-nerf_amd", f)) == h for f, h in files.items())
                if not ok:
                    print(f"bench.py: profiles/pmc_traffic.json predates the {group} kernel sources: its counters are not reported", file=sys.stderr)
                return ok

            if not fresh("lm_lookup"):
                pmc = {k: v for k, v in pmc.items() if not k.startswith("lm_lookup") and not k.startswith("hash_fwd")}
            if not fresh("decoder"):
                pmc = {k: v for k, v in pmc.items() if not k.startswith("mfma_busy") and not k.startswith("field_fwd") and not k.startswith("bwd_scatter")}
            # the PMC passes were taken on the default configuration only (own primes per table, early_T 1e-4, T = 2^21)
            default_cfg = args.early_T == 1e-4 and args.log2_T == 21 and args.n_side == pmc.get("n_side", 20) and not args.one_phase and not self.trained
            traffic = pmc.get(f"{dom}:{self.stage}") if default_cfg else None
            # the counter table holds the mean over ALL launches of an instantiation: phase A's and phase B's, which returned at once
            mix = 2 if (lm and self.phase) else 1
            if traffic is not None:
                traffic *= mix
            n_req = pmc.get(f"{dom}:{self.stage}:fetch_requests") if default_cfg else None
            n_req = n_req * mix if n_req is not None else None
            if lm:  # the level-major kernel is bound by the L2 -> L1 line fills of its gathers (86 % of them hit the L2), not by the fabric
                looked_up = None
                if self.phase:  # the phased forward: the launch's share of the samples
                    looked_up = torch.zeros(n_live, dtype=torch.uint8, device="cuda")
                    looked_up[hp._phase.list_a[: self.phase["phase_a_samples"]].long()] = 1
                n_g = lm_issued_gathers(aux["samples"], n_live, cfg.lm_halves_from, torch, mask=looked_up)
                req = dict(unit="G gathers/s (4-byte, L2-resident range)", achieved=n_g / (kern_ms * 1e-3) / 1e9, peak=pmc["l2_gather_ceiling_per_s"] / 1e9,
                           frac=n_g / (kern_ms * 1e-3) / pmc["l2_gather_ceiling_per_s"], fabric_requests_per_launch=n_req,
                           nominal_gathers_per_launch=n_live * 128, issued_gathers_per_launch=n_g,
                           l2_to_l1=(dict(line_fills_per_launch=pmc["lm_lookup:focal:l1_miss_requests"] * mix, bytes_per_fill=128,
                                          achieved_TBps=pmc["lm_lookup:focal:l1_miss_requests"] * mix * 128 / (kern_ms * 1e-3) / 1e12, peak_TBps=34.5,
                                          l2_hit_rate=pmc["lm_lookup:focal:l2_hit_rate"],
                                          source="TCP_TCC_READ_REQ_sum, TCC_HIT / TCC_MISS (profiles/*_pmc_focal_per_kernel.csv); peak: MI355X guide, L2")
                                     if default_cfg and self.stage == "focal" and "lm_lookup:focal:l1_miss_requests" in pmc else None),
                           source="gathers the kernel issues per launch (16 levels x 8 corners per sample, minus the corners that runs of samples in one "
                                  "cell share below level lm_halves_from: counted from this batch) over the live kernel time; peak = tools/gather_bench.hip on a 4 MB table")
            elif n_req:  # what actually bounds a random 4-byte gather: 64-byte L2-miss requests per second
                req = dict(unit="G L2-miss requests/s", achieved=n_req / (kern_ms * 1e-3) / 1e9, peak=pmc["gather_request_ceiling_per_s"] / 1e9,
                           frac=n_req / (kern_ms * 1e-3) / pmc["gather_request_ceiling_per_s"],
                           source="FETCH_SIZE/64 (profiles/pmc_traffic.json) over the live kernel time; peak = tools/gather_bench.hip")
        # MFMA utilisation of the decoder (north_star: "MFMA utilisation on the MLP against gfx950 peak"): SQ_VALU_MFMA_BUSY_CYCLES over SIMD
        # cycles from the --pmc pass, and the rate of issued / useful matrix FLOP over the live kernel time against the 2.5 PFLOP/s dense f16 peak
        mfma = {}
        if os.path.exists(pmc_path) and "field_fwd" in kern:
            busy = pmc.get(f"mfma_busy:field_fwd:{self.stage}") if default_cfg else None
            useful = n_eval * 61952.0 / (kern["field_fwd"] * 1e-3) / 1e12  # 61,952 FLOP per decoded sample; x3 issues 3 MFMAs per product
            mfma = dict(mfma_busy_frac=busy, mfma_tflops_useful=useful, mfma_tflops_issued=3.0 * useful, mfma_peak_tflops=2500.0,
                        mfma_kernel="field_fwd_x3_kernel", mfma_kernel_ms=kern["field_fwd"])
        return dict(bound="hbm", kernel=kernel_name, launches_per_step=launches, achieved=alg_bytes / (kern_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=traffic, avg_kernel_ms=kern_ms, **mfma,
                    timing=("HIP events on the launch stream around every launch of the timed region" if res["event_steps"] == res["steps"] and self.launch.startswith("eager")
                            else f"HIP events on the launch stream in {res['event_steps']} eager steps run right after the timed region (the timed steps are hipGraph replays, "
                                 "which hide individual launches from events)"),
                    alg_bytes_per_launch=alg_bytes, kernel_ms={k: round(v, 4) for k, v in kern.items()}, request_rate=req), n_live, n_eval

    def print_stage_times(self, res):
        tot = 0.0
        for name, ev in self.hp.timings.items():
            t = sum(a.elapsed_time(b) for a, b in ev) / res["event_steps"]
            tot += t
            print(f"[stage] {self.stage:6s} {name:16s} {t:8.3f} ms/step ({len(ev) // res['event_steps']} launches)", file=sys.stderr)
        print(f"[stage] {self.stage:6s} {'sum':16s} {tot:8.3f} ms/step", file=sys.stderr)


def _r(v, n=5):
    """Floats to n significant digits (the line must stay short)."""
    return float(f"{v:.{n}g}") if isinstance(v, float) else v


def compact_line(out):
    """The line the driver parses: the contract's keys, `config` (which the driver keeps whole) with the headline numbers of every
    sub-record hoisted into it, a short `roofline` and `cpu_baseline`.  < 2,000 characters."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "samples_per_sec", "ranks_seen")
    line = {k: _r(out[k], 6) for k in keep if k in out}
    c = out["config"]
    cfg = {k: _r(c[k]) for k in ("workload", "rays_per_batch", "mean_samples_per_ray", "log2_table_size", "n_cameras", "n_tree_nodes", "stage", "early_T",
                                  "alive_sample_frac", "evaluated_sample_frac", "backend", "loss") if c.get(k) is not None}
    cfg["launch"] = "graph" if str(c.get("launch", "")).startswith("hipGraph") else "eager"
    cfg["parallelism"] = c["parallelism"].split(":")[0].split(",")[0]
    if c.get("two_phase"):
        cfg["two_phase_K"] = c["two_phase"]["K"]
    if c.get("allreduce_selfcheck"):
        cfg["allreduce_selfcheck_ok"] = c["allreduce_selfcheck"]["ok"]
    errors = {}

    def hoist(sub, pairs):
        r = out.get(sub)
        if not isinstance(r, dict):
            return
        if "error" in r:
            errors[sub] = r["error"][:120]
            return
        for src, dst in pairs:
            if r.get(src) is not None:
                cfg[dst] = _r(r[src])

    hoist("global_stage", (("ms_per_step", "global_ms_per_step"), ("value", "global_rays_per_sec"), ("launch_short", "global_launch"),
                           ("replicas_bit_identical", "replicas_bit_identical"), ("wgrad_operands", "global_wgrad_operands"), ("x3_ms_per_step", "global_x3_ms_per_step")))
    if "global_wgrad_operands" in cfg:
        cfg["global_wgrad_operands"] = str(cfg["global_wgrad_operands"]).split(" (")[0]  # ("f16 rne+dither": which plane keeps its lo' half is in [bench-details])
    if out.get("n_gpus", 1) == 1:
        cfg.pop("global_launch", None)  # (the N > 1 line says whether the data-parallel step replays graph segments; at N = 1 it is "graph" like the headline)
    hoist("render", (("value", "render_ms_per_mpix"), ("x3_ms_per_mpix", "render_x3_ms_per_mpix")))
    if isinstance(out.get("render"), dict) and isinstance(out["render"].get("colour_x2_vs_x3"), dict):
        cfg["render_decoder"] = "x3+colour_x2"  # (the colour head's layer 2 on one f16 activation operand: gfn_launch_opts.colour_x2, the tile renderer's default)
        cfg["render_x2_max_drgb"] = _r(out["render"]["colour_x2_vs_x3"]["max_abs_rgb"])  # max |rgb - rgb of the x3 form| over the last frame
    if isinstance(out.get("render"), dict) and isinstance(out["render"].get("roofline"), dict):
        rr = out["render"]["roofline"]
        if rr.get("mfma_busy_frac") is not None:
            cfg["render_mfma_busy_frac"] = _r(rr["mfma_busy_frac"])
    hoist("plugin_step_focal", (("ms_per_step", "plugin_ms_per_step"),))
    hoist("plugin_step_global", (("ms_per_step", "plugin_global_ms_per_step"),))
    hoist("plugin_step_trained", (("ms_per_step", "plugin_trained_ms_per_step"),))
    hoist("rays8192", (("value", "rays8192_rays_per_sec"), ("global_rays_per_sec", "rays8192_global_rays_per_sec")))
    hoist("batch_assembly", (("ms_per_batch", "batch_assembly_ms"),))
    hoist("psnr", (("psnr_fast", "psnr"), ("psnr_reference", "psnr_reference"), ("delta", "psnr_delta"), ("se", "psnr_se"), ("n_seeds", "psnr_n_seeds"),
                   ("decoder_delta", "psnr_decoder_delta"), ("study_delta", "psnr_study_delta"), ("study_se", "psnr_study_se"), ("study_n_seeds", "psnr_study_n_seeds")))
    hoist("trained_state", (("value", "trained_rays_per_sec"), ("ms_per_step", "trained_ms_per_step"), ("alive_sample_frac", "trained_alive_sample_frac"),
                            ("evaluated_sample_frac", "trained_evaluated_sample_frac"), ("two_phase_K", "trained_two_phase_K"),
                            ("roofline_frac", "trained_roofline_frac"), ("global_ms_per_step", "trained_global_ms_per_step"),
                            ("render_ms_per_mpix", "trained_render_ms_per_mpix"), ("render_evaluated_frac", "trained_render_evaluated_frac"),
                            ("render_x3_ms_per_mpix", "trained_render_x3_ms_per_mpix"), ("render_colour_x2_max_abs_rgb", "trained_render_x2_max_drgb")))
    hoist("cpu_reference_field", (("train_rays_per_sec", "cpu_torch_field_rays_per_sec"),))
    for k in ("mfma_busy_frac",):
        if out["roofline"].get(k) is not None:
            cfg[k] = _r(out["roofline"][k])
    line["config"] = cfg
    ro = out["roofline"]
    ro = line["roofline"] = {k: _r(ro[k]) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_kernel_ms", "alg_bytes_per_launch", "launches_per_step")
                             if k in ro}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = dict(value=_r(cb["value"]), unit=cb["unit"], cores=cb["cores"], kind=cb["kind"], sample=cb["sample"].split(",")[0][:80])
    if errors:
        line["errors"] = errors
    ro["kernel"] = str(ro.get("kernel", "")).split(" (")[0]
    # the driver keeps the tail of the log: the line must stay under 2,000 characters whatever sub-records ran (everything is in [bench-details])
    for k in ("cpu_torch_field_rays_per_sec", "batch_assembly_ms", "rays_per_batch", "loss", "global_rays_per_sec", "n_tree_nodes", "log2_table_size",
              "trained_roofline_frac", "psnr_decoder_delta", "render_x2_max_drgb", "trained_two_phase_K", "trained_alive_sample_frac",
              "render_x3_ms_per_mpix", "trained_render_evaluated_frac", "trained_evaluated_sample_frac", "plugin_global_ms_per_step"):
        if len(json.dumps(line)) <= 1985:
            break
        cfg.pop(k, None)
    return line


# ------------------------------------------------------------------------------------------------ one rank
_OUT = sys.stdout  # where the JSON line goes (run_rank points file descriptor 1 itself at stderr)


def _own_stdout():
    """The line on stdout must be the only thing there: libraries write to file descriptor 1 behind Python's back (gloo prints
    "[Gloo] Rank 0 is connected to ..." at rendezvous), so the descriptor is duplicated for the JSON line and 1 itself is
    pointed at stderr for everything else."""
    global _OUT
    sys.stdout.flush()
    _OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)


def run_rank(args):
    import torch
    import torch.distributed as dist

    _own_stdout()

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import _lib, cluster, dist as gdist, scene

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    n_dev = torch.cuda.device_count()
    backend = "nccl"
    selfcheck = None
    global CTL_DEV
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if n_dev < world:
            if not args.share_gpu:
                raise SystemExit(f"bench.py: {world} ranks but {n_dev} GPU(s) visible (--share-gpu rehearses ranks on shared cards)")
            backend, local_rank = "gloo", local_rank % max(n_dev, 1)  # RCCL refuses two ranks on one device
        torch.cuda.set_device(local_rank)
        if args.backend != "auto":
            backend = args.backend
        if backend == "nccl":
            # RCCL over xGMI.  Every rank has its own GPU here: if RCCL cannot be brought up that is an ERROR (non-zero exit), never
            # a quiet fall-back to gloo -- a line that says n_gpus = 8 must have exchanged its gradients over RCCL.
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)
            if int(probe.item()) != world:
                raise SystemExit(f"bench.py: RCCL all_reduce of ones over {world} ranks gave {probe.item()}")
            selfcheck = gdist.allreduce_selfcheck(device="cuda")
            if not selfcheck["ok"]:
                raise SystemExit(f"bench.py: the direct all-reduce (all-to-all + shard mean + all-gather) disagrees with dist.all_reduce: {selfcheck}")
        else:
            CTL_DEV = "cpu"
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    _lib.require_device()  # the HIP extension + a gfx950 device, or fail loudly
    sync = gdist.TrainSync()
    ranks_seen = sync.ranks_seen(device=CTL_DEV)

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
    # focal blocks = equal-size clusters of the training cameras (perssampler.py:216-242): 4 blocks on one GPU, one per rank beyond
    n_clusters = 4 if world == 1 else world
    labels = cluster.equal_size_clusters(sc["c2w"][:, :, 3], n_clusters, seed=1234)
    my_block = gdist.my_blocks(n_clusters, world, rank)[0] if world > 1 else None
    my_cams = cluster.cameras_of(labels, my_block) if world > 1 else None

    headline_stage = args.stage
    run = StageRun(args, headline_stage, sc, world, rank, sync, torch,
                   cams=my_cams if headline_stage == "focal" else None,
                   n_blocks=(4 if world == 1 else 1) if headline_stage == "focal" else 0, block_seed=rank)
    if args.graph_experiment != "off":
        run.run_graph_experiment(args.steps, args.warmup, args.graph_experiment)
        return
    res = run.run(args.steps, args.warmup, dist)
    if rank == 0 and args.stage_times:
        run.print_stage_times(res)
    roofline, n_live, n_eval = run.roofline(res)
    cfg, hp = run.cfg, run.hp
    out = None
    if rank == 0:
        ms = res["elapsed"] / args.steps * 1e3
        value = N_RAYS * world * args.steps / res["elapsed"]
        w = res["aux"]["comp"]["weights"][:n_live]
        alive_frac = float((w > 0).float().mean().item()) if w is not None else None
        out = dict(metric="train_rays_per_sec", value=value, unit="rays/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=ms, higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="f32 (split-f16 x3 MFMA decoder, f16 table)" if cfg.mlp_precision == "x3" else "f32 (f16 table)", data="synthetic",
                   samples_per_sec=value * n_eval / N_RAYS,  # the samples the forward looked up and decoded (the batch holds n_live: batch_samples_per_sec)
                   batch_samples_per_sec=value * n_live / N_RAYS, ranks_seen=ranks_seen,
                   config=dict(workload=("GF-NeRF global+focal train step, 4 focal blocks resident (1 active/step), "
                                         if headline_stage == "focal" and world == 1 else
                                         "GF-NeRF focal stage, 1 block per GPU, rays from the block's camera cluster, " if headline_stage == "focal" else
                                         "GF-NeRF global stage train step (ray data parallel), ") +
                                        f"4096 rays x {run.target} samples, Rubble-scale synthetic cameras",
                               rays_per_batch=N_RAYS, mean_samples_per_ray=n_live / N_RAYS, log2_table_size=args.log2_T,
                               n_cameras=n_cams, n_cameras_spec=1600, n_tree_nodes=int(sc["n_nodes"]), n_warp_transforms=int(sc["n_trans"]),
                               fineness=run.fineness, stage=headline_stage, early_T=cfg.early_T, alive_sample_frac=alive_frac,
                               evaluated_sample_frac=n_eval / max(n_live, 1), evaluated_samples_per_sec=value * n_eval / N_RAYS,
                               fused_forward=bool(cfg.fused_forward),
                               block_primes="own per table (reference)",
                               mlp=("split-f16 x3 MFMA 32x32x16 (f32 accuracy)" if cfg.mlp_precision == "x3" else "f32 MFMA 32x32x2"),
                               parallelism=("single GPU" if world == 1 else
                                            f"{world} focal blocks, 1 per GPU, {len(my_cams)} cameras per block, no data-path collective" if headline_stage == "focal"
                                            else f"dp{world}: MAX all-reduce of octree statistics + direct all-reduce (all-to-all, shard sum, all-gather) of table / decoder / embedding gradients"),
                               backend=backend if world > 1 else None,
                               scene_build_s=round(t_build, 1), priming_steps=PRIMING, launch=run.launch,
                               graph_warmup_steps=GRAPH_WARMUP if run.launch.startswith("hipGraph") else 0,
                               host_enqueue_ms_per_step=res["host_enqueue"] / args.steps * 1e3,
                               async_adam=bool(cfg.async_adam), per_step_draws="march noise + S3IM permutations drawn on the device every timed step",
                               two_phase=run.phase, allreduce_selfcheck=selfcheck, loss=res["loss"], recaptures_in_timed_region=res["recaptures"]),
                   roofline=roofline)

    state = dict(pending=None)

    def emit(expired=False):
        """The ONE line on stdout: the contract's keys + `config` with every headline number of the sub-records hoisted into it
        (short enough for a log tail); the sub-records in full go to stderr (`[bench-details] {...}`)."""
        if not (rank == 0 and out is not None):
            return
        if expired and state["pending"]:
            out[state["pending"]] = dict(error=f"timed out after {args.sub_timeout:.0f} s")
        print("[bench-details] " + json.dumps(out), file=sys.stderr)
        sys.stderr.flush()
        print(json.dumps(compact_line(out)), file=_OUT)
        _OUT.flush()

    # ---- configs[4]: render (every rank renders the frames of the block it owns)
    if args.render_frames > 0:
        if world == 1:
            path = [(37 * (f + 1)) % n_cams for f in range(args.render_frames)]
            owners = None
            rblock = 0 if (headline_stage == "focal" and hp.blocks) else None
        else:  # render_frames frames per block, each routed to the block's owner (perssampler.py:369-376: the frame's block = the nearest camera's)
            path, owners = [], []
            for b in range(n_clusters):
                cb = cluster.cameras_of(labels, b)
                for f in range(args.render_frames):
                    path.append(int(cb[(37 * (f + 1)) % cb.size]))
            owners = gdist.frame_owner(labels[np.asarray(path)], world)
            rblock = 0 if (headline_stage == "focal" and hp.blocks) else None
        state["pending"] = "render"
        try:
            with Deadline(args.sub_timeout if world > 1 else 1e9, lambda: emit(expired=True)):
                r = render_bench(hp, sc, args, run.fineness, torch, path, rblock, world, rank, owners, dist)
        except Exception as e:  # noqa: BLE001 -- the headline stands on its own; the failure is in the line and in the exit code
            if world == 1:
                raise
            r = dict(error=f"{type(e).__name__}: {e}")
        state["pending"] = None
        if out is not None:
            out["render"] = r
    if world == 1 and args.render_frames > 0:
        out["batch_assembly"] = batch_assembly_bench(sc, torch)
    P_focal = hp.export_params() if (world == 1 and not args.no_cpu_baseline) else None
    fine_focal, batches_focal = run.fineness, run.batches_np
    # ---- configs[1]: the global stage under the same clock (sub-record); data parallel at N > 1
    grun = None
    identical_after_global = [None]
    if args.global_steps > 0 and headline_stage != "global":
        del run, hp, res
        torch.cuda.empty_cache()

        def global_record():
            nonlocal grun
            grun = StageRun(args, "global", sc, world, rank, sync, torch, cams=None, n_blocks=0)
            gres = grun.run(args.global_steps, args.global_warmup, dist)
            if rank == 0 and args.stage_times:
                grun.print_stage_times(gres)
            groof, g_live, g_eval = grun.roofline(gres)
            if args.inject_divergence and world > 1 and rank == 1:
                grun.hp.table.view(-1)[12345] += 1.0
            identical = sync.replicas_identical([grun.hp.table, grun.hp.mlp_flat, grun.hp.emb, grun.hp.nodes]) if world > 1 else None
            identical_after_global[0] = identical
            if out is not None:
                gv = N_RAYS * world * args.global_steps / gres["elapsed"]
                out["global_stage"] = dict(
                    metric="train_rays_per_sec", value=gv, unit="rays/s", ms_per_step=gres["elapsed"] / args.global_steps * 1e3,
                    steps=args.global_steps, warmup=args.global_warmup, samples_per_sec=gv * g_live / N_RAYS,
                    workload="GF-NeRF global stage, single Hash3DAnchored encoder + tiny MLP, 4096 rays x 128 samples (BASELINE configs[1]); all parameters "
                             "train, UpdateOctNodes (K5/K6) every step", mean_samples_per_ray=g_live / N_RAYS, evaluated_sample_frac=g_eval / max(g_live, 1),
                    forward=grun.hp.last_forward, fineness=grun.fineness, launch=grun.launch, two_phase=grun.phase,
                    launch_short="graph" if grun.launch.startswith("hipGraph") else "eager",
                    host_enqueue_ms_per_step=gres["host_enqueue"] / args.global_steps * 1e3,
                    parallelism="single GPU" if world == 1 else f"dp{world}: octree statistics MAX-reduced, gradients averaged (direct all-reduce) inside HotPath.train_step",
                    replicas_bit_identical=identical, sync=dict(sync.stats) if world > 1 else None, loss=gres["loss"], roofline=groof,
                    # (VERDICT r05 #5b) what the decoder's WEIGHT gradient is contracted from in this record: narrower than the reference's f32 nn.Linear
                    # backward when "rne" (one f16 plane per operand, round to nearest + per-sample dither; f64 bar 1e-4 of max |dW|); the per-sample chains
                    # (forward, d x, table gradient) are split-f16 x3 either way
                    wgrad_operands={"rne": "f16 rne+dither (dza1 hi + lo')", "x3": "split f16 hi + lo' (x3)", "mixed": "dz f16 rne, activations hi + lo'"}[args.wgrad_planes])
                if world == 1 and args.wgrad_planes != "x3" and args.global_x3_steps > 0:  # ... and the same step with x3 operands, beside it
                    import copy

                    a3 = copy.copy(args)
                    a3.wgrad_planes = "x3"
                    g3 = StageRun(a3, "global", sc, 1, 0, sync, torch, cams=None, n_blocks=0)
                    r3 = g3.run(args.global_x3_steps, args.global_warmup)
                    out["global_stage"]["x3_ms_per_step"] = r3["elapsed"] / args.global_x3_steps * 1e3
                    del g3, r3
                    torch.cuda.empty_cache()

        state["pending"] = "global_stage"
        try:
            with Deadline(args.sub_timeout if world > 1 else 1e9, lambda: emit(expired=True)):
                global_record()
        except Exception as e:  # the headline stands on its own; the failure is in the line and in the exit code
            if out is not None:
                out["global_stage"] = dict(error=f"{type(e).__name__}: {e}")
            print(f"bench.py: global_stage sub-record failed: {type(e).__name__}: {e}", file=sys.stderr)
        state["pending"] = None
    # ---- the same steps THROUGH the plugin surface (pipeline + Optimizers, the reference Trainer's call sequence), N = 1
    if out is not None and world == 1 and args.plugin_steps > 0:
        if grun is not None:
            P_global, fine_global, batches_global = grun.hp.export_params(), grun.fineness, grun.batches_np
        del grun
        grun = None
        torch.cuda.empty_cache()
        for stage_name in ("focal", "global"):
            try:
                out[f"plugin_step_{stage_name}"] = plugin_step_bench(sc, torch, stage_name, steps=args.plugin_steps, warmup=30 if stage_name == "focal" else 60)
            except Exception as e:  # noqa: BLE001
                out[f"plugin_step_{stage_name}"] = dict(error=f"{type(e).__name__}: {e}")
            torch.cuda.empty_cache()
    # ---- PSNR: the engine trained on a procedural scene, split-f16 decoder beside the exact-f32 one (the reference's arithmetic)
    pg, pf = (int(v) for v in args.psnr_steps.split(","))
    # ---- the same steps on a TRAINED state: the random-init scene is a uniform fog in which 43 % of the samples lie behind their
    # ray's termination point; a trained scene has surfaces and free space
    trained_seeds = [int(v) for v in args.trained_seeds.split(",") if v.strip() != ""]
    if out is not None and world == 1 and trained_seeds and pg > 0 and args.stage == "focal":
        try:
            out["trained_state"] = trained_state_records(sc, args, torch, sync, trained_seeds, pg, pf,
                                                         render_cams=[(37 * (f + 1)) % n_cams for f in range(args.render_frames)] if args.render_frames > 0 else None)
        except Exception as e:  # noqa: BLE001
            out["trained_state"] = dict(error=f"{type(e).__name__}: {e}")
        torch.cuda.empty_cache()
    # ---- the plugin surface on a TRAINED scene, trained through the surface itself
    if out is not None and world == 1 and args.plugin_steps > 0 and pg > 0 and pf >= args.plugin_steps + 100 and args.stage == "focal":
        try:
            out["plugin_step_trained"] = plugin_step_bench(sc, torch, "focal", steps=args.plugin_steps, warmup=30, trained=(pg, pf))
        except Exception as e:  # noqa: BLE001
            out["plugin_step_trained"] = dict(error=f"{type(e).__name__}: {e}")
        torch.cuda.empty_cache()
    # ---- the reference's default batch of 8192 rays (a sub-record; the headline stays on the contract's 4096)
    if out is not None and world == 1 and args.rays8192_steps > 0 and args.stage == "focal":
        try:
            out["rays8192"] = rays8192_records(sc, args, torch, sync, steps=args.rays8192_steps)
        except Exception as e:  # noqa: BLE001
            out["rays8192"] = dict(error=f"{type(e).__name__}: {e}")
        torch.cuda.empty_cache()
    # ---- PSNR: the engine trained on the procedural scene by the product's leg and by the reference-semantics leg, paired by seed
    if out is not None and world == 1 and pg + pf > 0 and args.psnr_seeds > 0:
        from gfnerf_amd import procedural

        try:
            ps = procedural.psnr_study(sc, seeds=tuple(range(args.psnr_seeds)), legs=("fast", "reference"), steps_global=pg, steps_focal=pf, log2_T=args.log2_T,
                                       detail=procedural.STUDY_DETAIL, lr_final=procedural.STUDY_LR_FINAL, holdout_every=8, progress=lambda m: print(f"[bench] {m}", file=sys.stderr))
            ps.update(psnr_study_record(pg, pf))
            out["psnr"] = ps
        except Exception as e:  # noqa: BLE001
            out["psnr"] = dict(error=f"{type(e).__name__}: {e}")
        torch.cuda.empty_cache()
    if out is not None and not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
        out["cpu_reference_field"] = cpu_reference_field()
        out["cpu_baseline"] = cpu_baseline(P_focal, batches_focal[0], args.log2_T, fine_focal, args.cpu_rays,
                                           0 if headline_stage == "focal" else None, scene)
        if "P_global" in locals():
            out["cpu_baseline_global"] = cpu_baseline_global(P_global, batches_global[0], args.log2_T, fine_global, scene)
    emit()
    failed = [k for k, v in (out or {}).items() if isinstance(v, dict) and "error" in v]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failed:  # a sub-record that failed is a failed run, whatever the headline says
        print(f"bench.py: failed sub-records: {failed}", file=sys.stderr)
        sys.exit(4)
    # data-parallel replicas that are not bit-identical after the timed steps: the exchanges lost or reordered something.  Every rank
    # holds the same verdict (an all-reduced checksum comparison), so every rank leaves with the same code
    if identical_after_global[0] is False:
        print("bench.py: the data-parallel replicas DIVERGED (replicas_bit_identical false)", file=sys.stderr)
        sys.exit(5)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()

"""The per-ray hot path, end to end, on one MI355X.

Mirrors the call order of GFNeRFModel.get_outputs (gfnerf/nerfacto.py:522-619):
  PersSampler (trace + march)  ->  Hash3DAnchored (+ residual block table in the focal
  stage, nerfacto_field.py:477-489)  ->  base_network  ->  trunc_exp density + SH/appearance
  head input  ->  mlp_head  ->  get_weights_f2nerf + renderers,
and for training the matching backward, the Charbonnier loss and Adam.  Every stage is a
call through the C ABI into hand-written HIP kernels (ops.py); samples stay PACKED and the
live sample count stays on the device (no host synchronisation inside a step).
"""
import collections
import math
import weakref
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, dist as gdist, ops, scene

STAGE_GLOBAL = "global"  # the reference's 'init_stage': global table + MLPs + embedding train
STAGE_FOCAL = "focal"    # 'block_stage': only the active block's residual table trains


@dataclass
class HotPathConfig:
    log2_T: int = 21                 # config.py:119
    hidden: int = 128                # config.py:124-125
    sample_l: float = 1.0 / 256      # perssampler.py:64
    scale_by_dis: bool = True        # perssampler.py:62
    global_near: float = 0.01        # perssampler.py:61
    max_oct_per_ray: int = 1024      # perssampler.py:60
    max_samples_per_ray: int = 1024  # PersSampler_cuda.cu:9 (capacity of the packed buffers per ray)
    early_T: float = 1e-4            # early-termination transmittance (0 = reference: none)
    scale_factor: float = 10.0       # config.py:49
    lr: float = 1e-2                 # config.py:130
    lr_block: float = 5e-3           # nerfacto.py:448-488 (block tables)
    adam_eps: float = 1e-15
    use_appearance_embedding: bool = True  # config.py:50
    table_f16_shadow: bool = True    # gather from f16 shadows refreshed by the Adam kernel (same values as :185's cast)
    grad_f16: bool = True            # table gradient pool in f16 with the x128 loss scale, like the reference (:151,:209)
    fused_field: bool = True         # decoder as one kernel per direction (needs hidden == 128)
    s3im_loss_mult: float = 1.0      # gfnerf/config.py:96 (0 disables the S3IM term of get_loss_dict, nerfacto.py:687-688)
    s3im_patch_height: int = 32      # config.py:100
    s3im_repeat_time: int = 10       # config.py:99
    fused_forward: bool = True       # lookup + decoder + compositing in one persistent kernel that leaves a ray once it is opaque
    fused_backward: bool = True      # decoder backward + hash scatter in one kernel (d feat never written)
    keep_features: bool = False      # also write the [N,32] hash features from the fused forward (tests / debugging)
    level_major: bool = True         # training forward: level-major lookup (L2-resident address ranges) + decoder + compositing
                                     # as three kernels instead of the one fused kernel (which stays the inference path)
    lds_accumulate_exact: bool = True  # exact fixed-point LDS sums (rounded to f16 once) where they provably fit; False: the reference's f16 adds
    level_major_global: bool = False # also in the global stage (one table), from the first step on
    level_major_global_after: int = 50  # ... or from this optimiser step on (-1: never): while the scene is empty the fused forward's
                                     # early termination wins (2.40 vs 2.69 ms/step); once nearly every sample is alive (a few dozen
                                     # steps) the level-major order does (2.77 vs 3.03 ms/step over 300 steps)
    lm_halves_from: int = 6          # levels >= this are gathered one address part at a time ...
    lm_split_log2: int = 0           # ... of 2^this parts per level (0 = 1: halves; 2: quarters, no gain inside the step)
    render_patch_width: int = 8      # inference (TileRenderer): patches of this x 32/this neighbouring pixels walked in lockstep by one wave
                                     # (gfn_fused_render); 0 = a wave per ray like the training forward
    render_colour_x2: bool = True    # inference (TileRenderer): the colour head's layer 2 (128 -> 128) with a one-f16 activation operand (ops.launch_opts(colour_x2=True):
                                     # two MFMAs per product instead of three; density, accumulation and depth unchanged bit for bit, rgb by that layer's f16
                                     # rounding -- DESIGN.md section 4 has the measured distance (5.5e-5 ... 1.3e-4 at the worst pixel of full-HD frames of trained scenes, 2.1e-4 the
                                     # worst case met: north_star's bar is 1e-3) and time (-8 %); False: the x3 form throughout
    graph_forward: bool = True       # callers that hand forward_train_graphed static sample buffers (the plugin surface) get the forward as one
                                     # hipGraph replay per (stage, block, buffer) after three eager runs: ~25 launches less for the host
    graph_forward_shared_pool: bool = True  # the captured forwards share one memory pool (their transient buffers overlap; replays are sequential)
    graph_backward: bool = True      # ... and the backward of such a replayed forward (compositing backward, decoder backward + record scatter, LDS accumulation)
                                     # as a second replayed graph on the same static buffers (HotPath.backward_graphed; focal stage)
    graph_one_phase: bool = True     # ... captured in its ONE-PHASE form (every sample, no lists: eight graph nodes instead of twenty-one, each of which costs a
                                     # replay ~6 us) while the planner's K says that a split does not pay -- a trained scene, where nearly every sample lies in
                                     # front of its ray's termination point.  The plan is looked at again every graph_one_phase_probe forwards (one eager phased
                                     # forward: its K travels to the host through a pinned word) and the phased capture comes back when K drops
    graph_one_phase_probe: int = 512
    graph_forward_keep: int = 4      # captured forwards kept (least recently used go first): each holds a full set of forward activations
                                     # (150-200 MB at 4096 x 192); 4 = two sample buffers x (the active block + the one before it)
    async_adam: bool = False         # focal stage: Adam (+ zeroing the pool) on a third stream beside the next step's lookup of the
                                     # frozen global table (HBM-bound beside address-path-bound)
    loss_before_fork: bool = True    # forward_backward enqueues the loss kernels before it forks the octree update / next sampler
    lds_accumulate: bool = True      # backward: write the table adds as records, sort + accumulate them in LDS (grad_part.hip)
    record_capacity: int = 1 << 29   # 8-byte record slots (4 GiB; 1,927 per (range, workgroup) stream at T = 2^21) -- a full stream falls back to atomics, at a
                                     # fifth of the rate.  The global stage writes 48 M records per step, a focal step 13-17 M on a young table (most
                                     # contributions round to zero at the pool's f16 x128 scale) and up to 90 M once every sample carries a gradient
                                     # (measured through the plugin surface on a scene trained there: bench.py plugin_step_trained)
    sampler_stream_priority: int = 0  # -1: the second stream (next batch's draw + sampler) as a HIGH-priority stream.  Measured in round 6 and left off: no gain
                                     # where it was meant to help (the plugin iteration: what the sampler lacks beside a backward on every CU is room, not
                                     # rank -- prefetch_cus_dense_replay), and the eager steps' lookups and Adam 20-25 % slower beside a sampler that outranks them
    prefetch_cus: int = 160          # CUs left to the fused backward while a prefetched sampler runs beside it (focal stage; 192 until the backward
                                     # walked the front list: since then the sampler, not the backward, ends last -- tools/sweep_cus.sh)
    prefetch_cus_global: int = 192   # the same in the global stage (its backward is longer: the sampler needs fewer CUs)
    prefetch_dense_frac: float = 0.75  # focal stage, a DENSE batch -- at least this share of its samples in front of the termination points (a trained
                                     # scene: 0.87-0.95 of 4096 x 192; the random-init fog has 0.57, a half-pruned scene 0.61-0.65) -- has a backward two
                                     # to three times as long as the sampler beside it, which then needs fewer CUs (round 5, tools/README.md: native replay
                                     # 1.57 -> 1.52 ms at 176-192 CUs, a cliff from 200 on; through the plugin surface, eager launches with the sampler
                                     # enqueued first, 2.30 -> 2.21 ms with no cap at all).  The two counts are last step's (a pinned pair the forward's side
                                     # branch copies, read late) or the capture's planning forward's (StepGraph)
    prefetch_cus_dense: int = 184    # ... CUs of the backward then, in a captured step
    prefetch_cus_dense_eager: int = 256  # ... and launched eagerly (256: no cap)
    prefetch_cus_dense_replay: int = 184  # ... and as the plugin surface's replayed backward (HotPath.backward_graphed), beside a sampler launched eagerly on the
                                     # second stream: with every CU (and register file) taken by the persistent backward the sampler's traversal and march only
                                     # START when it ends, and the next forward waits 0.22 ms for them (round 6, a scene trained through the surface, ms per
                                     # iteration at 256 / 224 / 216 / 208 / 200 / 192 / 184 / 160 CUs: 2.11 / 2.12 / 2.08 / 2.07 / 2.05 / 1.99 / 1.96 / 2.13; tools/README.md)
    record_merge: int = -1           # focal backward in record form: 0 / 1 = plain / run-merging records for EVERY launch (experiments: bench.py --record-merge);
                                     # -1: the policy below
    record_merge_dense: bool = False  # ... and such a batch merges runs of samples in one cell before it writes its gradient records -- decided on the DEVICE
                                     # from the step's own counts (gfn_launch_opts GFN_MERGE_DEVICE; needs the phased forward's front list).  On through the plugin
                                     # surface (gfnerf/nerfacto.py: 90 -> 56 M records per step and 2.22 -> 2.10 ms per iteration on a scene trained there);
                                     # off here because it is neutral on the natively trained scenes and a merged run's sum is rounded to f16 once more: the
                                     # list-walking and the every-sample form of the backward would no longer give the same bits (StepGraph against eager steps)
    two_phase: bool = True           # level-major training forward with early_T > 0: look up / decode only the first K samples of every ray,
                                     # then the rest of the rays that are still transparent (none, as a rule); K is planned on the device
                                     # from where the previous batch's rays ended (ops.two_phase_plan).  False: every sample
    two_phase_fixed_cost: float = 150e3  # what starting phase B costs, in samples (two table sweeps + small kernels ~ 0.15 ms)
    two_phase_min_saving: tuple = (0.12, 0.06)  # samples a split must save to be started / to be kept (break-even ~ 9 %: measured);
    two_phase_min_saving_planes: tuple = (0.30, 0.20)  # ... in the global stage: compacted samples write unaligned activation-plane rows
    mlp_precision: str = "x3"        # fused decoder arithmetic: "x3" split-f16 MFMA (f32 accuracy), "f32" exact f32 MFMA
    wgrad_planes: str = "rne"        # operands of the decoder's WEIGHT gradient (global stage; dW = sum over the batch's 5e5 samples of dz a):
                                     # "rne" one f16 plane per operand, rounded to nearest under a per-sample dither that makes neighbouring samples'
                                     # rounding errors independent (zero-mean per term, averages out in the sum; half the plane traffic, one MFMA
                                     # per fragment pair; dza1 alone keeps its lo' half: csrc/x3_device.h); "x3" hi + lo' planes for both (rounds 1-4:
                                     # f32-class per term); "mixed" dz one plane, activations hi + lo'.  The per-sample chains (forward,
                                     # d x = W^T dz, the table gradient) are x3 whatever this says
    plane_dither: bool = True        # ... the one-plane producers' per-sample dither (False: plain rounding; tests measure what it buys)
    wgrad_beside_accumulate: bool = False  # global stage, one rank: the decoder's weight-gradient kernel on a second stream beside the LDS accumulation
                                     # of the table gradient's records (both read what the backward wrote, neither the other's result)
    exact_oct_alphas: bool = True    # a training forward that UpdateOctNodes follows evaluates the density of EVERY sample, so that K5 reads true alphas
                                     # behind the rays' termination points like the reference does (forward_train's all_alphas); False: rounds 2-3
                                     # (alpha = 0 where the forward skipped a sample: a node's alpha statistic then decays whenever only occluded rays reach it)


class HotPath:
    def __init__(self, nodes, transes, n_cams, cfg=None, device=None, seed=0, n_blocks=0, sync=None):
        """device: None = the process's current HIP device (one process per GPU: torch.cuda.set_device(LOCAL_RANK)).
        sync: gfnerf_amd.dist.TrainSync of this process (default: one over the default process group, the identity when
        torch.distributed is not initialised): the global stage exchanges octree statistics (MAX) and gradients (mean)
        through it, so N ranks that start from the same seed stay bit-identical replicas."""
        self.cfg = cfg or HotPathConfig()
        self._seed = seed
        self.sync = sync if sync is not None else gdist.TrainSync()
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)  # noqa: E731

        def blob(a):  # numpy blob or a device uint8 tensor (shared with a PersSampler)
            if isinstance(a, torch.Tensor):
                return a.to(self.dev).contiguous().view(torch.uint8).reshape(-1)
            return t(np.asarray(a).view(np.uint8).reshape(-1))

        self.nodes = blob(nodes)
        self.transes = blob(transes)
        self.n_nodes = self.nodes.numel() // ops.TREE_NODE_BYTES
        self.nodes_compact = ops.tree_compact(self.nodes)  # 64-byte lines for the fast sampler
        self._sampler_ws = None
        self._record_ws = None  # hit lists of sample(record=True) when the packed form's workspace does not exist or is too small
        self._depth_key, self._depth_val = None, 0
        self._records = None
        self._opt_stream, self._opt_pending = None, False
        self.k_host = None          # see _forward_phased
        self._front_host = torch.zeros(2, dtype=torch.int32).pin_memory()  # [samples in front of the termination points, samples] of the last phased focal forward (_dense_hint)
        self.front_override = None  # StepGraph.capture: (in front, samples) of its planning forward, for the launches it captures
        self._aux_stream, self._aux_pending = None, False  # branch of the phased forward: front list + plan beside the loss
        self.after_lookup = None    # called once the training forward's table lookups are enqueued (the kernels that follow are
                                    # matrix-bound: side work started here does not slow the gathers down; gfnerf_amd.stepgraph)
        self._wgrad_parts = None    # per-workgroup partial weight gradients of the decoder (ops.wgrad_x3_layers)
        self._wg_stream = None      # cfg.wgrad_beside_accumulate: the stream of the weight-gradient kernel
        self._side, self._pre = None, None  # sampler of the NEXT batch on a second stream (prefetch_sample)
        self._deferred = []
        self.last_forward_replayed = False
        self.last_backward_replayed = False
        self._backward_eager_like = False
        self._one_phase, self._one_phase_uses, self._one_phase_probing = False, 0, 0  # forward_train_graphed: the captured forward's form (cfg.graph_one_phase)
        self._fwd_graphs = collections.OrderedDict()  # forward_train_graphed: key -> dict(n, graph, cam, comp, ctx, last_forward), least recently used first
        self._fwd_pool = None       # one memory pool for every captured forward (their transient buffers share it; replays are sequential)
        self.seg = None             # set by a segmented StepGraph capture: seg(fn) ends the graph segment being captured, notes fn as the
                                    # eager call between segments (a collective) and begins the next segment
        self.n_volumes = self.n_nodes  # nerfacto.py:268: one hash "volume" slot per tree node
        T = self.cfg.log2_T
        self.prim = t(scene.make_primes(self.n_volumes, seed=seed + 2))
        self.bias = torch.zeros((16 * self.n_volumes, 3), dtype=torch.float32, device=self.dev)
        self.bias_arg = None  # the pool is all zeros (rand_bias off): kernels skip the loads; set to self.bias otherwise
        self.table = t(scene.make_feat_pool(T, seed=seed + 2))  # base_encoding_init (Reset: U(-1e-2,1e-2))
        # residual block tables start at zero in the reference (nerfacto_field.py:345); tiny noise here
        # would hide nothing in training but zeros would hide kernel bugs in tests -> caller may overwrite
        self.blocks = [torch.zeros((16 * (1 << T), 2), dtype=torch.float32, device=self.dev) for _ in range(n_blocks)]
        # every Hash3DAnchored draws its own primes (Hash3DAnchored.cpp:43-50; add_table, nerfacto_field.py:341-347)
        self.block_prims = [t(scene.make_primes(self.n_volumes, seed=seed + 100 + i)) for i in range(n_blocks)]
        base, head = scene.make_mlps(seed=seed + 4, hidden=self.cfg.hidden)
        shapes = [w.shape for w, _ in base + head]
        sizes = [w.size + b.size for w, b in base + head]
        self.mlp_flat = torch.empty(sum(sizes), dtype=torch.float32, device=self.dev)
        self.mlp_grad = torch.zeros_like(self.mlp_flat)
        self.layers, self.layer_grads = [], []
        off = 0
        for (w, b), shp in zip(base + head, shapes):
            W = self.mlp_flat[off:off + w.size].view(shp); gW = self.mlp_grad[off:off + w.size].view(shp)
            off += w.size
            B = self.mlp_flat[off:off + b.size]; gB = self.mlp_grad[off:off + b.size]
            off += b.size
            W.copy_(t(w)); B.copy_(t(b))
            self.layers.append((W, B)); self.layer_grads.append((gW, gB))
        rng = np.random.default_rng(seed + 5)
        self.emb = t(rng.standard_normal((n_cams, 32)).astype(np.float32))  # nn.Embedding init N(0,1)
        self.emb_grad = torch.zeros_like(self.emb)
        self.table_grad = None
        self.record_counts_seen = None  # int32 like _records[1]: receives the stream counters each accumulate consumes (diagnostics, tests)
        self.table16 = None
        self.blocks16 = []
        self._plane_bufs = None
        self._x3_images = [None, None]   # split-f16 weight images of the fused decoder (forward, backward)
        self._x3_stale = True
        self.refresh_shadows()
        self._adam, self._adam_t = {}, {}
        self.step = 0
        self.weight_stats = torch.full((self.n_nodes,), 1000, dtype=torch.int64, device=self.dev)  # INIT_NODE_STAT
        self.alpha_stats = torch.full((self.n_nodes,), 1000, dtype=torch.int64, device=self.dev)
        self.visit_cnt = torch.zeros(self.n_nodes, dtype=torch.int64, device=self.dev)
        self.timed_ops = set()  # stage names to bracket with events on the launch stream (bench.py)
        self.timings = {}
        self._last_train_ctx = None
        # the plugin surface (gfnerf.optim.GfnAdam): optimisers that step engine-aliased parameters from the engine's own gradient
        # buffers, and the groups whose buffers hold this iteration's gradients ("table", "block<i>", "mlp", "emb")
        self.fused_optimizers = weakref.WeakSet()
        self.grad_ready = set()
        self._phase_state = None  # ops.PhaseState of the phased training forward (made for one batch size)
        self.phase_override = None  # False: one-phase forward whatever cfg.two_phase says (StepGraph: the plan said a split does not pay)

    def add_block(self, i, table=None, prims=None):
        """GFNeRFField.add_table (nerfacto_field.py:336-347): make residual block table i resident -- zeros and fresh primes
        (Hash3DAnchored.cpp:43-50) unless given (load_table)."""
        T = self.cfg.log2_T
        while len(self.blocks) <= i:
            self.blocks.append(None); self.block_prims.append(None)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)  # noqa: E731
        self.blocks[i] = torch.zeros((16 * (1 << T), 2), dtype=torch.float32, device=self.dev) if table is None else table
        if prims is None:
            prims = t(scene.make_primes(self.n_volumes, seed=self._seed + 100 + i))
        self.block_prims[i] = prims
        self.refresh_shadows()
        return self.blocks[i]

    def drop_block(self, i):
        """GFNeRFField.del_table (:349-365): release block i's table, its f16 shadow and its optimiser state."""
        self._sync_opt()
        self.forget_graphs()
        if i < len(self.blocks):
            self.blocks[i] = None; self.block_prims[i] = None
            if i < len(self.blocks16):
                self.blocks16[i] = None
        self.delete_optimizer(f"block{i}")

    def refresh_shadows(self):
        """(Re)build the f16 shadow tables after the masters were written from outside the optimiser."""
        if self.cfg.table_f16_shadow:
            self.table16 = ops.table_to_f16(self.table, self.table16)
            if len(self.blocks16) != len(self.blocks):
                self.blocks16 = (self.blocks16 + [None] * len(self.blocks))[: len(self.blocks)]
            self.blocks16 = [None if b is None else ops.table_to_f16(b, b16) for b, b16 in zip(self.blocks, self.blocks16)]
        self._x3_stale = True  # callers use this hook after writing parameters from outside the optimiser

    def _planes(self, cap):
        """Activation / gradient planes of the global stage (480 + 408 f16 columns; hi and lo' each, or one plane rounded to
        nearest: cfg.wgrad_planes), sized like the packed sample buffers and kept across steps."""
        if self._plane_bufs is None or self._plane_bufs[0][0].numel() < cap * ops.FIELD_X3_FWD_PLANE_LD:
            cap = (cap + 31) // 32 * 32  # whole 32-sample tiles (the planes are tile-major)
            # zero-initialised: rows of samples the fused forward skips (behind a ray's termination point) keep whatever finite
            # values they held, and meet zero gradients in gfn_wgrad_x3 -- garbage f16 could be NaN
            if self.cfg.wgrad_planes not in ("rne", "x3", "mixed"):
                raise _lib.GfnError(f"wgrad_planes must be 'rne', 'x3' or 'mixed', not {self.cfg.wgrad_planes!r}")
            mk = lambda ld, lo: (torch.zeros(cap * ld, dtype=torch.float16, device=self.dev),  # noqa: E731
                                 torch.zeros(cap * ld, dtype=torch.float16, device=self.dev) if lo else None)
            bwd_one = self.cfg.wgrad_planes != "x3"  # (its buffer then also holds dza1's lo' plane: ops.BWD_PLANE)
            self._plane_bufs = (mk(ops.FIELD_X3_FWD_PLANE_LD, self.cfg.wgrad_planes != "rne"),
                                mk(ops.FIELD_X3_BWD_PLANE_LD_RNE if bwd_one else ops.FIELD_X3_BWD_PLANE_LD, not bwd_one))
            # xp plane column -> hW0 column: base rows 1..15 = geo (16..30), row 0 = density channel (none), SH, appearance
            cm = np.concatenate([[-1], 15 + np.arange(1, 16), np.arange(16), 31 + np.arange(32)]).astype(np.int32)
            self._xp_col_map = torch.from_numpy(cm).to(self.dev)
            self._gmax = torch.zeros(1, dtype=torch.float32, device=self.dev)
        return self._plane_bufs

    def _plane_opts(self, planes):
        """Launch policy of a plane-producing forward: None (the default: dithered) unless cfg.plane_dither is off."""
        return ops.launch_opts(plain_rounding=True) if (planes is not None and not self.cfg.plane_dither) else None

    def _x3(self):
        """Current split-f16 weight images of the fused decoder (rebuilt after every weight update)."""
        if self._x3_stale or self._x3_images[0] is None:
            self._x3_images = list(ops.field_pack_x3(self.layers, self._x3_images[0], self._x3_images[1]))
            self._x3_stale = False
        return self._x3_images

    def _timed(self, name, fn, *a, **k):
        if name not in self.timed_ops and "*" not in self.timed_ops:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        self.timings.setdefault(name, []).append((e0, e1))
        return r

    def export_params(self):
        """Host copies of every parameter / state blob (for checkpoints and for the test oracle)."""
        self._sync_opt()
        c = lambda t: t.detach().cpu().numpy().copy()  # noqa: E731
        return dict(nodes=c(self.nodes), transes=c(self.transes), table=c(self.table), blocks=[None if b is None else c(b) for b in self.blocks],
                    block_prims=[None if p is None else c(p) for p in self.block_prims],
                    prim=c(self.prim), bias=c(self.bias), layers=[(c(W), c(B)) for W, B in self.layers], emb=c(self.emb))

    # ------------------------------------------------------------------ sampling
    def tree_depth(self):
        """Depth of the deepest node of the current blob (the traversal sizes its stacks by it); recomputed when the blob is
        replaced (ProcOctree, a loaded checkpoint) -- in-place updates (K6 / K7 invalidations) do not change it.  Synchronises
        when it recomputes: never under capture (StepGraph.capture and TileRenderer call it first)."""
        key = (self.nodes.data_ptr(), self.nodes.numel())
        if key != self._depth_key:
            self._depth_val = ops.tree_depth(self.nodes.view(-1, ops.TREE_NODE_BYTES))
            self._depth_key = key
        return self._depth_val

    def sample(self, rays_o, rays_d_raw, noise, need_world=False, out=None, record=False, pool_slots_per_ray=0, max_hits=None, own_workspace=False,
               opts=None):
        """PersSampler.GetSamples (PersSampler_cuda.cu:321-477), packed, table-space points.  out: the dict an earlier call
        returned, to be written again in place (the static sample buffers of a captured step, gfnerf_amd.stepgraph).
        record: traversal + counting march only (ops.sample_rays_record) -- the samples stay in the march's records, for
        render(tile_width=...), whose patch kernel reads them there (no scan, no fill, no packed copy); pool_slots_per_ray > 0: in pages
        from a pool of that many slots per ray on average instead of 1024 slots for every ray (ops.sample_rays_record); max_hits: leaves per
        ray of the record form's hit lists (default cfg.max_oct_per_ray = the reference's 1024: 12 KB per ray; a caller that passes less
        checks ops.HIT_CAP_REACHED in the overflow word); own_workspace: the hit lists belong to the returned dict (a ring of sample buffers);
        opts: ops.launch_opts(march_lanes=...) for the samplers' march (default: by batch size)."""
        c = self.cfg
        pre, self._pre = self._pre, None
        if not record and pre is not None and pre[0] is rays_o and pre[1] is rays_d_raw and pre[2] is noise and pre[3] == need_world:
            main = torch.cuda.current_stream()
            main.wait_stream(self._side)
            for v in list(pre[4].values()) + list(pre[4]["workspace"].values()):
                if isinstance(v, torch.Tensor):
                    v.record_stream(main)  # allocated on the side stream, consumed on this one
            return pre[4]
        elif pre is not None:
            torch.cuda.current_stream().wait_stream(self._side)  # a prefetch nobody collects still owns the workspace
        d = ops.normalize_dirs(rays_d_raw, out=None if out is None else out["dirs_unit"])
        R = rays_o.shape[0]
        # per-ray leftovers of the step that only depend on the rays: computed here so that a prefetched sampler takes them off the
        # main stream too (SH basis of the unit directions, nerfacto_field.py:519-521; the zero the loss kernels add to) -- and ahead
        # of the traversal, not behind the fill: the sampler's branch is what a replayed focal step waits for last
        sh = ops.sh4(d, out=None if out is None else out["sh"])
        if record:
            s = self._timed("sample", ops.sample_rays_record, rays_o, d, noise, self.nodes_compact, self.transes, c.sample_l, c.scale_by_dis,
                            near=c.global_near, far=1e8, max_hits=min(int(max_hits), c.max_oct_per_ray) if max_hits else c.max_oct_per_ray,
                            workspace=(out["workspace"] if out is not None else None) if own_workspace else
                            (self._record_ws if self._record_ws is not None else self._sampler_ws),  # (the hit lists of the packed form will do)
                            out=out, tree_depth=self.tree_depth(), pool_slots_per_ray=pool_slots_per_ray, opts=opts)
            if s["workspace"] is not self._sampler_ws and not own_workspace:
                self._record_ws = s["workspace"]
            s["dirs_unit"], s["sh"] = d, sh
            return s
        loss0 = torch.zeros(1, dtype=torch.float32, device=self.dev) if out is None else out["loss0_buf"].zero_()
        s = self._timed("sample", ops.sample_rays, rays_o, d, noise, self.nodes_compact, self.transes, c.sample_l,
                        c.scale_by_dis, near=c.global_near, far=1e8, max_hits=c.max_oct_per_ray,
                        pts_cap=max(R * c.max_samples_per_ray, 1), want_world=need_world, want_dirs=need_world,
                        table_space=True, workspace=self._sampler_ws, out=out, tree_depth=self.tree_depth(), opts=opts)
        self._sampler_ws = s["workspace"]
        s["dirs_unit"] = d
        s["n_dev"] = s["pts_offsets"][R:R + 1]
        s["sh"], s["loss0"] = sh, loss0
        s["loss0_buf"] = s["loss0"]  # forward_backward pops "loss0"; the buffer stays for the next in-place refill
        return s

    def _new_side_stream(self):
        """The second stream (next batch's draw + sampler, octree update): one that demonstrably runs BESIDE the current stream (ops.concurrent_stream)."""
        return ops.concurrent_stream(self.dev, role="sampler", priority=self.cfg.sampler_stream_priority)

    def side_stream(self):
        if self._side is None:
            self._side = self._new_side_stream()
        return self._side

    def defer(self, fn):
        """Host work the caller wants done once this iteration's backward and optimiser launches are enqueued (the plugin surface draws
        and samples the NEXT batch there: between the forward and the backward the host is what the GPU waits for).  Run by the fused
        optimiser after its launches (gfnerf/optim.py), or by whoever calls run_deferred() first."""
        self._deferred.append(fn)

    def run_deferred(self):
        todo, self._deferred = self._deferred, []
        for fn in todo:
            fn()

    def prefetch_sample(self, rays_o, rays_d_raw, noise, need_world=False, first=None, out=None, after=None):
        """Run the sampler of the NEXT batch on a second stream now; the next self.sample(...) with the very same tensors
        collects it.  The ray march is latency-bound and the fused backward beside it fabric-bound, so the two share the
        chip (the backward is capped to cfg.prefetch_cus CUs meanwhile).  Only valid while the octree does not change
        between the two steps, i.e. in the focal stage (UpdateOctNodes is global-stage only, nerfacto.py:598-616) -- or, in the
        global stage, when `first` is this step's octree update: the reference runs UpdateOctNodes inside get_outputs, right
        after the forward and before the backward (nerfacto.py:598-616), so update -> next sampler on the second stream is its
        order."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = self._new_side_stream()
        if after is not None:  # an event of the current stream: what the side branch has to wait for (not everything enqueued since)
            self._side.wait_event(after)
        else:
            self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            if first is not None:
                first()
            s = self.sample(rays_o, rays_d_raw, noise, need_world, out=out)
        self._pre = (rays_o, rays_d_raw, noise, need_world, s)

    # -------------------------------------------------------------------- field
    def _field_forward(self, s, cam_idx, block=None, fused=None, save=False, planes=None, lm=False):
        """GFNeRFField.get_density + get_outputs (nerfacto_field.py:412-591) on packed samples.
        fused: one decoder kernel (sigma, rgb [+ ReLU masks when save]); else the per-layer kernels,
        which keep every activation (needed for the weight gradients of the global stage)."""
        n_dev = s["n_dev"]
        T, V = self.cfg.log2_T, self.n_volumes
        fused = (self.cfg.fused_field and self.cfg.hidden == 128) if fused is None else fused
        lm = lm and fused and self.cfg.mlp_precision == "x3"  # the x3 decoder reads level-major features
        if not (lm and block is not None):
            self._sync_opt()
        if self.cfg.table_f16_shadow and lm and block is not None and self._opt_pending:
            # the frozen global table first, beside the previous step's Adam on the block table; then wait for that Adam
            feat = self._timed("hash_fwd", ops.hash_fwd_lm, self.table16, self.prim, self.bias_arg, s["warped_pts"], s["anchors"], V, T,
                               n_dev=n_dev, halves_from=self.cfg.lm_halves_from, split_log2=self.cfg.lm_split_log2, planar=True)
            self._sync_opt()
            self._timed("hash_fwd", ops.hash_fwd_lm, self.blocks16[block], self.block_prims[block], self.bias_arg, s["warped_pts"], s["anchors"],
                        V, T, n_dev=n_dev, halves_from=self.cfg.lm_halves_from, split_log2=self.cfg.lm_split_log2, planar=True, out=feat, accumulate=True)
        elif self.cfg.table_f16_shadow and lm:
            feat = self._timed("hash_fwd", ops.hash_fwd_lm, self.table16, self.prim, self.bias_arg, s["warped_pts"], s["anchors"], V, T,
                               table2_f16=self.blocks16[block] if block is not None else None,
                               prim_pool2=self.block_prims[block] if block is not None else None, n_dev=n_dev,
                               halves_from=self.cfg.lm_halves_from, split_log2=self.cfg.lm_split_log2, planar=True)
        elif self.cfg.table_f16_shadow and block is not None:  # residual block table (nerfacto_field.py:485-486) fused in
            feat = self._timed("hash_fwd", ops.hash_fwd_f16, self.table16, self.prim, self.bias_arg, s["warped_pts"],
                               s["anchors"], V, T, table2_f16=self.blocks16[block], prim_pool2=self.block_prims[block], n_dev=n_dev)
        elif self.cfg.table_f16_shadow:
            feat = self._timed("hash_fwd", ops.hash_fwd_f16, self.table16, self.prim, self.bias_arg, s["warped_pts"],
                               s["anchors"], V, T, n_dev=n_dev)
        else:
            feat = self._timed("hash_fwd", ops.hash_fwd, self.table, self.prim, self.bias_arg, s["warped_pts"], s["anchors"],
                               V, T, n_dev=n_dev)
            if block is not None:
                self._timed("hash_fwd_block", ops.hash_fwd, self.blocks[block], self.block_prims[block], self.bias_arg,
                            s["warped_pts"], s["anchors"], V, T, n_dev=n_dev, out=feat, accumulate=True)
        if self.after_lookup is not None:
            self.after_lookup()
        emb = self.emb if self.cfg.use_appearance_embedding else None
        if fused:
            sh = s["sh"] if s.get("sh") is not None else ops.sh4(s["dirs_unit"])
            prec = self.cfg.mlp_precision
            out = self._timed("field_fwd", ops.field_fwd, feat, s["ray_id"], sh, emb, cam_idx, self.layers, n_dev=n_dev, save=save,
                              precision=prec, image=self._x3()[0] if prec == "x3" else None, planes=planes, opts=self._plane_opts(planes))
            out["feat"] = feat
            return out
        (bW0, bb0), (bW1, bb1), (hW0, hb0), (hW1, hb1), (hW2, hb2) = self.layers
        a1 = self._timed("base_fwd", ops.linear_fwd, feat, bW0, bb0, "relu", n_dev=n_dev)
        h = self._timed("base_fwd", ops.linear_fwd, a1, bW1, bb1, "none", n_dev=n_dev)
        sh = s["sh"] if s.get("sh") is not None else ops.sh4(s["dirs_unit"])
        sigma, xh = self._timed("head_pack", ops.head_pack, h, sh, emb, cam_idx, s["ray_id"], n_dev=n_dev)
        c1 = self._timed("head_fwd", ops.linear_fwd, xh, hW0, hb0, "relu", n_dev=n_dev, in_dim=63)
        c2 = self._timed("head_fwd", ops.linear_fwd, c1, hW1, hb1, "relu", n_dev=n_dev)
        rgb = self._timed("head_fwd", ops.linear_fwd, c2, hW2, hb2, "sigmoid", n_dev=n_dev)
        return dict(feat=feat, a1=a1, h=h, xh=xh, c1=c1, c2=c2, rgb=rgb, sigma=sigma)

    def phase_state(self, n_rays, cap):
        if self._phase_state is None or self._phase_state.n_rays != n_rays or self._phase_state.cap < cap:
            self._phase_state = ops.PhaseState(n_rays, cap, self.dev)
        return self._phase_state

    @property
    def _phase(self):
        """The phase state as the last forward's plan left it (its branch joined into the current stream first)."""
        self._join_aux()
        return self._phase_state

    def _forward_phased(self, s, cam_idx, block, save, planes, want_depth):
        """The termination-aware level-major training forward (include/gfnerf_amd.h, gfn_composite_fwd_phased): phase A = the first K
        samples of every ray through lookup(s) + decoder + compositing; phase B = the rest of the rays that are still transparent
        behind them (its launches return at once when there is none); then the next step's K.  -> (field dict, comp dict)."""
        c = self.cfg
        self._join_aux()  # (a forward whose backward never ran: its plan must have written K before this one reads it)
        n_dev = s["n_dev"]
        T, V = c.log2_T, self.n_volumes
        R = s["pts_offsets"].numel() - 1
        st = self.phase_state(R, s["ts"].shape[0])
        emb = self.emb if c.use_appearance_embedding else None
        sh = s["sh"] if s.get("sh") is not None else ops.sh4(s["dirs_unit"])
        lm_kw = dict(n_dev=n_dev, halves_from=c.lm_halves_from, split_log2=c.lm_split_log2, planar=True)
        tables = [(self.table16, self.prim)] + ([(self.blocks16[block], self.block_prims[block])] if block is not None else [])

        def lookups(name, samples, feat, first):
            for i, (tab, prim) in enumerate(tables):
                if i == 1 and first:
                    self._sync_opt()  # the block table's lookup waits for the previous step's (asynchronous) Adam; the frozen global table's does not
                feat = self._timed(name, ops.hash_fwd_lm, tab, prim, self.bias_arg, s["warped_pts"], s["anchors"], V, T, out=feat,
                                   accumulate=i > 0, sample_list=samples, **lm_kw)
            return feat

        if block is None:
            self._sync_opt()
        f = ops.field_outputs(s["ts"].shape[0], self.dev, save=save)
        sa = self._timed("phase_list", ops.phase_list, s["pts_offsets"], st, zero=(f["sigma"], f["rgb"]))
        feat = lookups("hash_fwd", sa, None, True)
        if self.after_lookup is not None:
            self.after_lookup()
        self._timed("field_fwd", ops.field_fwd, feat, s["ray_id"], sh, emb, cam_idx, self.layers, n_dev=n_dev, save=save, precision="x3",
                    image=self._x3()[0], planes=planes, sample_list=sa, out=f, opts=self._plane_opts(planes))
        comp = self._timed("composite_fwd", ops.composite_fwd_phased, s["pts_offsets"], f["sigma"], s["dists"], s["ts"], f["rgb"], st, c.early_T,
                           want_t_range=want_depth)
        sb = self._timed("phase_list_b", ops.phase_list, s["pts_offsets"], st, phase_b=True)
        lookups("hash_fwd_b", sb, feat, False)
        self._timed("field_fwd_b", ops.field_fwd, feat, s["ray_id"], sh, emb, cam_idx, self.layers, n_dev=n_dev, save=save, precision="x3",
                    image=self._x3()[0], planes=planes, sample_list=sb, out=f, opts=self._plane_opts(planes))
        self._timed("composite_fwd_b", ops.composite_fwd_phased, s["pts_offsets"], f["sigma"], s["dists"], s["ts"], f["rgb"], st, c.early_T,
                    second=True, out=comp)
        # the backward's list and the next step's K are not the loss's business: a branch of their own, beside the loss kernels
        # (two dependent small launches less between the compositing and the backward); joined by backward() / the next forward
        main = torch.cuda.current_stream()
        if self._aux_stream is None:
            self._aux_stream = ops.concurrent_stream(self.dev, role="aux")
        self._aux_stream.wait_stream(main)
        with torch.cuda.stream(self._aux_stream):
            if planes is None:  # (focal stage) the backward walks exactly the samples in front of the termination points
                f["front"] = self._timed("phase_list_front", ops.phase_list, s["pts_offsets"], st, front=True)
            enter, stay = c.two_phase_min_saving_planes if planes is not None else c.two_phase_min_saving
            self._timed("phase_plan", ops.two_phase_plan, s["pts_offsets"], st, c.two_phase_fixed_cost, save_enter=enter, save_stay=stay)
            if self.k_host is not None:  # a pinned int32 [1]: the planned K travels to the host from this branch (read late, never waited for)
                self.k_host.copy_(st.k, non_blocking=True)
            if planes is None:  # and how long the backward's list is (_cu_cap reads it a step or two late: a hint, not a value)
                self._front_host[0:1].copy_(st.stats[4:5], non_blocking=True)
                self._front_host[1:2].copy_(n_dev, non_blocking=True)
        self._aux_pending = True
        f["feat"] = feat
        return f, comp

    def _join_aux(self):
        """The branch that builds the backward's sample list and plans the next K (_forward_phased) -> this stream."""
        if self._aux_pending:
            torch.cuda.current_stream().wait_stream(self._aux_stream)
            self._aux_pending = False

    def _fused_ok(self, block):
        c = self.cfg
        return c.fused_forward and c.fused_field and c.hidden == 128 and c.mlp_precision == "x3" and c.table_f16_shadow

    def _forward_fused(self, s, cam_idx, block, early_T, save=False, planes=None, want_weights=True):
        """Hash lookup + decoder + compositing in one launch (csrc/fused_fwd.hip) -> (field dict, comp dict)."""
        self._sync_opt()
        if self.after_lookup is not None:
            self.after_lookup()
        sh = s["sh"] if s.get("sh") is not None else ops.sh4(s["dirs_unit"])
        emb = self.emb if self.cfg.use_appearance_embedding else None
        out = self._timed("fused_fwd", ops.fused_forward, s, self.table16, self.prim, self.n_volumes,
                          self.cfg.log2_T, self._x3()[0], sh, emb, cam_idx,
                          table2_f16=self.blocks16[block] if block is not None else None,
                          prim_pool2=self.block_prims[block] if block is not None else None, bias_pool=self.bias_arg,
                          early_T=early_T, save=save, want_weights=want_weights, want_feat=self.cfg.keep_features, planes=planes,
                          opts=self._plane_opts(planes))
        f = dict(sigma=out["sigma"], rgb=out["rgb_samples"], h0=out["h0"], masks=out["masks"], feat=out["feat"])
        comp = dict(weights=out["weights"], alphas=out["alphas"], rgb=out["rgb"], acc=out["acc"], depth_raw=out["depth_raw"],
                    t_minmax=out["t_minmax"])
        return f, comp

    def uses_patch_kernel(self, n_rays, tile_width, block=None):
        """Will render(tile_width=...) of n_rays rays go through the patch kernel (gfn_fused_render: lookup + decoder + compositing,
        32 neighbouring rays in lockstep)?  Then -- and only then -- the samples may be in the record form (sample(record=True)): callers
        that sample ahead of render (the tile pipeline) decide the form with this one predicate."""
        pw = self.cfg.render_patch_width
        return bool(tile_width is not None and pw > 0 and self._fused_ok(block) and n_rays % 32 == 0 and
                    (tile_width == 0 or (tile_width % pw == 0 and n_rays % tile_width == 0 and (n_rays // tile_width) % (32 // pw) == 0)))

    def render(self, rays_o, rays_d_raw, cam_idx, noise, block=None, early_T=None, samples=None, tile_width=None, opts=None, cam_uniform=False):
        """Inference forward of get_outputs: {rgb, accumulation, depth, oct_depth} (nerfacto.py:545-566).
        samples: the result of self.sample for these rays when the caller ran it already (pipelined tiles).
        tile_width (not None): the rays are neighbours -- a block of pixels this wide, row major, or (0) a run of a row -- and only
        the four images are wanted: the lookup + decoder + compositing go through gfn_fused_render, which walks patches of 32
        neighbouring rays in lockstep (field / weights are then not returned).  opts: ops.launch_opts(decoder_cus=...) of the patch kernel's
        launch (TileRenderer leaves CUs to the next tile's sampler).  cam_uniform: every entry of cam_idx is the same camera (the rays of one image):
        the patch kernel folds that camera's appearance embedding into the head's bias (ops.fused_render)."""
        c = self.cfg
        early = c.early_T if early_T is None else early_T
        R = rays_o.shape[0]
        pw = c.render_patch_width
        patches = self.uses_patch_kernel(R, tile_width, block)
        s = self.sample(rays_o, rays_d_raw, noise, record=patches) if samples is None else samples  # (the patch kernel reads the march's records)
        recorded = s.get("pts_offsets") is None  # sample(record=True): only the patch kernel reads that form
        if (s["counts"].numel() if recorded else s["pts_offsets"].numel() - 1) != R:
            raise _lib.GfnError("render: `samples` were made for another ray count")
        if patches:
            self._sync_opt()
            sh = s["sh"] if s.get("sh") is not None else ops.sh4(s["dirs_unit"])
            comp = self._timed("fused_render", ops.fused_render, s, self.table16, self.prim, self.n_volumes, c.log2_T,
                               self._x3()[0], sh, self.emb if c.use_appearance_embedding else None, cam_idx,
                               table2_f16=self.blocks16[block] if block is not None else None,
                               prim_pool2=self.block_prims[block] if block is not None else None, bias_pool=self.bias_arg,
                               early_T=early, tile_width=tile_width, patch_width=pw, opts=opts, cam_uniform=cam_uniform)
            depth = ops.depth_finalize(comp["depth_raw"], comp["t_minmax"], c.scale_factor)
            rgb = torch.clamp(torch.nan_to_num(comp["rgb"]), 0.0, 1.0)
            return dict(rgb=rgb, accumulation=comp["acc"][:, None], depth=depth[:, None],
                        oct_depth=s["first_oct_dis"][:, None] / c.scale_factor, samples=s, field=None, weights=None)
        if recorded:
            raise _lib.GfnError("render: samples of sample(record=True) need tile_width and a ray count the patch kernel takes (whole patches of 32)")
        if self._fused_ok(block):
            f, comp = self._forward_fused(s, cam_idx, block, early)
        else:
            f = self._field_forward(s, cam_idx, block)
            comp = ops.composite_fwd(s["pts_offsets"], f["sigma"], s["dists"], s["ts"], f["rgb"], early_T=early)
        depth = ops.depth_finalize(comp["depth_raw"], comp["t_minmax"], c.scale_factor)
        rgb = torch.clamp(torch.nan_to_num(comp["rgb"]), 0.0, 1.0)  # eval branch of RGBRenderer (renderers.py:131-137)
        return dict(rgb=rgb, accumulation=comp["acc"][:, None], depth=depth[:, None],
                    oct_depth=s["first_oct_dis"][:, None] / c.scale_factor, samples=s, field=f, weights=comp["weights"],
                    alphas=comp["alphas"])

    # ----------------------------------------------------------------- training
    def _adam_state(self, name, p):
        if name not in self._adam:
            self._adam[name] = (torch.zeros_like(p), torch.zeros_like(p))
            self._adam_t[name] = 0
        return self._adam[name]

    def _adam_tick(self, name):
        """Every parameter group is its own torch.optim.Adam in the reference (Optimizers, engine/optimizers.py:100-133): its
        bias correction counts that group's steps, not the trainer's."""
        self._adam_t[name] += 1
        return self._adam_t[name]

    def delete_optimizer(self, name):
        """Optimizers.delete_optimizer (engine/optimizers.py:161-168; called by the update_optimizer callback,
        gfnerf/nerfacto.py:448-488, for the groups that leave the active set): the group's Adam moments and step count go; a
        later step of the group starts a fresh optimizer, like add_optimizer does.  Groups: "table", "mlp", "emb", "block<i>"."""
        self._adam.pop(name, None)
        self._adam_t.pop(name, None)

    def update_optimizer(self, stage, active_blocks=()):
        """The update_optimizer callback (gfnerf/nerfacto.py:448-488): init stage keeps the global encoding + fields and drops
        every block's optimizer; block stage drops those and keeps only the active blocks' (added lazily at lr_block = 5e-3)."""
        if stage == STAGE_GLOBAL:
            for name in [k for k in self._adam if k.startswith("block")]:
                self.delete_optimizer(name)
        else:
            keep = {f"block{b}" for b in active_blocks}
            for name in [k for k in self._adam if k not in keep]:
                self.delete_optimizer(name)

    def _sync_opt(self):
        """Everything after this point sees the last (asynchronous) optimiser step and its zeroed gradient pool."""
        if self._opt_pending:
            torch.cuda.current_stream().wait_stream(self._opt_stream)
            self._opt_pending = False

    def _reduce_records(self, rec):
        """gfn_grad_records_reduce over every slot range: the record streams of the last backward -> the f16 x128 gradient pool."""
        c = self.cfg
        self._timed("grad_reduce", ops.grad_records_reduce, rec, self.table_grad, c.log2_T, counts_seen=self.record_counts_seen,
                    f16_accumulate=not c.lds_accumulate_exact)

    def _adam_table(self, name, p, m, v, lr, shadow, hyper):
        """Adam on the live elements of a table (elements [0, 17T): the rest can never receive a gradient)."""
        c = self.cfg
        self._timed("adam", ops.adam_step_table, p, self.table_grad, m, v, self._adam_tick(name), lr=lr, eps=c.adam_eps, shadow_f16=shadow,
                    n=17 * (1 << c.log2_T), hyper=hyper)

    def _grad_live(self):
        """Elements [0, 17T) of the gradient pool: the only ones a backward can write (level-overlap quirk), so the only ones to
        zero, reduce across ranks and hand to Adam."""
        return self.table_grad.view(-1)[: 17 * (1 << self.cfg.log2_T)]

    def zero_grads(self, stage, block=None):
        if self.table_grad is None:
            self.table_grad = torch.zeros(self.table.shape, dtype=torch.float16 if self.cfg.grad_f16 else torch.float32,
                                          device=self.dev)
        if not self._opt_pending:  # else: zeroed behind the asynchronous Adam
            ops.fill_zero(self._grad_live())
        if stage == STAGE_GLOBAL:
            ops.fill_zero(self.mlp_grad)
            ops.fill_zero(self.emb_grad)

    def forward_train(self, rays_o, rays_d_raw, cam_idx, noise, stage=STAGE_GLOBAL, block=None, want_depth=True, samples=None, all_alphas=False):
        """Training forward of get_outputs.  Returns (comp, ctx): comp holds rgb [R,3], acc, depth_raw, weights,
        alphas; ctx is what backward() needs (the reference keeps the whole autograd graph alive instead).
        want_depth=False (the native train step, whose losses do not read depth) skips the batch-global t range that
        DepthRenderer's clip needs (renderers.py:281): comp["t_minmax"] is then None.
        all_alphas: comp["alphas"] must be the TRUE alpha of every sample, also behind a ray's termination point -- what UpdateOctNodes
        reads (K5 compares each node's largest alpha with its thresholds, PersSampler_cuda.cu:21-75: a dense node that the rays of one
        batch only reach behind their cuts must still count as occupied, or its alpha statistic decays 16x faster than the
        reference's weight statistic lets it).  The density of every sample is then evaluated: no ray leaves the fused forward
        kernel early, no phased forward; the cut at early_T still applies to the colour (< 1e-4), the weights (0 instead of < 1e-4,
        below every threshold K5 can form for a ray that has a cut) and the backward."""
        c = self.cfg
        train_mlp = stage == STAGE_GLOBAL
        self.last_forward_replayed = False
        self._timed("zero_grads", self.zero_grads, stage, block)
        s = self.sample(rays_o, rays_d_raw, noise) if samples is None else samples  # samples: this batch's, made already
        # focal stage (frozen decoder): the ReLU masks are all the backward needs; global stage: the x3 kernels also keep
        # every layer input as split-f16 planes for the weight gradients (the exact-f32 fused kernels have no such form)
        fused = c.fused_field and c.hidden == 128 and (not train_mlp or c.mlp_precision == "x3")
        planes = self._planes(s["ts"].shape[0])[0] if (fused and train_mlp) else None
        blk = block if stage == STAGE_FOCAL else None
        # level-major lookup (own primes per table): three kernels beat the fused one because the gathers then hit L2
        # (with one table the fused kernel's early termination makes it a tie at first and a loss once the scene has density:
        # cfg.level_major_global / level_major_global_after)
        all_alphas = bool(all_alphas) and c.early_T > 0 and c.exact_oct_alphas
        lm_global = self.level_major_global_now(all_alphas)
        lm = (c.level_major and (blk is not None or lm_global) and fused and c.mlp_precision == "x3" and c.table_f16_shadow
              and not c.keep_features)
        phased = lm and c.two_phase and c.early_T > 0 and self.phase_override is not False and not all_alphas
        use_fused_kernel = fused and self._fused_ok(blk) and not lm and not all_alphas
        self.last_forward = ("lm2" if phased else "lm") if lm else ("fused" if use_fused_kernel else "unfused")
        if phased:
            f, comp = self._forward_phased(s, cam_idx, blk, True, planes, want_depth)
        elif use_fused_kernel:
            f, comp = self._forward_fused(s, cam_idx, blk, c.early_T, save=True, planes=planes)
        else:
            f = self._field_forward(s, cam_idx, blk, fused=fused, save=fused, planes=planes, lm=lm)
            comp = self._timed("composite_fwd", ops.composite_fwd, s["pts_offsets"], f["sigma"], s["dists"], s["ts"], f["rgb"],
                               early_T=c.early_T, want_t_range=want_depth)
        ctx = dict(samples=s, field=f, comp=comp, cam_idx=cam_idx, stage=stage, fused=fused, block=block)
        self._last_train_ctx = ctx
        return comp, ctx

    def level_major_global_now(self, all_alphas=False):
        """Global stage: would forward_train take the level-major forward at THIS optimiser step?  (cfg.level_major_global, or from step
        cfg.level_major_global_after on, or whenever UpdateOctNodes needs every sample's alpha.)  A captured step has the answer of its
        capture baked in: StepGraph compares and captures again when it changes."""
        c = self.cfg
        return bool(c.level_major_global or (c.level_major_global_after >= 0 and self.step >= c.level_major_global_after) or
                    (all_alphas and c.early_T > 0 and c.exact_oct_alphas))

    def forget_graphs(self):
        """Whatever a captured forward has baked in (tree blob, table addresses) is about to change."""
        self._fwd_graphs.clear()

    def forward_train_graphed(self, cam_idx, stage, block, samples, want_depth=True, cam_static=False):
        """forward_train(samples=...) for a caller whose sample buffers are static (HotPath.sample(out=...): the same dict every other
        step) -- after three eager runs per (stage, block, buffer, forward kind) the launches are captured once and replayed: the
        training iteration of the plugin surface is bound by the host, and the forward is a third of its launches.  Outputs (comp,
        ctx) of a replay are the capture's tensors, written again: valid until the NEXT replay of any captured forward (the captures share one
        memory pool, so a later forward's transient buffers may reuse an earlier one's; a training iteration consumes them in its own backward,
        before the next forward).  At most cfg.graph_forward_keep captures are kept, the least recently used go first (each holds a full set
        of forward activations).  A capture with want_depth also holds the two per-ray leftovers of get_outputs -- comp["depth"] (DepthRenderer's clip + the
        scale, nerfacto.py:545-546) and comp["oct_depth"] (:555) -- so that a replayed iteration launches nothing for them.  cam_static: cam_idx is a
        buffer of the caller's that keeps its address (written before every call: the static sample set's own camera column): read in place, not copied."""
        c = self.cfg
        lm_global = c.level_major_global or (c.level_major_global_after >= 0 and self.step >= c.level_major_global_after)
        blk = block if stage == STAGE_FOCAL else None
        usable = (c.graph_forward and self.seg is None and self.after_lookup is None and not self.timed_ops
                  and not torch.cuda.is_current_stream_capturing())
        # the forward's form: phased (the planner splits, or has not spoken yet) or one phase (its K says a split does not pay: cfg.graph_one_phase)
        can_one = usable and c.graph_one_phase and stage == STAGE_FOCAL and c.two_phase and c.early_T > 0 and self.phase_override is None
        if can_one and self.k_host is None:
            self.k_host = torch.zeros(1, dtype=torch.int32).pin_memory()  # the planned K, copied by every phased forward's side branch (read late)
        if can_one and self._one_phase:
            self._one_phase_uses += 1
            if self._one_phase_uses % max(c.graph_one_phase_probe, 1) == 0:  # look at the plan again: this forward eager and phased (same results)
                self._one_phase_probing = 2
                return self.forward_train(None, None, cam_idx, None, stage, block, want_depth=want_depth, samples=samples)
            if self._one_phase_probing:  # (the probe's K has had an iteration or two to reach the pinned word)
                self._one_phase_probing -= 1
                if self._one_phase_probing == 0 and int(self.k_host[0]) < 1024:
                    self._one_phase = False
        one = bool(can_one and self._one_phase)
        key = (id(samples), stage, block, bool(want_depth), self.nodes.data_ptr(), self.n_nodes, lm_global, False if one else self.phase_override, c.early_T, c.two_phase,
               self.table16.data_ptr() if self.table16 is not None else 0,
               0 if blk is None else (self.blocks16[blk].data_ptr() if c.table_f16_shadow else self.blocks[blk].data_ptr(), self.block_prims[blk].data_ptr()))
        ent = self._fwd_graphs.setdefault(key, dict(n=0, samples=samples))  # (holds the buffers: their id stays theirs)
        self._fwd_graphs.move_to_end(key)

        def forward(cam):
            prev = self.phase_override
            if one:
                self.phase_override = False
            try:
                return self.forward_train(None, None, cam, None, stage, block, want_depth=want_depth, samples=samples)
            finally:
                self.phase_override = prev

        if not usable or ("graph" not in ent and (ent["n"] < (1 if one else 3) or self.table_grad is None)):
            ent["n"] += 1
            return forward(cam_idx)
        if "graph" not in ent and can_one and not one and self._phase_state is not None:
            # the eager phased runs have planned K: a plan of one phase (K = 1024; read exactly, the capture synchronises anyway) is captured as such --
            # from the next call on (the one-phase form runs eagerly once on these buffers first)
            self._join_aux()
            if int(self._phase_state.k.item()) >= 1024:
                self._one_phase, self._one_phase_uses, self._one_phase_probing = True, 0, 0
                ent["n"] += 1
                return forward(cam_idx)
        if "graph" not in ent:
            cam = cam_idx if cam_static else cam_idx.clone()
            self._join_aux()
            self._sync_opt()
            if stage == STAGE_GLOBAL:
                self._x3_stale = True  # the decoder changes every step in this stage: the repack of its weight images belongs to every replay
            else:
                self._x3()             # frozen decoder: packed now, not under capture
            torch.cuda.synchronize()
            # a new key (another block became active, the plan changed the forward kind, ...): the graphs of keys not used for longest
            # go, with the activations they hold, and so do the counters of keys that never reached a capture
            captured = [k for k, e in self._fwd_graphs.items() if "graph" in e]
            for k in captured[: max(len(captured) - (c.graph_forward_keep - 1), 0)]:
                del self._fwd_graphs[k]
            for k in [k for k, e in self._fwd_graphs.items() if "graph" not in e and k != key][:-8]:
                del self._fwd_graphs[k]
            if self._fwd_pool is None and c.graph_forward_shared_pool:
                self._fwd_pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._fwd_pool if c.graph_forward_shared_pool else None, capture_error_mode="thread_local"):
                comp, ctx = forward(cam)
                self._join_aux()  # (the branch that builds the backward's list: inside the graph)
                if want_depth:
                    comp["depth"] = ops.depth_finalize(comp["depth_raw"], comp["t_minmax"], c.scale_factor)[:, None]
                    comp["oct_depth"] = samples["first_oct_dis"][:, None] / c.scale_factor
            ent.update(graph=g, cam=cam, comp=comp, ctx=ctx, last_forward=self.last_forward)
            ctx["graph_key"] = key  # (backward_graphed finds the capture its context belongs to)
        if ent["cam"].data_ptr() != cam_idx.data_ptr():
            ent["cam"].copy_(cam_idx)
        self._join_aux()
        self._sync_opt()
        if stage != STAGE_GLOBAL and self._x3_stale:
            self._x3()  # the decoder was written from outside (a checkpoint): repacked in place, the replay reads the same images
        ent["graph"].replay()
        self.last_forward_replayed = True
        self.last_forward, self._last_train_ctx = ent["last_forward"], ent["ctx"]
        if stage == STAGE_GLOBAL:
            self._x3_stale = False
        return ent["comp"], ent["ctx"]

    def grad_in_buffer(self, ctx):
        """Where the backward of the captured forward `ctx` belongs to wants d loss / d rgb [R,3] written (a static buffer its replayed backward reads in
        place), or None: a caller that produces the gradient there (gfnerf/nerfacto.py's fused loss node) saves backward_graphed a copy."""
        ent = self._entry_of(ctx)
        if ent is None or not self.cfg.graph_backward:
            return None
        if "g_rgb" not in ent:
            ent["g_rgb"] = torch.zeros_like(ent["comp"]["rgb"])
        return ent["g_rgb"]

    def _entry_of(self, ctx):
        ent = self._fwd_graphs.get(ctx.get("graph_key")) if ctx is not None else None
        return ent if (ent is not None and ent.get("ctx") is ctx) else None

    def backward_graphed(self, ctx, g_rgb, g_acc=None):
        """backward(ctx, g_rgb) for the context of a REPLAYED forward (forward_train_graphed): the same launches -- compositing backward, decoder backward +
        record scatter (the gated pair), LDS accumulation -- captured once per forward capture on its static buffers and replayed from then on; d loss /
        d rgb is read from the capture's own buffer (grad_in_buffer: written there by the caller, or copied).  Everything else (another stage, d acc, a
        context that is not a capture's, timed ops, a segmented or enclosing capture): the eager backward."""
        c = self.cfg
        self.last_backward_replayed = False
        ent = self._entry_of(ctx)
        usable = (ent is not None and c.graph_backward and c.graph_forward and g_acc is None and ctx["stage"] == STAGE_FOCAL and self.seg is None
                  and not self.timed_ops and self.record_counts_seen is None and not torch.cuda.is_current_stream_capturing())
        if not usable:
            return self.backward(ctx, g_rgb, g_acc)
        buf = self.grad_in_buffer(ctx)
        if g_rgb.data_ptr() != buf.data_ptr():
            buf.copy_(g_rgb)
        if "bgraph" not in ent:
            ent["bn"] = ent.get("bn", 0) + 1
            if ent["bn"] < 2:  # (one eager run on these very buffers first: lazily made state -- record buffers, kernel attributes, the merge gate's words)
                return self.backward(ctx, buf)
            self._sync_opt()
            self._join_aux()
            self._x3()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            self._backward_eager_like = True  # (the sampler beside this graph is launched eagerly on the second stream: the eager CU share, not a captured step's)
            try:
                with torch.cuda.graph(g, pool=self._fwd_pool if c.graph_forward_shared_pool else None, capture_error_mode="thread_local"):
                    self.backward(ctx, buf)
            finally:
                self._backward_eager_like = False
            ent["bgraph"], ent["b_cus"] = g, self._cu_cap()
        self._sync_opt()
        self._join_aux()
        ent["bgraph"].replay()
        self.last_backward_replayed = True

    def _record_buffers(self):
        c = self.cfg
        if not (c.lds_accumulate and c.grad_f16 and c.log2_T <= 21):
            return None
        if self._records is None:
            n_streams = _lib.lib().gfn_grad_record_ranges(c.log2_T) * 256
            self._records = ops.grad_records_buffers(c.log2_T, min(c.record_capacity, n_streams * 2048), self.dev)  # <= 2048 records per stream
        return self._records

    def _dense_hint(self):
        """Is the focal batch at hand DENSE (cfg.prefetch_dense_frac of its samples in front of the termination points)?  The host's view: the counts of
        StepGraph's planning forward while it captures, else those of the last phased forward (a pinned pair, read late: a hint)."""
        front, n_all = self.front_override if self.front_override is not None else (int(self._front_host[0]), int(self._front_host[1]))
        return n_all > 0 and front >= self.cfg.prefetch_dense_frac * n_all

    def _cu_cap(self, stage=STAGE_FOCAL):
        """CUs of the persistent backward kernel (0: all): while a prefetched sampler runs on the second stream it leaves that one some of the CUs --
        fewer of them when the focal batch is dense (its backward is long: _dense_hint).  Only the grid depends on this: the gradient's bits do not."""
        c = self.cfg
        if self._pre is None:
            return 0
        n = c.prefetch_cus_global if stage == STAGE_GLOBAL else c.prefetch_cus
        if stage != STAGE_GLOBAL and self._dense_hint():
            n = c.prefetch_cus_dense if torch.cuda.is_current_stream_capturing() else c.prefetch_cus_dense_eager
            if self._backward_eager_like:
                n = c.prefetch_cus_dense_replay
        return n if 0 < n < 256 else 0

    def _merge_gate(self, front, n_dev):
        """cfg.record_merge_dense: the focal backward merges the record runs of a DENSE batch (cfg.prefetch_dense_frac).  front = (list, count) of the phased
        forward, or None (a one-phase forward: every sample is in front): the DEVICE decides from this step's own counts (ops.launch_opts merge_gate: both
        forms are launched, one does the work), so a replayed forward and an eager one lead to the same bits.  -> the gate, or None: no merging."""
        if not self.cfg.record_merge_dense:
            return None
        return (front[1] if front is not None else n_dev, n_dev, float(self.cfg.prefetch_dense_frac))

    def backward(self, ctx, g_rgb, g_acc=None):
        """Gradients of every trainable parameter of the stage from d loss / d rgb [R,3] (and optionally d acc [R]): final on the
        current stream when this returns (data parallel: the table gradient's exchange may still be in flight, see sync_grads)."""
        c = self.cfg
        self._sync_opt()  # the gradient pool is zeroed behind the asynchronous Adam
        self._join_aux()
        s, f, cam_idx = ctx["samples"], ctx["field"], ctx["cam_idx"]
        train_mlp = ctx["stage"] == STAGE_GLOBAL
        n_dev = s["n_dev"]
        d_sigma, d_rgb = self._timed("composite_bwd", ops.composite_bwd, s["pts_offsets"], f["sigma"], s["dists"], f["rgb"],
                                     g_rgb, g_acc, early_T=c.early_T)
        if ctx["fused"] and train_mlp:
            fp, bp = self._planes(s["ts"].shape[0])
            ops.fill_zero(self._gmax)
            self._timed("field_bwd", ops.grad_max, d_sigma, d_rgb, f["h0"], self._gmax, n_dev=n_dev)
            demb = self.emb_grad if c.use_appearance_embedding else None
            scattered = c.fused_backward
            if scattered:
                rec = self._record_buffers()
                # (the CU cap leaves CUs to the octree update + sampler on the second stream)
                self._timed("bwd_scatter", ops.field_bwd_scatter_x3, d_sigma, d_rgb, f["rgb"], f["h0"], f["masks"], self._x3()[1],
                            s["warped_pts"], s["anchors"], self.prim, self.bias_arg, self.n_volumes, c.log2_T, self.table_grad,
                            gmax=self._gmax, planes=bp, ray_id=s["ray_id"], cam_idx=cam_idx, demb=demb, n_dev=n_dev, records=rec,
                            opts=ops.launch_opts(decoder_cus=self._cu_cap(STAGE_GLOBAL), plain_rounding=not c.plane_dither))
                wg_side = None
                if c.wgrad_beside_accumulate and rec is not None and not self.sync.exchanges and self.seg is None:
                    if self._wg_stream is None:
                        self._wg_stream = ops.concurrent_stream(self.dev, role="wgrad")
                    wg_side = self._wg_stream
                    wg_side.wait_stream(torch.cuda.current_stream())  # forks behind the backward kernel: planes and records are final
                if rec is not None:
                    self._reduce_records(rec)
                dfeat = None
                # data parallel: the table gradient is final here -- its exchange runs beside the weight-gradient kernels below
                live = self._grad_live()
                self._eager(lambda: self.sync.start_grads([live]))
                ctx["table_exchange_started"] = True
            else:
                dfeat = self._timed("field_bwd", ops.field_bwd_x3, d_sigma, d_rgb, f["rgb"], f["h0"], f["masks"], self._x3()[1], self._gmax,
                                    bp, ray_id=s["ray_id"], cam_idx=cam_idx, demb=demb, n_dev=n_dev, opts=self._plane_opts(bp))
            g = self.layer_grads
            def pv(bufs, ld, k, table):  # (hi_flat, lo_flat | None, width) of plane k
                hi, w = ops.plane_flat(bufs[0], ld, k, table)
                return hi, (None if bufs[1] is None else ops.plane_flat(bufs[1], ld, k, table)[0]), w
            fv = lambda k: pv(fp, ops.FIELD_X3_FWD_PLANE_LD, k, ops.FWD_PLANE)  # noqa: E731
            bld = ops.FIELD_X3_BWD_PLANE_LD if bp[1] is not None else ops.FIELD_X3_BWD_PLANE_LD_RNE

            def bv(k):
                hi, lo, w = pv(bp, bld, k, ops.BWD_PLANE)
                if k == "dza1" and lo is None:  # the one-plane form keeps this plane's lo' half, in the same buffer
                    lo = ops.plane_flat(bp[0], bld, "dza1_lo", ops.BWD_PLANE)[0]
                return hi, lo, w
            n_cap = s["ts"].shape[0]
            if self._wgrad_parts is None:
                self._wgrad_parts = ops.wgrad_parts(self.mlp_grad.numel(), self.dev)
            layers = [(bv(dzk), fv(ak), g[li][0], g[li][1], mv, cmap)
                      for li, dzk, ak, mv, cmap in ((4, "dz3", "c2", 3, None), (3, "dz2", "c1", 128, None), (2, "dz1", "xp", 128, self._xp_col_map),
                                                    (1, "dh", "a1", 16, None), (0, "dza1", "feat", 128, None))]
            if scattered and wg_side is not None:
                with torch.cuda.stream(wg_side):
                    ops.wgrad_x3_layers(layers, self.mlp_grad, n_cap, n_dev=n_dev, gmax=self._gmax, parts=self._wgrad_parts)
                torch.cuda.current_stream().wait_stream(wg_side)
            else:
                self._timed("wgrad", ops.wgrad_x3_layers, layers, self.mlp_grad, n_cap, n_dev=n_dev, gmax=self._gmax, parts=self._wgrad_parts)
        elif ctx["fused"] and c.fused_backward and c.mlp_precision == "x3":
            rec = self._record_buffers()
            if rec is None or c.record_merge >= 0:
                lo = ops.launch_opts(decoder_cus=self._cu_cap(), record_merge=max(c.record_merge, 0) if rec is not None else 0)
            else:
                lo = ops.launch_opts(decoder_cus=self._cu_cap(), merge_gate=self._merge_gate(f.get("front"), n_dev))
            self._timed("bwd_scatter", ops.field_bwd_scatter_x3, d_sigma, d_rgb, f["rgb"], f["h0"], f["masks"], self._x3()[1],
                        s["warped_pts"], s["anchors"], self.block_prims[ctx["block"]], self.bias_arg, self.n_volumes, c.log2_T,
                        self.table_grad, n_dev=n_dev, records=rec, sample_list=f.get("front"), opts=lo)
            if rec is not None:
                self._reduce_records(rec)
            dfeat = None
        elif ctx["fused"]:
            prec = c.mlp_precision
            dfeat = self._timed("field_bwd", ops.field_bwd_dx, d_sigma, d_rgb, f["rgb"], f["h0"], f["masks"], self.layers, n_dev=n_dev,
                                precision=prec, image=self._x3()[1] if prec == "x3" else None)
        else:
            (bW0, _), (bW1, _), (hW0, _), (hW1, _), (hW2, _) = self.layers
            g = self.layer_grads
            gw = lambda i: (g[i][0], g[i][1]) if train_mlp else (None, None)  # noqa: E731
            dc2, _, _ = self._timed("head_bwd", ops.linear_bwd, f["c2"], hW2, f["rgb"], d_rgb, "sigmoid", dW=gw(4)[0], db=gw(4)[1], need_dw=train_mlp, n_dev=n_dev)
            dc1, _, _ = self._timed("head_bwd", ops.linear_bwd, f["c1"], hW1, f["c2"], dc2, "relu", dW=gw(3)[0], db=gw(3)[1], need_dw=train_mlp, n_dev=n_dev)
            dxh, _, _ = self._timed("head_bwd", ops.linear_bwd, f["xh"], hW0, f["c1"], dc1, "relu", dW=gw(2)[0], db=gw(2)[1], need_dw=train_mlp, n_dev=n_dev, in_dim=63)
            demb = self.emb_grad if (train_mlp and c.use_appearance_embedding) else None
            dh = ops.head_unpack_bwd(f["h"], d_sigma, dxh, cam_idx, s["pts_offsets"], demb=demb, n_dev=n_dev)
            da1, _, _ = self._timed("base_bwd", ops.linear_bwd, f["a1"], bW1, f["h"], dh, "none", dW=gw(1)[0], db=gw(1)[1], need_dw=train_mlp, n_dev=n_dev)
            dfeat, _, _ = self._timed("base_bwd", ops.linear_bwd, f["feat"], bW0, f["a1"], da1, "relu", dW=gw(0)[0], db=gw(0)[1], need_dw=train_mlp, n_dev=n_dev)
        # both tables see the same d(feat) (sum of two lookups); only the trainable one is scattered, through ITS primes
        if dfeat is None:
            return  # already scattered by the fused kernel
        prim = self.prim if train_mlp else self.block_prims[ctx["block"]]
        self._timed("hash_bwd", ops.hash_bwd, prim, self.bias_arg, s["warped_pts"], s["anchors"], dfeat, self.n_volumes,
                    c.log2_T, grad_pool=self.table_grad, n_dev=n_dev, grad_f16=c.grad_f16)

    def forward_backward(self, rays_o, rays_d_raw, cam_idx, gt_rgb, noise, stage=STAGE_GLOBAL, block=None, s3im_index=None, prefetch=None,
                         update_oct=False, samples=None, prefetch_out=None):
        """One loss evaluation + all parameter gradients (no optimiser step).  Returns (loss[1], aux).
        s3im_index: the arange + randperm index vector of S3IM.forward (losses.py:780-789); drawn here when absent.
        prefetch: (rays_o, rays_d_raw, noise) of the NEXT step -- its sampler runs beside this step's backward (focal stage; global
        stage only together with update_oct).  update_oct (global stage): run UpdateOctNodes on this step's samples right after
        the forward, like GFNeRFModel.get_outputs does."""
        c = self.cfg
        # (UpdateOctNodes reads every sample's alpha: see forward_train)
        comp, ctx = self.forward_train(rays_o, rays_d_raw, cam_idx, noise, stage, block, want_depth=False, samples=samples,
                                       all_alphas=update_oct and stage == STAGE_GLOBAL)
        fwd_done = None
        segmented = self.seg is not None
        if c.loss_before_fork and prefetch is not None and not segmented:
            fwd_done = torch.cuda.Event()
            fwd_done.record()  # the side branch (octree update, next sampler) starts behind the forward, whatever is enqueued first
        if segmented and stage == STAGE_GLOBAL and update_oct:
            # a segmented capture cannot end a segment while a branch is open: K5 -> MAX exchange -> K6 run on THIS stream, between the
            # forward and the loss; the side branch keeps the next batch's sampler
            self._timed("oct_update", self.update_oct_nodes, ctx)

        def fork():
            if stage == STAGE_GLOBAL and update_oct and not segmented:
                if prefetch is not None:
                    self.prefetch_sample(*prefetch, first=lambda: self._timed("oct_update", self.update_oct_nodes, ctx), out=prefetch_out,
                                         after=fwd_done)
                else:
                    self._timed("oct_update", self.update_oct_nodes, ctx)
            elif prefetch is not None and (stage == STAGE_FOCAL or (segmented and update_oct)):
                self.prefetch_sample(*prefetch, out=prefetch_out, after=fwd_done)

        if not c.loss_before_fork:
            fork()
        loss, g_rgb = self._timed("loss", ops.charbonnier, comp["rgb"], gt_rgb, loss=ctx["samples"].pop("loss0", None))  # rgb_loss (nerfacto.py:636)
        R = gt_rgb.shape[0]
        if c.s3im_loss_mult > 0 and (R * c.s3im_repeat_time) % c.s3im_patch_height == 0 and R * c.s3im_repeat_time >= 4 * c.s3im_patch_height:
            if s3im_index is None:
                s3im_index = ops.s3im_index(R, c.s3im_repeat_time, device=self.dev)
            self._timed("loss", ops.s3im, gt_rgb, comp["rgb"], s3im_index, patch_h=c.s3im_patch_height, mult=c.s3im_loss_mult,
                        loss=loss, g_pred=g_rgb)
        if c.loss_before_fork:
            fork()
        self.backward(ctx, g_rgb)
        return loss, ctx

    def adam_groups(self, stage, block=None):
        """The parameter groups optimizer_step(stage, block) steps, with their learning rates (nerfacto.py:448-488)."""
        c = self.cfg
        if stage == STAGE_GLOBAL:
            return [("table", c.lr), ("mlp", c.lr)] + ([("emb", c.lr)] if c.use_appearance_embedding else [])
        return [(f"block{block}", c.lr_block)]

    def optimizer_step(self, stage=STAGE_GLOBAL, block=None, hyper=None):
        """hyper: {group name: device f32 [2]} -- the step-dependent Adam factors read from the device instead of being passed
        by value (a captured step; the host refreshes them before every replay and advances the group's step count itself)."""
        c = self.cfg
        hy = (lambda name: hyper[name]) if hyper is not None else (lambda name: None)
        self.step += 1
        # only elements [0, 17T) of a table are addressable (level-overlap quirk); the rest never receive a
        # gradient, so dense Adam leaves them untouched and they are skipped exactly
        n_live = 17 * (1 << c.log2_T)
        if stage == STAGE_GLOBAL:
            m, v = self._adam_state("table", self.table)
            self._adam_table("table", self.table, m, v, c.lr, self.table16, hy("table"))
            self._x3_stale = True
            for name, p, gr in (("mlp", self.mlp_flat, self.mlp_grad), ("emb", self.emb, self.emb_grad)):
                if name == "emb" and not c.use_appearance_embedding:
                    continue
                mm, vv = self._adam_state(name, p)
                self._timed("adam", ops.adam_step, p, gr, mm, vv, self._adam_tick(name), lr=c.lr, eps=c.adam_eps, hyper=hy(name))
        else:
            p = self.blocks[block]
            m, v = self._adam_state(f"block{block}", p)
            run = lambda: self._adam_table(f"block{block}", p, m, v, c.lr_block, self.blocks16[block] if c.table_f16_shadow else None,  # noqa: E731
                                           hy(f"block{block}"))
            if c.async_adam and c.level_major:
                if self._opt_stream is None:
                    self._opt_stream = ops.concurrent_stream(self.dev, role="adam")
                self._opt_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._opt_stream):
                    run()
                    ops.fill_zero(self._grad_live())  # the next step's zero_grads
                self._opt_pending = True
            else:
                run()

    # ------------------------------------------------------- optimiser state owned by the caller (the plugin surface)
    def fused_claims(self, p):
        """True when a live gfnerf.optim.GfnAdam steps parameter tensor `p` (it reads the engine's gradient buffers: no dense
        autograd gradient is wanted for p)."""
        return any(opt.owns(p) for opt in list(self.fused_optimizers))

    def layer_grad_of(self, p):
        """The engine's gradient view of decoder tensor `p` (a view of mlp_flat)."""
        off = (p.data_ptr() - self.mlp_flat.data_ptr()) // 4
        return self.mlp_grad[off:off + p.numel()].view(p.shape)

    def decoder_changed(self):
        self._x3_stale = True

    def external_adam(self, name, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-15, grad=None):
        """One torch.optim.Adam step of parameter group `name` ("table", "block<i>", "mlp", "emb") whose state (m, v, step count
        t) lives with the caller -- the Trainer's torch.optim.Optimizer (engine/optimizers.py:73-133; gfnerf.optim.GfnAdam).
        grad: None = the buffer the last backward filled (a table: the f16 x128 pool); a tensor = a dense gradient like `p`."""
        c = self.cfg
        self._sync_opt()
        if name == "table" or name.startswith("block"):
            i = None if name == "table" else int(name[5:])
            p = self.table if i is None else self.blocks[i]
            shadow = None
            if c.table_f16_shadow:
                shadow = self.table16 if i is None else self.blocks16[i]
            g = self.table_grad if grad is None else grad.view(p.shape)
            self._timed("adam", ops.adam_step_table, p, g, m, v, t, lr=lr, beta1=beta1, beta2=beta2, eps=eps, shadow_f16=shadow,
                        n=17 * (1 << c.log2_T))
            self.step += 1  # one per training step of either stage (forward_train reads it: level_major_global_after)
        elif name == "mlp":
            self._timed("adam", ops.adam_step, self.mlp_flat, self.mlp_grad if grad is None else grad, m, v, t, lr=lr, beta1=beta1, beta2=beta2, eps=eps)
            self._x3_stale = True
        elif name == "emb":
            self._timed("adam", ops.adam_step, self.emb, self.emb_grad if grad is None else grad, m, v, t, lr=lr, beta1=beta1, beta2=beta2, eps=eps)
        else:
            raise _lib.GfnError(f"external_adam: unknown parameter group {name!r}")

    def grads(self, stage=STAGE_GLOBAL):
        """Gradient tensors that data-parallel training must all-reduce before optimizer_step."""
        self._sync_opt()
        # only elements [0, 17T) of a table can receive a gradient (level-overlap quirk): reduce 71 MB, not 134 MB
        live = self._grad_live()
        if stage == STAGE_GLOBAL:
            out = [live, self.mlp_grad]
            if self.cfg.use_appearance_embedding:
                out.append(self.emb_grad)
            return out
        return [live]

    def _eager(self, fn):
        """An exchange point of the data-parallel step: a collective, which is never captured.  Eager step: run it.  Segmented
        capture (gfnerf_amd.stepgraph): the graph segment ends here, fn is replayed between this segment and the next."""
        if self.seg is None:
            fn()
        else:
            self.seg(fn)

    def sync_grads(self, stage, ctx=None):
        """Data-parallel global stage (scripts/train.py:124-137, gf_pipeline.py:136-138 -- DDP there, which misses the hash
        table): average this step's gradients over the ranks before Adam.  The table exchange may already be in flight
        (backward starts it as soon as the table gradient is final); the decoder / embedding gradients follow here.  The focal
        stage trains one block table per rank with everything shared frozen: nothing to exchange."""
        if stage != STAGE_GLOBAL or (not self.sync.exchanges and self.seg is None):
            return
        g = self.grads(stage)
        if ctx is not None and ctx.pop("table_exchange_started", False):
            self._eager(lambda: self.sync.finish_grads(g[1:]))
        else:
            self._eager(lambda: (self.sync.start_grads(g[:1]), self.sync.finish_grads(g[1:])))

    def train_step(self, rays_o, rays_d_raw, cam_idx, gt_rgb, noise, stage=STAGE_GLOBAL, block=None, s3im_index=None,
                   prefetch=None, update_oct=False, samples=None, prefetch_out=None, hyper=None):
        """Trainer.train_iteration (trainer.py:382-441) restricted to the hot path.  samples / prefetch_out / hyper: see
        gfnerf_amd.stepgraph (the step captured in a hipGraph)."""
        loss, aux = self.forward_backward(rays_o, rays_d_raw, cam_idx, gt_rgb, noise, stage, block, s3im_index=s3im_index, prefetch=prefetch,
                                          update_oct=update_oct, samples=samples, prefetch_out=prefetch_out)
        self.sync_grads(stage, aux)
        self.optimizer_step(stage, block, hyper=hyper)
        return loss, aux

    def update_oct_nodes(self, aux):
        """persampler.update_oct_nodes (nerfacto.py:606-613) on the packed samples of the last step.  Data parallel: the
        per-node maxima K5 produced from this rank's rays are MAX-reduced over the ranks before the statistics update and K6."""
        s, comp = aux["samples"], aux["comp"]
        exchange = None
        if self.sync.exchanges or self.seg is not None:
            exchange = lambda ts: self._eager(lambda: self.sync.reduce_octree_stats(ts))  # noqa: E731
        ops.update_nodes(s["pts_offsets"], s["anchors"], comp["weights"], comp["alphas"], self.weight_stats,
                         self.alpha_stats, self.visit_cnt, self.nodes, reduce_max=exchange)
        ops.tree_compact(self.nodes, self.nodes_compact)  # K6 may have invalidated leaves

    def _drop_stale_samples(self):
        """The tree is about to change: samples marched through the old tree (a prefetched next batch, the last step's
        context) carry old node indices and must not be collected or fed to UpdateOctNodes."""
        if self._pre is not None:
            torch.cuda.current_stream().wait_stream(self._side)  # the prefetch still owns the sampler workspace
            self._pre = None
        self._last_train_ctx = None
        self.forget_graphs()


    def proc_octree(self, compact=True, subdivide=False, brute_force=False):
        """PersOctree::ProcOctree (PersSampler.cpp:154-417) on this path's tree, on the DEVICE (csrc/octree_proc.hip; the reference
        copies the blob to the host and back, PersSampler_cuda.cu:667-676): only the new node count is read back.  Node indices
        change; hash volumes (trans_idx) do not.  Then the 64-byte sampler image is rebuilt."""
        if not compact:
            raise _lib.GfnError("proc_octree: compact = False is not a state the reference can be in (PersSampler_cuda.cu:668-675)")
        self._drop_stale_samples()
        self.nodes, self.weight_stats, self.alpha_stats = ops.octree_proc(self.nodes, self.weight_stats, self.alpha_stats, self.visit_cnt,
                                                                          subdivide=subdivide, brute_force=brute_force)
        self.n_nodes = self.nodes.numel() // ops.TREE_NODE_BYTES
        self.visit_cnt = torch.zeros(self.n_nodes, dtype=torch.int64, device=self.dev)  # :415
        self.nodes_compact = ops.tree_compact(self.nodes)
        self._sampler_ws = None

    def mark_invisible_nodes(self, intri, w2c, bounds):
        """K7 MarkInvisibleNodes (PersSampler_cuda.cu:709-742)."""
        self._drop_stale_samples()
        ops.mark_invisible_nodes(self.nodes, intri, w2c, bounds)
        ops.tree_compact(self.nodes, self.nodes_compact)


def fineness_at(step, init=16.0, decay_end=10000):
    """PersSampler::UpdateRayMarch (PersSampler.cpp:958-967)."""
    if step >= decay_end:
        return 1.0
    p = float(step) / float(decay_end)
    return math.exp(math.log(1.0) * p + math.log(init) * (1.0 - p))

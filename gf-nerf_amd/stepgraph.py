"""A training step of the hot path as a hipGraph replay.

HotPath.train_step enqueues ~45 launches, fills and stream hand-offs per step from Python.  On a quiet host that costs
0.5 ms of CPU time per step and hides behind the 1.7 - 2.4 ms the GPU needs; on a busy host (the CPUs of a GPU box are
shared) it does not, and the step becomes host-bound (measured: 2.5 - 3.7 ms/step instead of 1.74 on boxes whose host took
1.4 - 3.4 ms to enqueue one).  The step has no host synchronisation and no data-dependent launch, so it is capturable: this
module captures it ONCE (twice: the sampler ping-pong below) with torch.cuda.graphs and replays it -- one launch per step,
80 us of host time.

What a replay contains (`StepGraph.step`), exactly one of each like the eager step:
    side branch:  the NEXT step's S3IM permutations and march noise drawn on the device -> [global stage: UpdateOctNodes of
                  THIS step, right after the forward] -> sampler of the NEXT batch's rays into the other sample buffer
    main branch:  forward on THIS batch's samples -> loss -> backward -> table-gradient accumulation -> Adam
The graphs G[0], G[1] differ only in which of the two static sample buffers S[0], S[1] is read and which is written
(replay k reads S[k % 2], its side branch fills S[(k + 1) % 2] for replay k + 1); with several focal blocks trained in
rotation (`block=[0, 1, 2, 3]`: step k trains block k mod 4) there is one graph per position of the cycle lcm(2, blocks).  Inputs live in static device buffers the
caller refills between replays (`cam / gt` of this batch, `o / d` of the next); Adam's step-dependent factors are read from
a device vector the host refreshes before every replay (gfn_adam_hyper_host), so bias correction and learning-rate schedule
advance although the launches are frozen.

The data-parallel global step (world > 1) is captured in SEGMENTS: collectives are not captured (gloo cannot be, and an RCCL call
inside a graph would be this stack's first), so the capture ends at each of the step's three exchange points, notes the collective as
an eager call, and begins the next segment:
    graph 1: hand-over, forward, K5        | MAX all-reduce of the octree statistics
    graph 2: K6, loss, backward + gradient accumulation (side branch: the next batch's sampler)   | table-gradient exchange starts (own stream)
    graph 3: decoder weight gradients      | small all-reduce of the decoder / embedding gradients, wait for the table exchange
    graph 4: Adam (side branch: nothing left)
A replay is 4 graph launches + the collectives instead of ~60 launches.  `segmented=True` with one rank breaks at the same points
with identity exchanges (tests).  A tree that changes size (ProcOctree every compact_freq steps) needs a re-capture -- `StepGraph.invalidate`."""
import numpy as np
import torch

from . import _lib, hotpath as hp_mod, ops


class StepGraph:
    def __init__(self, hp, stage, n_rays, fineness, block=None, update_oct=False, train_noise=True, lr_schedule=None, draws_after_lookup=True,
                 replan_every=2048, segmented=None, seed=None):
        """lr_schedule: step -> learning-rate multiplier of the stage's groups (None: constant).
        replan_every: the phased (termination-aware) forward costs a handful of small launches even when its plan is "one phase, every
        sample" (an empty scene: nothing terminates).  At capture the plan is read back once (a forward of the batch at hand, no
        training step) and the step is captured WITHOUT the phase machinery when it says so; after this many replays the step is
        captured again, so that the choice follows the scene as it fills."""
        self.segmented = (hp.sync.exchanges and stage == hp_mod.STAGE_GLOBAL) if segmented is None else bool(segmented)
        if hp.sync.exchanges and stage == hp_mod.STAGE_GLOBAL and not self.segmented:
            raise _lib.GfnError("StepGraph: the data-parallel global step exchanges gradients with collectives; capture it segmented")
        self.hp, self.stage, self.R = hp, stage, int(n_rays)
        self.blocks = list(block) if isinstance(block, (list, tuple)) else [block]
        self.cycle = len(self.blocks) if len(self.blocks) % 2 == 0 else 2 * len(self.blocks)  # lcm(2, number of blocks)
        self.fineness, self.train_noise, self.update_oct = float(fineness), train_noise, update_oct and stage == hp_mod.STAGE_GLOBAL
        self.lr_schedule = lr_schedule
        self.draws_after_lookup = draws_after_lookup
        dev = hp.dev
        f = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)  # noqa: E731
        self.cam = torch.zeros(self.R, dtype=torch.int32, device=dev)
        self.gt = f(self.R, 3)
        self.nxt_o, self.nxt_d = f(self.R, 3), f(self.R, 3)
        self.nxt_d[:, 2] = -1.0
        self.groups = []
        for b in self.blocks:
            self.groups += [g for g in hp.adam_groups(stage, b) if g not in self.groups]
        self.hyper_dev = f(len(self.groups), 2)
        self.hyper = {name: self.hyper_dev[i] for i, (name, _) in enumerate(self.groups)}
        self._ring = [torch.zeros((len(self.groups), 2), dtype=torch.float32).pin_memory() for _ in range(256)]
        self._ring_ev, self._ring_pos = [None] * len(self._ring), 0
        self.noise = torch.zeros(1024 + self.R + 10, dtype=torch.float32, device=dev)  # the NEXT batch's march noise
        self.rng = ops.StepDraws(self.R, hp.cfg.s3im_repeat_time, seed=seed, device=dev)  # the device's own Philox stream: torch's generator
        # costs a captured step two fills in front of every replay
        self.S = [None, None]
        self.s3 = [None, None]  # S3IM permutations of the step that reads S[p], drawn one step ahead like the samples
        self.G = [None] * self.cycle
        self.loss = [None] * self.cycle
        self.aux = [None] * self.cycle
        self.p = 0  # position in the cycle of the next replay
        self.primed = False
        self._tree = None
        self.replan_every, self._since_capture, self.phased = int(replan_every), 0, None
        self._k_host = torch.full((1,), -1, dtype=torch.int32).pin_memory()  # the plan's K as the replays leave it (copied inside the graph)
        self._idle = 0
        self._lm_captured = None  # global stage: did the captured forward take the level-major kernels (HotPath.level_major_global_now)?
        self.captures = 0  # how often the step was captured (the first time, tree changes, re-plans of the phases)

    # ------------------------------------------------------------------ set-up
    def _noise(self, out=None):
        return ops.sampler_noise(self.R, self.fineness, train=self.train_noise, device=self.hp.dev, out=out)

    def prime(self, rays_o, rays_d):
        """Sample the FIRST batch's rays (eagerly) into S[0]; from then on every replay samples the next batch itself."""
        hp = self.hp
        for k in range(2):  # both buffers exist before anything is captured (same shapes, same options)
            self.nxt_o.copy_(rays_o); self.nxt_d.copy_(rays_d)
            self.S[k] = hp.sample(self.nxt_o, self.nxt_d, self._noise(), out=self.S[k])
            if hp.cfg.s3im_loss_mult > 0:
                self.s3[k] = ops.s3im_index(self.R, hp.cfg.s3im_repeat_time, device=hp.dev, out=self.s3[k])
        self._noise(out=self.noise)  # (eval mode: the constant the replays read; training mode: rewritten by every replay's draws)
        self.p, self.primed = 0, True  # the first replay reads S[0]

    def _step_eager_or_captured(self, j):
        """The launches of step j of the cycle: block blocks[j mod n], reading S[j mod 2] and filling the other buffer (called
        under capture)."""
        hp = self.hp
        p, block = j % 2, self.blocks[j % len(self.blocks)]
        main = torch.cuda.current_stream()
        if hp._side is None:
            hp._side = hp._new_side_stream()
        side = hp._side

        def draws():  # the NEXT step's random draws, on the device: started behind this step's lookups (a sort beside them
            side.wait_stream(main)  # costs the gathers 10 %), beside the decoder, ahead of the next batch's sampler
            with torch.cuda.stream(side):
                s3 = self.s3[1 - p] if hp.cfg.s3im_loss_mult > 0 else None
                if self.train_noise or s3 is not None:  # (eval-mode noise is the constant prime() wrote)
                    self.rng.draw(noise_out=self.noise if self.train_noise else None, fineness=self.fineness, s3im_out=s3, write_arange=False)

        if self.draws_after_lookup:
            hp.after_lookup = draws
        else:
            draws()
        hp.k_host = self._k_host  # the planned K travels to the host with every replay (no synchronisation: step() reads it late)
        try:
            loss, aux = hp.train_step(None, None, self.cam, self.gt, None, stage=self.stage, block=block, s3im_index=self.s3[p],
                                      prefetch=(self.nxt_o, self.nxt_d, self.noise), update_oct=self.update_oct, samples=self.S[p],
                                      prefetch_out=self.S[1 - p], hyper=self.hyper)
        finally:
            hp.after_lookup, hp.k_host = None, None
        assert hp._pre is not None and hp._pre[4]["pts_offsets"] is self.S[1 - p]["pts_offsets"], "the prefetch must fill the other sample buffer"
        self.S[1 - p] = hp._pre[4]
        hp._pre = None  # collected: the next replay reads it as its `samples`
        if not self.segmented:
            main.wait_stream(side)  # (segmented: the capture joins whatever branch is open when a segment ends)
        return loss, aux

    def capture(self):
        """One capture per parity.  Everything a step creates lazily and keeps (gradient pool, record streams, activation planes,
        Adam moments, kernel attributes, weight images) must exist already -- a buffer first allocated under capture would live in
        the graph's pool and be re-initialised by every replay -- so at least one EAGER train_step of this stage comes first."""
        assert self.primed, "StepGraph.prime(first batch's rays) before the first step"
        hp = self.hp
        self.captures += 1
        if hp.table_grad is None or (self.stage == hp_mod.STAGE_GLOBAL and hp._plane_bufs is None and hp.cfg.fused_field):
            raise _lib.GfnError("StepGraph: run one eager HotPath.train_step of this stage before capturing (lazily created buffers)")
        for name, _ in self.groups:
            hp._adam_state(name, self._param_of(name))
        hp.tree_depth()  # (synchronises when the tree is new: not under capture)
        self._refresh_hyper(advance=False)
        # does a termination-aware split pay on this scene right now?  Plan it from the batch at hand (forward only), read the plan back.
        # The verdict ("one phase") is this capture's business only: hp.phase_override is set around the capture and put back, so that
        # other users of the engine (an eager forward_train, the plugin's forward_train_graphed, procedural.train_and_eval) keep the
        # termination-aware forward; the replays have it baked in.
        outer_override = hp.phase_override
        try:
            hp.phase_override = None
            self.phased = None
            if hp.cfg.two_phase and hp.cfg.early_T > 0:
                j0 = self.p
                stale = hp._x3_stale  # (the forward repacks stale decoder weight images: the captured step must still do that itself)
                hp.forward_train(None, None, self.cam, None, stage=self.stage, block=self.blocks[j0 % len(self.blocks)], want_depth=False, samples=self.S[j0 % 2],
                                 all_alphas=self.update_oct)  # (the forward the captured step will run)
                hp._x3_stale = stale
                if hp.last_forward == "lm2":
                    self.phased = int(hp._phase.k.item()) < 1024
                    hp.phase_override = None if self.phased else False
                    if self.stage != hp_mod.STAGE_GLOBAL:  # how long the backward will be: the CUs it keeps beside the sampler (HotPath._cu_cap)
                        hp.front_override = (int(hp._phase.stats[4].item()), int(self.S[j0 % 2]["pts_offsets"][-1].item()))
            self._since_capture, self._idle = 0, 0
            self._lm_captured = hp.level_major_global_now(self.update_oct) if self.stage == hp_mod.STAGE_GLOBAL else None
            self._k_host.fill_(-1)
            hp._join_aux()  # (a branch left open by an eager forward must not be joined under capture)
            torch.cuda.synchronize()
            steps_before, ticks_before = hp.step, dict(hp._adam_t)
            pool = None
            for j in range(self.cycle):
                if self.segmented:
                    self.G[j], pool = self._capture_segments(j, pool)
                    continue
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):  # (a communicator's watchdog thread may poll events meanwhile)
                    self.loss[j], self.aux[j] = self._step_eager_or_captured(j)
                pool = g.pool()
                self.G[j] = g
            hp.step, hp._adam_t = steps_before, ticks_before  # capturing executes nothing: the counters must not move
        finally:
            hp.phase_override = outer_override
            hp.front_override = None
        self._tree = (hp.nodes.data_ptr(), hp.n_nodes)

    def _capture_segments(self, j, pool):
        """Step j as [graph, eager call, graph, ...]: hp.seg (called by HotPath at every exchange point) closes the segment being
        captured -- open branches joined first -- and opens the next one in the same memory pool."""
        hp = self.hp
        segs, state = [], {}
        cur = torch.cuda.current_stream()
        cap = torch.cuda.Stream(device=hp.dev)
        cap.wait_stream(cur)

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=state.get("pool"), capture_error_mode="thread_local")
            state["g"] = g

        def end():
            hp._join_aux()
            if hp._side is not None:
                with torch.cuda.stream(hp._side):
                    forked = torch.cuda.is_current_stream_capturing()
                if forked:
                    cap.wait_stream(hp._side)
            state["g"].capture_end()
            state["pool"] = state["g"].pool()
            segs.append(state["g"])

        def seg(fn):
            end()
            segs.append(fn)
            begin()

        state["pool"] = pool
        with torch.cuda.stream(cap):
            begin()
            hp.seg = seg
            try:
                self.loss[j], self.aux[j] = self._step_eager_or_captured(j)
            finally:
                hp.seg = None
                end()
        cur.wait_stream(cap)
        return segs, state["pool"]

    def invalidate(self):
        """The tree (or anything else baked into the launches) changed: capture again at the next step."""
        self.G = [None] * self.cycle

    # ------------------------------------------------------------------ per step
    def _refresh_hyper(self, advance=True, ride_with=None):
        """Adam's factors of the NEXT step of every group; advance: the groups of the step about to be replayed take it.
        ride_with: (dst, src) pairs of the batch hand-over -- the factors then travel in the same launch (gfn_copy_list reads the pinned
        block in place) instead of a host-to-device copy of their own in front of the replay."""
        hp = self.hp
        step_for_lr = hp.step
        active = {n for n, _ in hp.adam_groups(self.stage, self.blocks[self.p % len(self.blocks)])}
        # a ring of pinned blocks: the copy below executes when the GPU gets there, steps after the host wrote it -- one buffer
        # written again by the next call would hand a replay the factors of a LATER step.  A block is written again only once the
        # copy that read it has run (its event), which also keeps the host at most a ring ahead of the device.
        k = self._ring_pos
        self._ring_pos = (k + 1) % len(self._ring)
        if self._ring_ev[k] is not None:
            self._ring_ev[k].synchronize()
        host = self._ring[k]
        for i, (name, lr) in enumerate(self.groups):
            if name not in hp._adam_t:
                hp._adam_state(name, self._param_of(name))
            t = hp._adam_t[name] + 1
            if advance and name in active:
                hp._adam_t[name] = t
            mult = 1.0 if self.lr_schedule is None else float(self.lr_schedule(step_for_lr))
            host[i] = torch.from_numpy(ops.adam_hyper(t, lr * mult))
        if ride_with is not None:
            ops.copy_list(list(ride_with) + [(self.hyper_dev, host)])
        else:
            self.hyper_dev.copy_(host, non_blocking=True)
        self._ring_ev[k] = torch.cuda.Event()
        self._ring_ev[k].record()
        if advance:
            hp.step += 1

    def _param_of(self, name):
        hp = self.hp
        return {"table": hp.table, "mlp": hp.mlp_flat, "emb": hp.emb}.get(name) if not name.startswith("block") else hp.blocks[int(name[5:])]

    def step(self, cam_idx, gt_rgb, next_rays_o, next_rays_d):
        """One training step on the batch whose rays the previous step (or prime) sampled; next_rays_*: the rays of the batch
        the NEXT call will train on.  -> (loss [1] device tensor of this step, aux) -- static buffers, overwritten by the next call."""
        hp = self.hp
        self._since_capture += 1
        if self.phased:  # captured with the phase machinery: still worth it?  (the plan of a few replays ago, read without waiting)
            self._idle = self._idle + 1 if int(self._k_host[0]) >= 1024 else 0
        # (self.phased is None: the phased forward cannot apply here -- early_T = 0, or not the level-major forward -- and a re-capture
        # would cost a full synchronisation and capture for the same graph)
        if self.phased is not None and hp.cfg.two_phase and hp.cfg.early_T > 0 and (self._since_capture > self.replan_every or self._idle >= 16):
            self.invalidate()  # plan the phases (and choose the forward kernels) again: the scene has moved on
        if self.stage == hp_mod.STAGE_GLOBAL and self._lm_captured is not None and hp.level_major_global_now(self.update_oct) != self._lm_captured:
            self.invalidate()  # the step crossed cfg.level_major_global_after: the capture has the fused forward baked in
        if self.G[0] is None or self._tree != (hp.nodes.data_ptr(), hp.n_nodes):
            self.capture()
        # one launch in front of every replay (not four memcpy nodes and a host-to-device copy): the batch and Adam's factors
        self._refresh_hyper(advance=True, ride_with=[(self.cam, cam_idx.contiguous()), (self.gt, gt_rgb.contiguous()), (self.nxt_o, next_rays_o.contiguous()),
                                                     (self.nxt_d, next_rays_d.contiguous())])
        j = self.p
        if self.segmented:
            for x in self.G[j]:
                if isinstance(x, torch.cuda.CUDAGraph):
                    x.replay()
                else:
                    x()  # a collective between two segments
        else:
            self.G[j].replay()
        self.p = (j + 1) % self.cycle
        return self.loss[j], self.aux[j]

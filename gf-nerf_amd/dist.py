"""Multi-GPU layer of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests and the one-card rehearsal).  Rays are independent units end to end:

  * global stage -- ray data parallelism: every rank holds a replica of tree, global table, decoder and embedding and
    renders its own rays.  Two exchanges per step, both driven from HotPath through `TrainSync`:
      - the octree statistics K5 produces (per-node maxima over rays: w_adder / a_adder / mark / visit_cnt) are combined
        with an elementwise MAX before the statistics update + K6, so every replica invalidates the same leaves
        (the reference accumulates them with atomicMax, order independent: PersSampler_cuda.cu:518-574);
      - the gradients are averaged before Adam.  The 71 MB f16 table gradient goes through a DIRECT all-reduce: every rank
        sends shard j to rank j (all-to-all: 7 peers over their own xGMI links at once), sums its shard in f32
        (gfn_reduce_shards), and the shards are all-gathered back -- 2 x 7/8 of the buffer over 7 links in parallel, where
        a ring would push the same bytes through one link at a time.  It runs on a communication stream beside the
        weight-gradient kernels of the backward (start_grads after the table gradient is final, finish_grads before Adam).
      (the reference relies on DDP, which never sees the hash table: hash_3d_anchored.py:26,73-78; scripts/train.py:124-137,
      gf_pipeline.py:136-138.)
  * focal stage -- one residual block table per rank, rays drawn from that block's camera cluster; everything shared is
    frozen in this stage (nerfacto_field.py:460-461,528,549), so there is NO data-path collective; only stage boundaries
    talk (broadcast_from, gather_block_tables).
  * render -- frames go to the rank that owns the frame's block (frame_owner), one gather of finished images.
"""
import torch
import torch.distributed as dist


def world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def allreduce_mean(tensors, group=None, force=False):
    """In-place average over ranks of a list of gradient tensors (force: issue the collectives on a one-rank group too).  f16 tensors (the x128-scaled table-gradient pool) are summed in
    f32, chunk by chunk, and rounded to f16 once -- like the direct path (gfn_reduce_shards): no overflow however many ranks add
    up, no bits lost to a pre-scaled f16 sum; everything else is scaled by 1/world and summed in its own type."""
    w = world(group)
    if w == 1 and not force:
        return tensors
    for t in tensors:
        if t.dtype == torch.float16:
            flat = t.view(-1) if t.is_contiguous() else None
            if flat is None:
                raise ValueError("allreduce_mean: f16 gradients must be contiguous")
            for chunk in flat.split(1 << 24):
                f = chunk.float()
                dist.all_reduce(f, op=dist.ReduceOp.SUM, group=group)
                chunk.copy_(f.mul_(1.0 / w))
        else:
            t.mul_(1.0 / w)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return tensors


def allreduce_max_(tensors, group=None, force=False):
    """In-place elementwise max over ranks (octree node statistics / visit counts)."""
    if world(group) == 1 and not force:
        return tensors
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return tensors


def _hip_shard_mean(recv, n_shards, out):
    from . import ops

    ops.reduce_shards(recv, n_shards, out, scale=1.0 / n_shards)


def allreduce_mean_direct(t, group=None, recv=None, shard_mean=None, force=False):
    """Average the flat contiguous tensor `t` over ranks in place: all-to-all of shards, local f32 sum x 1/world of this
    rank's shard (shard_mean(recv, world, out_shard): the HIP kernel by default), all-gather of the reduced shards.
    t.numel() must be a multiple of 8 x world (callers fall back to allreduce_mean otherwise).  recv: scratch like t.
    force: run the three steps on a one-rank group too (the all-to-all and the all-gather are then copies through the communicator)."""
    w = world(group)
    if w == 1 and not force:
        return t
    n = t.numel()
    assert t.dim() == 1 and t.is_contiguous() and n % (8 * w) == 0, "allreduce_mean_direct: flat tensor of 8*world*k elements"
    shard = n // w
    recv = torch.empty_like(t) if recv is None else recv
    dist.all_to_all_single(recv, t, group=group)       # recv[i*shard:(i+1)*shard] = rank i's copy of MY shard
    mine = t[rank(group) * shard:(rank(group) + 1) * shard]
    (shard_mean or _hip_shard_mean)(recv, w, mine)     # written in place into my slice of t ...
    dist.all_gather_into_tensor(t, mine.clone(), group=group)  # ... and published to everyone (same bits on every rank)
    return t


def allreduce_selfcheck(n=1 << 20, device="cuda", group=None, shard_mean=None, force=False):
    """Start-up check of the gradient exchange on the live communicator: allreduce_mean_direct (all_to_all_single + gfn_reduce_shards
    + all_gather_into_tensor) of a seeded f16 buffer of n elements -- at the x128 scale of the table-gradient pool, with entries a
    plain f16 SUM over the ranks would overflow -- against dist.all_reduce of the same values in f32.  -> dict(ok, max_abs_err,
    n, dtype) (identical on every rank)."""
    w, r = world(group), rank(group)
    if w == 1 and not force:
        return dict(ok=True, max_abs_err=0.0, n=0, world=1)
    n = n // (8 * w) * (8 * w)
    g = torch.Generator(device="cpu").manual_seed(1234 + r)
    x32 = (torch.randn(n, generator=g) * 64.0)
    x32[:: 97] = 40000.0  # w of these would overflow an f16 sum (65504)
    x16 = x32.to(torch.float16).to(device)
    want = x16.float()
    dist.all_reduce(want, op=dist.ReduceOp.SUM, group=group)
    want = (want / w).to(torch.float16)
    got = allreduce_mean_direct(x16.clone(), group, shard_mean=shard_mean, force=force)
    err = (got.float() - want.float()).abs().max()
    worst = err.clone()
    dist.all_reduce(worst, op=dist.ReduceOp.MAX, group=group)
    finite = torch.isfinite(got.float()).all().to(torch.float32)
    dist.all_reduce(finite, op=dist.ReduceOp.MIN, group=group)
    tol = 32.0  # one f16 ulp at 40000 (the mean is rounded to f16 once on either side)
    return dict(ok=bool(worst.item() <= tol and finite.item() == 1.0), max_abs_err=float(worst.item()), n=int(n), world=w, dtype="f16")


def block_owner(block, world_size):
    """Rank that owns focal block `block` (one block per GPU, round robin beyond that)."""
    return block % world_size


def my_blocks(n_blocks, world_size=None, my_rank=None):
    w = world() if world_size is None else world_size
    r = rank() if my_rank is None else my_rank
    return [b for b in range(n_blocks) if block_owner(b, w) == r]


def frame_owner(frame_blocks, world_size):
    """Render routing: frame f is rendered by the rank that owns the frame's block (frame_blocks[f]: the split of the nearest
    training camera, perssampler.py:369-376) -- frame-parallel, no table moves.  -> list of owner ranks."""
    return [block_owner(int(b), world_size) for b in frame_blocks]


def frame_shard_by_block(frame_blocks, world_size, my_rank):
    """Frames of a camera path this rank renders."""
    return [f for f, o in enumerate(frame_owner(frame_blocks, world_size)) if o == my_rank]


def broadcast_from(tensors, src=0, group=None):
    """Stage boundary: publish the (now frozen) global table / decoder from `src` to every rank."""
    if world(group) == 1:
        return tensors
    for t in tensors:
        dist.broadcast(t, src=src, group=group)
    return tensors


def gather_block_tables(local_tables, n_blocks, dst=0, group=None, shape=None, dtype=torch.float32, device=None):
    """Checkpoint: collect every rank's block tables on `dst` -> {block: tensor} (None elsewhere).
    local_tables: {block: tensor} owned by this rank.  shape / dtype / device describe a table for a `dst` that owns none
    (n_blocks < world); by default they are read off a local table."""
    w, r = world(group), rank(group)
    if w == 1:
        return dict(local_tables)
    out = {} if r == dst else None
    if r == dst and local_tables:
        like = next(iter(local_tables.values()))
        shape, dtype, device = (shape or tuple(like.shape)), like.dtype, like.device
    for b in range(n_blocks):
        owner = block_owner(b, w)
        if owner == dst:
            if r == dst:
                out[b] = local_tables[b]
            continue
        if r == owner:
            dist.send(local_tables[b], dst=dst, group=group)
        elif r == dst:
            if shape is None:
                raise ValueError("gather_block_tables: dst owns no block; pass shape= (16 << log2_T, 2)")
            buf = torch.empty(shape, dtype=dtype, device=device)
            dist.recv(buf, src=owner, group=group)
            out[b] = buf
    return out


class TrainSync:
    """What HotPath calls during a data-parallel training step.  Built once per process after init_process_group; with one
    rank every method is the identity, so the single-GPU step takes the same code path.

        sync = TrainSync()                      # default group
        hp = HotPath(..., sync=sync)
        hp.train_step(...)                      # K5 -> reduce_octree_stats -> K6 ; backward -> start_grads -> wgrad -> finish_grads -> Adam

    overlap: run the gradient exchange on its own stream (device tensors only).  direct: use the all-to-all based
    all-reduce for big flat buffers (None: when the backend has all_to_all, i.e. always on nccl/RCCL).  shard_mean: see
    allreduce_mean_direct (the CPU tests inject a torch restatement; the product default is the HIP kernel)."""

    DIRECT_MIN_ELEMS = 1 << 20

    def __init__(self, group=None, overlap=True, direct=None, shard_mean=None, single=False, force_exchange=False):
        """single: behave as a one-rank job whatever the process group says (a rank's purely local step, e.g. in tests).
        force_exchange: issue every collective of the step even when the group has ONE rank (the world-1 shortcuts off): the
        communicator, the communication stream's ordering and the segmented replay then run exactly as with N ranks -- how a
        one-GPU box exercises the RCCL path (tests/test_gpu_dist.py)."""
        self.group = group
        self.world, self.rank = (1, 0) if single else (world(group), rank(group))
        self.exchanges = (self.world > 1 or bool(force_exchange)) and not single  # does a step talk to the group at all?
        self.overlap = overlap
        self.direct = direct
        self.shard_mean = shard_mean
        self._stream = None
        self._pending = None
        self._recv = None
        self.stats = dict(octree_max_reduces=0, grad_reduces=0, direct_reduces=0)

    # ---------------------------------------------------------------- octree statistics (between K5 and K6)
    def reduce_octree_stats(self, tensors):
        """tensors: [scratch (w_adder | a_adder | mark) i64 [3 n_nodes], visit_cnt i64 [n_nodes]] -> elementwise MAX over ranks."""
        if self.exchanges:
            allreduce_max_(tensors, self.group, force=True)
            self.stats["octree_max_reduces"] += 1
        return tensors

    # ---------------------------------------------------------------- gradients
    def _use_direct(self, t):
        direct = self.direct
        if direct is None:  # RCCL has the all-to-all; gloo's device-tensor support is not relied upon
            direct = dist.get_backend(self.group) == "nccl"
        if not direct or t.dim() != 1 or not t.is_contiguous():
            return False
        return t.numel() >= self.DIRECT_MIN_ELEMS and t.numel() % (8 * self.world) == 0

    def _reduce(self, tensors):
        for t in tensors:
            if self._use_direct(t):
                if self._recv is None or self._recv.numel() != t.numel() or self._recv.dtype != t.dtype or self._recv.device != t.device:
                    self._recv = torch.empty_like(t)
                allreduce_mean_direct(t, self.group, recv=self._recv, shard_mean=self.shard_mean, force=True)
                self.stats["direct_reduces"] += 1
            else:
                allreduce_mean([t], self.group, force=True)
        self.stats["grad_reduces"] += 1

    def start_grads(self, tensors):
        """Begin averaging `tensors` (final as of the current stream) over the ranks; returns at once.  On the device the
        exchange runs on the communication stream, so kernels launched next on the current stream overlap it."""
        if not self.exchanges:
            return
        assert self._pending is None, "start_grads: the previous exchange was not finished"
        if self.overlap and tensors and tensors[0].is_cuda:
            cur = torch.cuda.current_stream()
            if self._stream is None:
                from . import ops

                self._stream = ops.concurrent_stream(tensors[0].device, role="comm")  # (a stream sharing the launch stream's hardware queue would not overlap)
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                self._reduce(tensors)
            for t in tensors:
                t.record_stream(self._stream)
            self._pending = True
        else:
            self._reduce(tensors)
            self._pending = False

    def finish_grads(self, tensors=()):
        """Average `tensors` (the small late gradients: decoder, embedding) now and wait for what start_grads began."""
        if not self.exchanges:
            return
        if tensors:
            self._reduce(list(tensors))
        if self._pending:
            torch.cuda.current_stream().wait_stream(self._stream)
        self._pending = None

    # ---------------------------------------------------------------- checks
    def ranks_seen(self, device=None):
        """Number of ranks the communicator really connects (sum of ones)."""
        if self.world == 1:
            return 1
        one = torch.ones(1, dtype=torch.int64, device=device or "cpu")
        dist.all_reduce(one, group=self.group)
        return int(one.item())

    def replicas_identical(self, tensors):
        """True when every rank holds the same bits in every tensor (MAX and MIN of a byte checksum pair agree)."""
        if self.world == 1:
            return True
        ok = True
        for t in tensors:
            v = t.detach().contiguous().view(torch.uint8).reshape(-1)
            pad = (-v.numel()) % 8
            if pad:
                v = torch.cat([v, v.new_zeros(pad)])
            q = v.view(torch.int64)
            # order-sensitive 2-word checksum (sum and index-weighted sum, wrapping)
            idx = torch.arange(1, q.numel() + 1, dtype=torch.int64, device=q.device)
            cs = torch.stack([q.sum(), (q * idx).sum()])
            hi, lo = cs.clone(), cs.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            ok = ok and bool((hi == lo).all().item())
        return ok

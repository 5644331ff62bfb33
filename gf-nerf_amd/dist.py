"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  Rays are independent units end to end:

  * global stage -- ray data parallelism: every rank holds a replica of tree, global table, decoder and
    embedding and renders its own rays; the per-rank mean-loss gradients are summed and averaged before Adam
    (the reference relies on DDP, which never sees the hash table: hash_3d_anchored.py:26,73-78);
  * focal stage -- one residual block table per rank; everything shared is frozen in this stage
    (nerfacto_field.py:460-461,528,549), so there is NO data-path collective; only stage boundaries talk.

Octree node statistics are max-reduced (the reference accumulates them with atomicMax, order independent).
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_mean(tensors, group=None):
    """In-place average over ranks of a list of gradient tensors (f32 or the f16 x128-scaled table pool)."""
    w = world()
    if w == 1:
        return tensors
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.mul_(1.0 / w)
    return tensors


def allreduce_max_(tensors, group=None):
    """In-place elementwise max over ranks (octree node statistics / visit counts)."""
    if world() == 1:
        return tensors
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return tensors


def block_owner(block, world_size):
    """Rank that owns focal block `block` (one block per GPU, round robin beyond that)."""
    return block % world_size


def my_blocks(n_blocks, world_size=None, my_rank=None):
    w = world() if world_size is None else world_size
    r = rank() if my_rank is None else my_rank
    return [b for b in range(n_blocks) if block_owner(b, w) == r]


def broadcast_from(tensors, src=0, group=None):
    """Stage boundary: publish the (now frozen) global table / decoder from `src` to every rank."""
    if world() == 1:
        return tensors
    for t in tensors:
        dist.broadcast(t, src=src, group=group)
    return tensors


def gather_block_tables(local_tables, n_blocks, dst=0, group=None):
    """Checkpoint: collect every rank's block tables on `dst` -> {block: tensor} (None elsewhere).
    local_tables: {block: tensor} owned by this rank."""
    w, r = world(), rank()
    if w == 1:
        return dict(local_tables)
    out = {} if r == dst else None
    for b in range(n_blocks):
        owner = block_owner(b, w)
        if owner == dst:
            if r == dst:
                out[b] = local_tables[b]
            continue
        if r == owner:
            dist.send(local_tables[b], dst=dst, group=group)
        elif r == dst:
            buf = torch.empty_like(next(iter(local_tables.values())))
            dist.recv(buf, src=owner, group=group)
            out[b] = buf
    return out

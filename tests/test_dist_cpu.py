"""World-size-2 gloo tests (CPU) of the multi-GPU path: gradient averaging of the ray-data-parallel global
stage, max-reduction of octree statistics, one-block-per-rank ownership and the checkpoint gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import dist as gd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        grads = [torch.randn(1000, generator=g), torch.randn(17, 3, generator=g)]
        mine = [t.clone() for t in grads]
        gd.allreduce_mean(grads)
        stats = [torch.tensor([1000, -3, 512 + rank, 7 * rank], dtype=torch.int64)]
        gd.allreduce_max_(stats)
        blocks = gd.my_blocks(5)
        tables = {b: torch.full((4, 2), float(b)) for b in blocks}
        gathered = gd.gather_block_tables(tables, 5, dst=0)
        shared = [torch.full((3,), float(rank))]
        gd.broadcast_from(shared, src=0)
        q.put((rank, [m.numpy() for m in mine], [t.numpy() for t in grads], stats[0].numpy(), blocks,
               None if gathered is None else {b: t.numpy() for b, t in gathered.items()}, shared[0].numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_collectives():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, avg0, st0, b0, g0, sh0), (r1, mine1, avg1, st1, b1, g1, sh1) = res
    for a0, a1, m0, m1 in zip(avg0, avg1, mine0, mine1):
        np.testing.assert_allclose(a0, (m0 + m1) / 2, rtol=1e-6)
        assert np.array_equal(a0, a1)  # replicas stay bit-identical
    assert st0.tolist() == [1000, -3, 513, 7] and st1.tolist() == st0.tolist()
    assert b0 == [0, 2, 4] and b1 == [1, 3]
    assert g1 is None and sorted(g0) == [0, 1, 2, 3, 4]
    for b, t in g0.items():
        assert np.all(t == b)
    assert np.all(sh0 == 0) and np.all(sh1 == 0)


def test_single_process_is_identity():
    sys.path.insert(0, ROOT)
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import dist as gd

    t = [torch.ones(3)]
    assert gd.allreduce_mean(t)[0].tolist() == [1, 1, 1]
    assert gd.my_blocks(3) == [0, 1, 2] and gd.world() == 1 and gd.rank() == 0

"""World-size-2 and world-size-8 gloo tests (CPU) of the multi-GPU path: the exchanges HotPath.train_step drives through
gfnerf_amd.dist.TrainSync (MAX-reduce of the octree statistics between K5 and K6, the direct all-reduce of the table
gradient started inside the backward, the late decoder / embedding gradients), one-block-per-rank ownership, block-owner
frame routing and the checkpoint gather.  The ranks run the product's host code on CPU tensors; the one device kernel of
that path (gfn_reduce_shards, the local shard sum) is replaced by an injected torch restatement, which the GPU test
`test_reduce_shards_matches_torch` holds equal to the kernel."""
import os
import socket
import sys
import types

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


def torch_shard_mean(recv, n_shards, out):
    """What gfn_reduce_shards computes: f32 sum of the copies in rank order, x 1/n, rounded once to the buffer's type."""
    acc = torch.zeros(out.numel(), dtype=torch.float32)
    for s in range(n_shards):
        acc += recv.view(n_shards, -1)[s].float()
    out.copy_((acc * (1.0 / n_shards)).to(out.dtype))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import dist as gd
    from gfnerf_amd import hotpath

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
        # dst owns no block (n_blocks < world): the receive buffer comes from `shape`, not from a local table
        lone = gd.gather_block_tables({0: torch.full((4, 2), 7.0)} if rank == 0 else {}, 1, dst=1, shape=(4, 2))
        shared = [torch.full((3,), float(rank))]
        gd.broadcast_from(shared, src=0)

        # ---- the HotPath-level hooks, on fake tensors
        sync = gd.TrainSync(direct=True, shard_mean=torch_shard_mean)
        n_nodes = 37
        scratch = torch.randint(-1, 600, (3 * n_nodes,), generator=g, dtype=torch.int64)
        visit = torch.randint(0, 50, (n_nodes,), generator=g, dtype=torch.int64)
        oct_mine = (scratch.clone(), visit.clone())
        sync.reduce_octree_stats([scratch, visit])  # what ops.update_nodes calls between K5 and K6
        n_tab = 17 * (1 << 16)  # the live part of a T = 2^16 table: a multiple of 8 * world, above DIRECT_MIN_ELEMS
        table_grad = (torch.randn(n_tab, generator=g) * 40).half()
        table_grad[5] = 60000.0 if rank == 0 else 50000.0  # a plain f16 SUM of these overflows to inf
        mlp_grad, emb_grad = torch.randn(31251, generator=g), torch.randn(40, 32, generator=g)
        g_mine = (table_grad.clone(), mlp_grad.clone(), emb_grad.clone())
        def fake_hp(grads):  # what HotPath.sync_grads touches: the exchange points run eagerly (seg is set only under a segmented capture)
            ns = types.SimpleNamespace(sync=sync, grads=grads, seg=None)
            ns._eager = lambda fn: hotpath.HotPath._eager(ns, fn)
            return ns

        fake = fake_hp(lambda stage: [table_grad, mlp_grad, emb_grad])
        # 1) the fused backward started the table exchange itself (ctx flag), sync_grads sends the late ones and waits
        sync.start_grads([table_grad])
        hotpath.HotPath.sync_grads(fake, hotpath.STAGE_GLOBAL, {"table_exchange_started": True})
        first = (table_grad.clone(), mlp_grad.clone(), emb_grad.clone())
        # 2) nothing started yet (unfused paths): sync_grads does everything; the focal stage exchanges nothing
        t2, m2, e2 = g_mine[0].clone(), g_mine[1].clone(), g_mine[2].clone()
        fake2 = fake_hp(lambda stage: [t2, m2, e2])
        hotpath.HotPath.sync_grads(fake2, hotpath.STAGE_GLOBAL, {})
        t3 = g_mine[0].clone()
        hotpath.HotPath.sync_grads(fake_hp(lambda stage: [t3]), hotpath.STAGE_FOCAL, {})
        # the f16 pool through the plain path (sizes the direct path cannot split, gloo rehearsals): summed in f32, rounded once
        t4 = g_mine[0].clone()
        gd.allreduce_mean([t4])
        check = gd.allreduce_selfcheck(n=1 << 16, device="cpu", shard_mean=torch_shard_mean)
        same = sync.replicas_identical([table_grad, mlp_grad])
        differ = sync.replicas_identical([torch.full((9,), float(rank))])
        q.put(dict(rank=rank, mine=[m.numpy() for m in mine], avg=[t.numpy() for t in grads], stats=stats[0].numpy(), blocks=blocks,
                   gathered=None if gathered is None else {b: t.numpy() for b, t in gathered.items()},
                   lone=None if lone is None else {b: t.numpy() for b, t in lone.items()}, shared=shared[0].numpy(),
                   oct_mine=[t.numpy() for t in oct_mine], oct=[scratch.numpy(), visit.numpy()],
                   g_mine=[t.float().numpy() for t in g_mine], first=[t.float().numpy() for t in first],
                   second=[t.float().numpy() for t in (t2, m2, e2)], focal_untouched=bool(torch.equal(t3, g_mine[0])),
                   plain_f16=t4.float().numpy(), selfcheck=check,
                   same=same, differ=differ, ranks_seen=sync.ranks_seen(), stats_sync=dict(sync.stats)))
    finally:
        dist.destroy_process_group()


def test_two_rank_collectives_and_hotpath_hooks():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res
    for a0, a1, m0, m1 in zip(r0["avg"], r1["avg"], r0["mine"], r1["mine"]):
        np.testing.assert_allclose(a0, (m0 + m1) / 2, rtol=1e-6)
        assert np.array_equal(a0, a1)  # replicas stay bit-identical
    assert r0["stats"].tolist() == [1000, -3, 513, 7] and r1["stats"].tolist() == r0["stats"].tolist()
    assert r0["blocks"] == [0, 2, 4] and r1["blocks"] == [1, 3]
    assert r1["gathered"] is None and sorted(r0["gathered"]) == [0, 1, 2, 3, 4]
    for b, t in r0["gathered"].items():
        assert np.all(t == b)
    assert r0["lone"] is None and list(r1["lone"]) == [0] and np.all(r1["lone"][0] == 7.0)
    assert np.all(r0["shared"] == 0) and np.all(r1["shared"] == 0)
    # octree statistics: elementwise maximum of the two ranks' K5 results, identical on both
    for k in range(2):
        want = np.maximum(r0["oct_mine"][k], r1["oct_mine"][k])
        assert np.array_equal(r0["oct"][k], want) and np.array_equal(r1["oct"][k], want)
    # gradients: the mean of the two ranks' (f32 sum, one rounding), no overflow, identical bits on both ranks, both call orders
    for key in ("first", "second"):
        for k in range(3):
            want = (r0["g_mine"][k] + r1["g_mine"][k]) / 2
            if k == 0:
                want = want.astype(np.float16).astype(np.float32)
            np.testing.assert_allclose(r0[key][k], want, rtol=2e-6 if k else 0, atol=0)
            assert np.array_equal(r0[key][k], r1[key][k])
        assert np.isfinite(r0[key][0]).all() and r0[key][0][5] == np.float32(np.float16(54976.0))  # (59968 + 49984) / 2
    assert np.array_equal(r0["plain_f16"], r0["first"][0]) and np.array_equal(r1["plain_f16"], r0["first"][0])  # same bits as the direct path
    assert r0["selfcheck"]["ok"] and r0["selfcheck"] == r1["selfcheck"] and r0["selfcheck"]["world"] == 2
    assert r0["focal_untouched"] and r1["focal_untouched"]
    assert r0["same"] and r1["same"] and not r0["differ"] and not r1["differ"]
    assert r0["ranks_seen"] == 2 and r1["ranks_seen"] == 2
    assert r0["stats_sync"]["octree_max_reduces"] == 1 and r0["stats_sync"]["direct_reduces"] == 2


def _pattern(n, r, scale=1.0):
    """Rank r's table-gradient pool for the world-8 rehearsal: f16-exact values every rank can rebuild for every other rank (a
    multiplicative hash of the index, shifted per rank), x128-scale magnitudes, a few entries a plain f16 sum would overflow on."""
    i = torch.arange(n, dtype=torch.int64)
    v = (((i * 2654435761 + r * 40503) >> 7) % 4096 - 2048).to(torch.float32) * (scale / 16.0)
    v[5::100003] = 50000.0 + 100.0 * r
    return v.to(torch.float16)


def _worker8(rank, world, port, q, n_tab):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import cluster, dist as gd
    from gfnerf_amd import hotpath

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = dict(rank=rank)
        sync = gd.TrainSync(direct=True, shard_mean=torch_shard_mean)
        # ---- the table gradient at its REAL element count (17 * 2^21 f16 = 71 MB): shard = n / world
        table_grad = _pattern(n_tab, rank)
        assert sync._use_direct(table_grad) and n_tab % (8 * world) == 0
        g = torch.Generator().manual_seed(100 + rank)
        mlp_grad, emb_grad = torch.randn(31251, generator=g), torch.randn(40, 32, generator=g)
        mlp_mine = mlp_grad.clone()

        def fake_hp(grads):
            ns = types.SimpleNamespace(sync=sync, grads=grads, seg=None)
            ns._eager = lambda fn: hotpath.HotPath._eager(ns, fn)
            return ns

        sync.start_grads([table_grad])  # what the fused backward does once the table gradient is final
        hotpath.HotPath.sync_grads(fake_hp(lambda stage: [table_grad, mlp_grad, emb_grad]), hotpath.STAGE_GLOBAL, {"table_exchange_started": True})
        # expected: f32 sum of the ranks' pools in rank order x 1/world, rounded to f16 once (gfn_reduce_shards' contract), chunk by chunk
        ok, finite = True, True
        for c0 in range(0, n_tab, 1 << 22):
            c1 = min(n_tab, c0 + (1 << 22))
            acc = torch.zeros(c1 - c0, dtype=torch.float32)
            i = torch.arange(c0, c1, dtype=torch.int64)
            for r in range(world):
                v = (((i * 2654435761 + r * 40503) >> 7) % 4096 - 2048).to(torch.float32) * (1.0 / 16.0)
                first = (5 - c0) % 100003
                v[first::100003] = 50000.0 + 100.0 * r
                acc += v.to(torch.float16).float()
            want = (acc * (1.0 / world)).to(torch.float16)
            ok = ok and bool(torch.equal(table_grad[c0:c1], want))
            finite = finite and bool(torch.isfinite(table_grad[c0:c1].float()).all())
        res.update(table_mean_exact=ok, table_finite=finite, direct_reduces=sync.stats["direct_reduces"], shard=n_tab // world)
        # the decoder gradient: mean of the eight (every rank can rebuild the others' draws)
        want_mlp = torch.stack([torch.randn(31251, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]).double().mean(0)
        res["mlp_err"] = float((mlp_grad.double() - want_mlp).abs().max())
        res["mlp_changed"] = not torch.equal(mlp_grad, mlp_mine)
        # ---- octree statistics: elementwise MAX of the eight ranks' K5 results
        n_nodes = 10809
        mk = lambda r: (torch.randint(-1, 600, (3 * n_nodes,), generator=torch.Generator().manual_seed(7 + r), dtype=torch.int64),  # noqa: E731
                        torch.randint(0, 50, (n_nodes,), generator=torch.Generator().manual_seed(70 + r), dtype=torch.int64))
        scratch, visit = (t.clone() for t in mk(rank))
        sync.reduce_octree_stats([scratch, visit])
        alls = [mk(r) for r in range(world)]
        res["oct_max_exact"] = bool(torch.equal(scratch, torch.stack([a[0] for a in alls]).max(0).values) and
                                    torch.equal(visit, torch.stack([a[1] for a in alls]).max(0).values))
        # ---- one block per rank: ownership, frame routing, the checkpoint gather, the stage-boundary broadcast
        res["blocks8"], res["blocks16"] = gd.my_blocks(8), gd.my_blocks(16)
        rng = np.random.default_rng(0)
        pos = rng.uniform(-40, 40, (1600, 3))
        labels = cluster.equal_size_clusters(pos, 8)
        path = rng.integers(0, 1600, 40)
        res["owners"] = gd.frame_owner(labels[path], world)
        res["my_frames"] = gd.frame_shard_by_block(labels[path], world, rank)
        res["path_labels"] = labels[path].tolist()
        res["my_cams"] = int(cluster.cameras_of(labels, gd.my_blocks(8)[0]).size)
        tables = {b: torch.full((64, 2), float(b)) for b in gd.my_blocks(8)}
        gathered = gd.gather_block_tables(tables, 8, dst=0)
        res["gathered"] = None if gathered is None else {int(b): float(t.mean()) for b, t in gathered.items()}
        shared = [torch.full((3,), float(rank))]
        gd.broadcast_from(shared, src=3)
        res["broadcast"] = shared[0].tolist()
        res["selfcheck"] = gd.allreduce_selfcheck(n=1 << 18, device="cpu", shard_mean=torch_shard_mean)
        res["same"] = sync.replicas_identical([table_grad, mlp_grad, scratch])
        res["differ"] = sync.replicas_identical([torch.full((9,), float(rank == 5))])  # ONE rank out of eight differs
        res["ranks_seen"] = sync.ranks_seen()
        q.put(res)
    finally:
        dist.destroy_process_group()


def test_eight_rank_rehearsal():
    """the exchanges of the data-parallel step and the one-block-per-rank bookkeeping at the world size of the node
    (BASELINE configs[3]: 8 x MI355X) -- TrainSync's direct path with shard = n / 8 on the table gradient's real element count
    (17 * 2^21 f16), its result bit for bit the f32 mean rounded once; the K5 MAX exchange on a bench-sized tree; my_blocks /
    frame_owner / frame_shard_by_block for 8 blocks and 1,600 cameras; gather_block_tables; replicas_identical telling one deviating
    rank out of eight.  gloo on the CPU: what only RCCL over xGMI can still break is listed in DESIGN.md section 6."""
    world, n_tab = 8, 17 << 21
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q, n_tab)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda x: x["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r, x in enumerate(res):
        assert x["table_mean_exact"] and x["table_finite"] and x["direct_reduces"] == 1 and x["shard"] == n_tab // 8, x
        assert x["mlp_err"] < 1e-6 and x["mlp_changed"] and x["oct_max_exact"], x
        assert x["blocks8"] == [r] and x["blocks16"] == [r, r + 8]
        assert x["owners"] == [lb % 8 for lb in x["path_labels"]]
        assert x["my_frames"] == [f for f, lb in enumerate(x["path_labels"]) if lb % 8 == r]
        assert x["my_cams"] == 200  # 1,600 cameras in 8 equal clusters
        assert x["broadcast"] == [3.0, 3.0, 3.0]
        assert x["selfcheck"]["ok"] and x["selfcheck"]["world"] == 8
        assert x["same"] and not x["differ"] and x["ranks_seen"] == 8
    assert sorted(f for x in res for f in x["my_frames"]) == list(range(40))  # every frame of the path rendered exactly once
    assert res[0]["gathered"] == {b: float(b) for b in range(8)} and all(x["gathered"] is None for x in res[1:])


def test_single_process_is_identity():
    sys.path.insert(0, ROOT)
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import dist as gd

    t = [torch.ones(3)]
    assert gd.allreduce_mean(t)[0].tolist() == [1, 1, 1]
    assert gd.my_blocks(3) == [0, 1, 2] and gd.world() == 1 and gd.rank() == 0
    sync = gd.TrainSync()
    assert sync.world == 1 and sync.ranks_seen() == 1 and sync.replicas_identical(t)
    sync.start_grads(t); sync.finish_grads(t)
    assert sync.reduce_octree_stats(t)[0].tolist() == [1, 1, 1]


def test_frame_routing_follows_block_ownership():
    sys.path.insert(0, ROOT)
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import cluster, dist as gd

    rng = np.random.default_rng(0)
    pos = rng.uniform(-40, 40, (203, 3))
    for k in (2, 4, 8):
        labels = cluster.equal_size_clusters(pos, k)
        sizes = np.bincount(labels, minlength=k)
        assert sorted(sizes.tolist(), reverse=True) == cluster.cluster_sizes(203, k)  # equal up to one, none empty
        assert np.array_equal(labels, cluster.equal_size_clusters(pos, k))           # deterministic
        # compact: a camera is on average closer to its own cluster's centre than to the other centres
        cen = np.stack([pos[labels == j].mean(0) for j in range(k)])
        d = np.linalg.norm(pos[:, None] - cen[None], axis=-1)
        assert (d.argmin(1) == labels).mean() > 0.8
        # rank r's cameras are all in cluster r; a frame goes to the rank owning the block of its camera
        for r in range(k):
            cams = cluster.cameras_of(labels, gd.my_blocks(k, k, r)[0])
            assert np.all(labels[cams] == r) and cams.size == sizes[r]
        path = rng.integers(0, 203, 25)
        owners = gd.frame_owner(labels[path], k)
        assert owners == [int(labels[c]) % k for c in path]
        shards = [gd.frame_shard_by_block(labels[path], k, r) for r in range(k)]
        assert sorted(f for s in shards for f in s) == list(range(25))


def test_focal_blocks_are_routed_one_per_rank():
    """GFNeRFModel.stage_of: one process follows the reference's schedule (block = period mod n_split, perssampler.py:316-317); W ranks
    train W different blocks in every period and cover all blocks in n_split / W periods."""
    from gfnerf_amd.gfnerf.nerfacto import focal_block

    S, P, N = 30000, 10000, 8
    assert focal_block(0, S, P, N) is None and focal_block(S - 1, S, P, N, 8, 3) is None and focal_block(-1, S, P, N) is None
    assert [focal_block(S + p * P, S, P, N) for p in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    for world in (2, 4, 8):
        seen = []
        for period in range(N // world):
            blocks = [focal_block(S + period * P + 17, S, P, N, world, r) for r in range(world)]
            assert len(set(blocks)) == world  # no two ranks on one table
            seen += blocks
        assert sorted(seen) == list(range(N))

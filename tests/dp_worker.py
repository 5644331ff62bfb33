"""One rank of the two-process data-parallel test (tests/test_gpu_dist.py): the global stage of the hot path on ONE card
shared by both ranks (gloo: RCCL refuses two ranks on one device), driven through HotPath.train_step exactly as bench.py
and a trainer do.  Not collected by pytest (no test_ prefix); started as `python dp_worker.py RANK WORLD PORT OUT`.

Checks, all reported in OUT (JSON, written by rank 0):
  * the averaged gradients HotPath.sync_grads leaves behind = the mean of the two ranks' purely local gradients;
  * after the first step's UpdateOctNodes the tree and the node statistics equal what ONE process computes from the
    union of both ranks' rays (the MAX-reduce between K5 and K6);
  * after 3 steps (the last two with the next batch's sampler prefetched on the second stream) tables, decoder,
    embedding, node blob and statistics are bit-identical on both ranks.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    direct = len(sys.argv) > 5 and sys.argv[5] == "direct"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    import torch
    import torch.distributed as dist

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import dist as gd, hotpath, ops, scene

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    try:
        sc = scene.build_scene(n_side=5, max_depth=8, seed=3)
        cfg = hotpath.HotPathConfig(log2_T=16, early_T=1e-4)
        R = 512
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        noise = t(scene.make_noise(R, False, 6.0))  # eval-mode march: independent of a ray's index in its batch

        def rays(r, k):
            o, d, cam = scene.make_rays(sc["c2w"], R, seed=10 * r + k)
            gt = np.random.default_rng(50 + 10 * r + k).random((R, 3)).astype(np.float32)
            return o, d, cam, gt

        def fresh(sync):
            hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, sync=sync)
            hp.table.mul_(50.0)  # visible density from the first step on
            hp.refresh_shadows()
            return hp

        s3 = ops.s3im_index(R, cfg.s3im_repeat_time, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
        mine = [tuple(t(a) for a in rays(rank, k)) for k in range(4)]

        # ---- purely local gradients of step 0 (no exchange), gathered on every rank
        loc = fresh(gd.TrainSync(single=True))
        loc.forward_backward(*mine[0][:4], noise, stage="global", s3im_index=s3)
        local_grads = [g.detach().float().cpu() for g in loc.grads("global")]
        everyone = [None] * world
        dist.all_gather_object(everyone, [g.numpy() for g in local_grads])
        del loc

        # ---- the data-parallel replica
        sync = gd.TrainSync(direct=True if direct else None)
        hp = fresh(sync)
        loss, ctx = hp.forward_backward(*mine[0][:4], noise, stage="global", s3im_index=s3, update_oct=True)
        hp.sync_grads("global", ctx)
        got = [g.detach().float().cpu().numpy() for g in hp.grads("global")]
        errs = []
        for k in range(len(got)):
            want = sum(e[k] for e in everyone) / world
            if k == 0:  # f16 pool: mean in f32, one rounding (gfn_reduce_shards) or pre-scaled f16 sum (all_reduce)
                scale = max(float(np.abs(want).max()), 1e-30)
                errs.append(float(np.abs(got[k] - want).max() / scale))
            else:
                errs.append(float(np.abs(got[k] - want).max() / max(float(np.abs(want).max()), 1e-30)))
        res["grad_rel_err"] = errs
        res["table_grad_nonzero"] = int((got[0] != 0).sum())
        tree_after_1 = [hp.nodes.clone(), hp.weight_stats.clone(), hp.alpha_stats.clone(), hp.visit_cnt.clone()]
        hp.optimizer_step("global")

        # ---- the same first step's octree update from ONE process holding both ranks' rays
        if rank == 0:
            uni = fresh(gd.TrainSync(single=True))
            both = [np.concatenate([rays(r, 0)[i] for r in range(world)]) for i in range(4)]
            comp, uctx = uni.forward_train(t(both[0]), t(both[1]), t(both[2]), t(scene.make_noise(world * R, False, 6.0)), stage="global")
            uni.update_oct_nodes(uctx)
            res["union_tree_equal"] = bool(torch.equal(uni.nodes, tree_after_1[0]) and torch.equal(uni.weight_stats, tree_after_1[1])
                                           and torch.equal(uni.alpha_stats, tree_after_1[2]) and torch.equal(uni.visit_cnt, tree_after_1[3]))
            res["invalidated_nodes"] = int((tree_after_1[1] < 0).sum().item() + (tree_after_1[2] < 0).sum().item())
            res["visited_nodes"] = int((tree_after_1[3] > 0).sum().item())
            del uni

        # ---- two more steps through train_step, the next batch's sampler prefetched behind the octree update
        for k in (1, 2):
            nxt = mine[(k + 1) % 4]
            hp.train_step(*mine[k][:4], noise, stage="global", s3im_index=s3, prefetch=(nxt[0], nxt[1], noise), update_oct=True)
        torch.cuda.synchronize()
        res["replicas_identical"] = sync.replicas_identical([hp.table, hp.table16, hp.mlp_flat, hp.emb, hp.nodes, hp.weight_stats,
                                                             hp.alpha_stats, hp.visit_cnt])
        mixed = hp.table.clone()
        mixed.view(-1)[7] += float(rank)  # and the check does notice a replica that drifted
        res["drift_detected"] = not sync.replicas_identical([mixed])
        res["sync_stats"] = dict(sync.stats)
        res["loss"] = float(loss.item())

        # ---- the same data-parallel step captured in segments (gfnerf_amd.stepgraph: graphs between the three exchange points, the
        # collectives eager) against the eager data-parallel step, from one state, without random draws
        from gfnerf_amd import stepgraph

        cfg0 = hotpath.HotPathConfig(log2_T=16, early_T=1e-4, s3im_loss_mult=0.0)

        def fresh0(sync_):
            h = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg0, seed=0, sync=sync_)
            h.table.mul_(50.0)
            h.refresh_shadows()
            return h

        se, sg_sync = gd.TrainSync(direct=True if direct else None), gd.TrainSync(direct=True if direct else None)
        he, hg = fresh0(se), fresh0(sg_sync)
        K = 3
        for h in (he, hg):  # step 0 eagerly on both (the capture needs the lazily created buffers)
            h.train_step(*mine[0][:4], noise, stage="global", update_oct=True)
        for k in range(1, K + 1):
            nxt = mine[(k + 1) % 4]
            he.train_step(*mine[k % 4][:4], noise, stage="global", prefetch=(nxt[0], nxt[1], noise), update_oct=True)
        g = stepgraph.StepGraph(hg, "global", R, 6.0, update_oct=True, train_noise=False)
        assert g.segmented
        g.prime(mine[1][0], mine[1][1])
        for k in range(1, K + 1):
            nxt = mine[(k + 1) % 4]
            g.step(mine[k % 4][2], mine[k % 4][3], nxt[0], nxt[1])
        torch.cuda.synchronize()
        res["segments"] = ["graph" if isinstance(x, torch.cuda.CUDAGraph) else "eager" for x in g.G[0]]
        res["graph_replicas_identical"] = sg_sync.replicas_identical([hg.table, hg.table16, hg.mlp_flat, hg.emb, hg.nodes, hg.weight_stats,
                                                                     hg.alpha_stats, hg.visit_cnt])
        res["graph_sync_stats"] = dict(sg_sync.stats)
        res["graph_tree_equal_eager"] = bool(torch.equal(hg.nodes, he.nodes) and torch.equal(hg.visit_cnt, he.visit_cnt)
                                            and torch.equal(hg.weight_stats, he.weight_stats) and torch.equal(hg.alpha_stats, he.alpha_stats))
        far = (hg.table - he.table).abs() > 2e-3 * float(he.table.abs().max())
        res["graph_table_far_frac"] = float(far.float().mean())
        res["graph_mlp_rel"] = float((hg.mlp_flat - he.mlp_flat).abs().max() / he.mlp_flat.abs().max())
        res["graph_steps"] = [hg.step, he.step]
        res["ranks_seen"] = sync.ranks_seen()
        res["ok"] = True
    except Exception as e:  # noqa: BLE001 -- reported through OUT, the test prints it
        import traceback

        res["ok"] = False
        res["error"] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
    finally:
        if rank == 0:
            with open(out_path, "w") as f:
                json.dump(res, f)
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    sys.exit(0 if res.get("ok") else 1)


if __name__ == "__main__":
    main()

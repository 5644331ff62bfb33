"""The data-parallel global step on REAL RCCL with the one GPU a test box has: a process group of world size 1 on backend "nccl"
(= RCCL on ROCm), with TrainSync(force_exchange=True) so that none of the world-1 shortcuts is taken -- every collective of the step
(the MAX all-reduce of the octree statistics between K5 and K6, all_to_all_single -> gfn_reduce_shards -> all_gather_into_tensor of
the f16 table gradient on the communication stream, the small all-reduce of the decoder / embedding gradients) goes through the
communicator, eagerly and between the four graph segments of a segmented StepGraph.  With one rank the mean of one is the value
itself, so the run must be the single-process run -- bit for bit in everything integer (tree, node statistics), and as close in the
floating-point state as a second single-process run is to the first (the control: at this table size some gradient-record streams
overflow into f16 atomics, so the plain step does not repeat itself bit for bit).  scripts/train.py:124-137, gf_pipeline.py:136-138
are the reference's counterpart: DDP over NCCL.  Not collected by pytest; started by tests/test_gpu_dist.py as `python rccl_worker.py PORT OUT`."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port, out_path = sys.argv[1], sys.argv[2]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    import torch
    import torch.distributed as dist

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import dist as gd, hotpath, ops, scene, stepgraph

    torch.cuda.set_device(0)
    res = {}
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        res["backend"] = dist.get_backend()
        probe = torch.ones(4, device="cuda")
        dist.all_reduce(probe)
        res["allreduce_of_ones"] = probe.tolist()
        res["selfcheck"] = gd.allreduce_selfcheck(device="cuda", force=True)  # all_to_all_single + gfn_reduce_shards + all_gather_into_tensor
        # the three collectives of the direct all-reduce on a buffer of the table gradient's real size (17 T f16 at T = 2^21 = 71 MB)
        big = (torch.randn(17 * (1 << 21), device="cuda") * 64).to(torch.float16)
        want = big.clone()
        gd.allreduce_mean_direct(big, force=True)
        res["direct_71MB_identity"] = bool(torch.equal(big, want))
        del big, want

        sc = scene.build_scene(n_side=5, max_depth=8, seed=3)
        R = 512
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        noise = t(scene.make_noise(R, False, 6.0))

        def rays(k):
            o, d, cam = scene.make_rays(sc["c2w"], R, seed=k)
            gt = np.random.default_rng(50 + k).random((R, 3)).astype(np.float32)
            return tuple(t(a) for a in (o, d, cam, gt))

        batches = [rays(k) for k in range(4)]

        def fresh(sync, **kw):
            # (the decoder's dW of a batch depends on nothing but the batch; the appearance embedding's gradient is summed with float
            # atomics, whose order is the hardware's: the control run below measures that, the assertion is against the control)
            cfg = hotpath.HotPathConfig(log2_T=16, early_T=1e-4, **kw)
            hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=cfg, seed=0, sync=sync)
            hp.table.mul_(50.0)
            hp.refresh_shadows()
            return hp

        def state(hp):
            return [hp.table, hp.table16, hp.mlp_flat, hp.emb, hp.nodes, hp.weight_stats, hp.alpha_stats, hp.visit_cnt]

        def same(a, b):
            return [bool(torch.equal(x, y)) for x, y in zip(state(a), state(b))]

        def close(a, b):
            return [float((x.float() - y.float()).abs().max() / max(float(y.float().abs().max()), 1e-30)) for x, y in zip(state(a), state(b))]

        s3 = ops.s3im_index(R, 10, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))

        # ---- the eager data-parallel step: one process (shortcuts, no group) | the same again (control) | every exchange through RCCL
        def eager_run(sync, **kw):
            hp = fresh(sync, **kw)
            first = None
            for k in range(3):
                nxt = batches[(k + 1) % 4]
                hp.train_step(*batches[k], noise, stage="global", s3im_index=s3, prefetch=(nxt[0], nxt[1], noise) if k else None, update_oct=True)
                if k == 0:
                    first = [x.clone() for x in state(hp)]
            torch.cuda.synchronize()
            return hp, first

        def far(a, b):  # fraction of table entries further apart than 2e-3 of the largest (a last-bit difference of a near-zero gradient is a
            return float(((a.table - b.table).abs() > 2e-3 * float(b.table.abs().max())).float().mean())  # full Adam step at eps = 1e-15)

        (plain, plain1), (control, control1) = eager_run(gd.TrainSync(single=True)), eager_run(gd.TrainSync(single=True))
        forced_sync = gd.TrainSync(force_exchange=True)
        forced, forced1 = eager_run(forced_sync)
        res["eager_control_identical_step1"] = [bool(torch.equal(x, y)) for x, y in zip(plain1, control1)]
        res["eager_forced_identical_step1"] = [bool(torch.equal(x, y)) for x, y in zip(plain1, forced1)]
        res["eager_control_identical"] = same(plain, control)
        res["eager_forced_identical"] = same(plain, forced)
        res["eager_forced_rel"] = close(plain, forced)
        # decoder / embedding entries further apart than 2e-4 of the largest: a flipped Adam step moves ONE weight, an exchange that lost or mis-scaled a gradient most
        far_of = lambda a, b: [float(((x.float() - y.float()).abs() > 2e-4 * float(y.float().abs().max())).float().mean()) for x, y in ((a.mlp_flat, b.mlp_flat), (a.emb, b.emb))]  # noqa: E731
        res["eager_forced_far"], res["eager_control_far"] = far_of(plain, forced), far_of(plain, control)
        res["eager_control_rel"] = close(plain, control)
        res["eager_forced_table_far_frac"], res["eager_control_table_far_frac"] = far(plain, forced), far(plain, control)
        # without the appearance embedding (its gradient is the step's only float-atomic sum): is the step then repeatable bit for bit?
        (p2, _), (c2, _) = eager_run(gd.TrainSync(single=True), use_appearance_embedding=False), eager_run(gd.TrainSync(single=True), use_appearance_embedding=False)
        f2, _ = eager_run(gd.TrainSync(force_exchange=True), use_appearance_embedding=False)
        res["noemb_control_identical"], res["noemb_forced_identical"] = same(p2, c2), same(p2, f2)
        del p2, c2, f2
        res["eager_sync_stats"] = dict(forced_sync.stats)
        res["uses_direct"] = bool(forced_sync._use_direct(forced._grad_live()))
        res["comm_stream_used"] = forced_sync._stream is not None
        del control, forced

        # ---- the segmented replay: four graph segments, the three exchanges eager between them, on RCCL
        def graph_run(sync, segmented):
            hp = fresh(sync, s3im_loss_mult=0.0)
            hp.train_step(*batches[0], noise, stage="global", update_oct=True)  # (lazily created buffers exist before the capture)
            g = stepgraph.StepGraph(hp, "global", R, 6.0, update_oct=True, train_noise=False, segmented=segmented)
            g.prime(batches[1][0], batches[1][1])
            for k in range(1, 5):
                nxt = batches[(k + 1) % 4]
                g.step(batches[k % 4][2], batches[k % 4][3], nxt[0], nxt[1])
            torch.cuda.synchronize()
            return hp, g

        gsync = gd.TrainSync(force_exchange=True)
        hg, g = graph_run(gsync, None)  # segmented because the sync exchanges
        hs, g1 = graph_run(gd.TrainSync(single=True), True)   # the same segments with identity exchanges, no communicator
        h1, g2 = graph_run(gd.TrainSync(single=True), False)  # the one-graph step
        res["segments"] = ["graph" if isinstance(x, torch.cuda.CUDAGraph) else "eager" for x in g.G[0]]
        res["graph_forced_vs_segmented_identical"] = same(hs, hg)
        res["graph_forced_vs_segmented_rel"] = close(hs, hg)
        res["graph_forced_vs_onegraph_rel"] = close(h1, hg)
        res["graph_sync_stats"] = dict(gsync.stats)
        res["graph_steps"] = [hg.step, hs.step, h1.step]
        res["finite"] = bool(all(torch.isfinite(x.float()).all() for x in state(hg)[:4]))
        res["ok"] = True
    except Exception as e:  # noqa: BLE001 -- reported through OUT, the test prints it
        import traceback

        res["ok"] = False
        res["error"] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
    finally:
        with open(out_path, "w") as f:
            json.dump(res, f)
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    sys.exit(0 if res.get("ok") else 1)


if __name__ == "__main__":
    main()

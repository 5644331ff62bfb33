"""The plugin-surface focal iteration on the plugin-TRAINED scene (bench.plugin_step_bench(trained=(3000, 1000))) under its switches: graph replay of the
forward or the eager forward, the backward's run merge of records on or off, no prefetch (tools/README.md, round 5)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gfnerf_amd  # noqa: E402,F401
from gfnerf_amd import scene  # noqa: E402

sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir="/tmp/gfnerf_amd_scene")
from gfnerf_amd import _lib  # noqa: E402
from gfnerf_amd import hotpath  # noqa: E402
VARIANTS = dict(base=[("graph", {}, 0), ("graph+merge", {}, 1), ("eager forward", dict(graph_forward=False), 0), ("eager forward+merge", dict(graph_forward=False), 1),
                      ("graph, no prefetch", dict(prefetch=False), 0)],
                cus=[(f"graph, backward on {n or 256} CUs beside the sampler", dict(hot_cfg=dict(prefetch_cus=n, prefetch_cus_dense_eager=n or 256)), 0) for n in (0, 128, 160, 192, 224)],
                policy=[("the engine's policy (dense batch: every CU, records merged)", {}, None), ("merge fixed off", {}, 0), ("merge fixed on", {}, 1)],
                # round 6: the backward as a second replayed graph, the fused loss node, the S3IM index drawn ahead
                r06=[("forward + backward replayed", {}, None), ("backward eager", dict(hot_cfg=dict(graph_backward=False)), None),
                     ("forward + backward eager", dict(graph_forward=False), None)],
                r06ev=[("forward + backward replayed (GFN_PLUGIN_EVENTS=1 prints where the iteration goes)", {}, None)],
                r06cus=[(f"replayed, backward on {n} CUs", dict(hot_cfg=dict(prefetch_cus=n, prefetch_cus_dense_replay=n)), None) for n in (256, 224, 192, 160)],
                r06cus2=[(f"replayed, backward on {n} CUs", dict(hot_cfg=dict(prefetch_cus_dense_replay=n)), None) for n in (184, 200, 208, 216)],
                r06one=[("one-phase capture off", dict(hot_cfg=dict(graph_one_phase=False)), None), ("one-phase capture on (default)", {}, None)])
for name, kw, merge in VARIANTS[sys.argv[1] if len(sys.argv) > 1 else "base"]:
    if merge is not None:  # (None: the engine decides per launch on the device, HotPath._merge_gate)
        kw = dict(kw, hot_cfg=dict(kw.get("hot_cfg", {}), record_merge=int(merge)))
    r = bench.plugin_step_bench(sc, torch, "focal", steps=200, warmup=30, trained=(3000, 1000), **kw)
    print(name, round(r["ms_per_step"], 4), "host", round(r["host_ms_per_step"], 4), "forward", r["forward"], "samples/ray", round(r["mean_samples_per_ray"], 1),
          {k: v for k, v in r["trained"].items() if "record" in k or "stream" in k or "loss" in k or "grad" in k}, flush=True)
    torch.cuda.empty_cache()

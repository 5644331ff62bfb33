"""Round 6 diagnosis: the plugin leg's FOCAL stage gains less PSNR than the reference leg's in tools/psnr_study.py (0.31 against 0.66 dB) although its
global stage does not lose.  Which part of the plugin's focal iteration is it?  The same seeds through the plugin leg with parts of its launch policy
switched off (HotPathConfig fields set after the pipeline's setup) and through the native product leg ("fast").
    python tools/exp_plugin_focal_gain.py [seeds, comma separated] [variants, comma separated]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd  # noqa: F401
from gfnerf_amd import procedural, scene

VARIANTS = {
    "plugin": dict(procedural.LEGS["plugin"]),
    "no_graphs": dict(procedural.LEGS["plugin"], graph_forward=False),
    "no_graph_backward": dict(procedural.LEGS["plugin"], graph_backward=False),
    "no_one_phase": dict(procedural.LEGS["plugin"], graph_one_phase=False),
    "no_merge": dict(procedural.LEGS["plugin"], record_merge=0),
    "fast": dict(procedural.LEGS["fast"]),
    "reference": dict(procedural.LEGS["reference"]),
}
# timing probes (a loss that depends on WHEN things run is a race): the same plugin leg with a synchronisation after every optimiser step / before every
# backward, or with the "second stream" being the main stream (no concurrency at all)
for _n in ("sync_iter", "sync_before_backward", "one_stream", "plugin_again"):
    VARIANTS[_n] = dict(procedural.LEGS["plugin"])


def probe(name):
    from gfnerf_amd import hotpath
    from gfnerf_amd.gfnerf import optim
    undo = []
    if name == "sync_iter":
        orig = optim.GfnAdam.step
        optim.GfnAdam.step = lambda self, closure=None: (orig(self, closure), torch.cuda.synchronize())[0]
        undo.append(lambda: setattr(optim.GfnAdam, "step", orig))
    if name == "sync_before_backward":
        orig_b = hotpath.HotPath.backward_graphed
        hotpath.HotPath.backward_graphed = lambda self, ctx, g, g_acc=None: (torch.cuda.synchronize(), orig_b(self, ctx, g, g_acc))[1]
        undo.append(lambda: setattr(hotpath.HotPath, "backward_graphed", orig_b))
    if name == "one_stream":
        orig_s = hotpath.HotPath._new_side_stream
        hotpath.HotPath._new_side_stream = lambda self: torch.cuda.current_stream()
        undo.append(lambda: setattr(hotpath.HotPath, "_new_side_stream", orig_s))
    return undo


seeds = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "36,41").split(",")]
names = (sys.argv[2] if len(sys.argv) > 2 else "plugin,no_graphs,no_graph_backward,no_one_phase,no_merge,fast").split(",")
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
for seed in seeds:
    for name in names:
        t0 = time.perf_counter()
        undo = probe(name)
        r = procedural.train_and_eval(sc, seed=seed, steps_global=procedural.STUDY_STEPS[0], steps_focal=procedural.STUDY_STEPS[1], detail=procedural.STUDY_DETAIL,
                                      lr_final=procedural.STUDY_LR_FINAL, holdout_every=8, **VARIANTS[name])
        print(f"seed {seed} {name:18s}: global {r['psnr_global']:.3f} final {r['psnr']:.3f} gain {r['psnr'] - r['psnr_global']:+.3f}  "
              f"{r.get('plugin', '')}  ({time.perf_counter() - t0:.0f} s)", flush=True)
        for u in undo:
            u()
        del r
        torch.cuda.empty_cache()

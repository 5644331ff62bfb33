"""Round 6 diagnosis: is the REPLAYED backward of the plugin surface's focal iteration the eager backward?  Inside a real training run through the surface
(procedural.train_and_eval, leg "plugin"; the next batch's draw + sampler running on the second stream as always), every replayed backward is followed by
a synchronisation, a copy of the table-gradient pool, and the EAGER backward of the same context from the same d loss / d rgb into a zeroed pool: the two
pools must be the same bits.  GFN_CHECK_SYNC_BEFORE=1 also synchronises BEFORE the replay (a difference that goes away with it is a race with the second
stream).    python tools/exp_plugin_replay_check.py [seed [steps_global steps_focal]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd  # noqa: F401
from gfnerf_amd import hotpath, ops, procedural, scene

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 36
sg = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
sf = int(sys.argv[3]) if len(sys.argv) > 3 else 300
sync_before = os.environ.get("GFN_CHECK_SYNC_BEFORE") == "1"
stats = dict(replayed=0, differ=0, worst=0.0, first=None, forms={})
orig = hotpath.HotPath.backward_graphed


def checked(self, ctx, g_rgb, g_acc=None):
    if sync_before:
        torch.cuda.synchronize()
    orig(self, ctx, g_rgb, g_acc)
    if not self.last_backward_replayed:
        return
    torch.cuda.synchronize()
    live = self._grad_live()
    g1 = live.clone()
    ops.fill_zero(live)
    self.backward(ctx, self.grad_in_buffer(ctx))
    torch.cuda.synchronize()
    g2 = live.clone()
    stats["replayed"] += 1
    form = self.last_forward
    stats["forms"][form] = stats["forms"].get(form, 0) + 1
    if not torch.equal(g1.view(torch.int16), g2.view(torch.int16)):
        stats["differ"] += 1
        d = (g1.float() - g2.float()).abs()
        nz1, nz2 = int((g1 != 0).sum()), int((g2 != 0).sum())
        rel = float(d.sum() / g2.float().abs().sum().clamp(min=1e-30))
        stats["worst"] = max(stats["worst"], rel)
        if stats["first"] is None or stats["differ"] % 50 == 0:
            stats["first"] = stats["first"] or stats["replayed"]
            print(f"  replay {stats['replayed']} ({form}): pools differ: nonzero {nz1} vs {nz2} (eager), sum|diff| / sum|eager| {rel:.3e}, max|diff| {float(d.max()):.3e}, "
                  f"elements differing {int((d > 0).sum())}", flush=True)
    live.copy_(g1)


hotpath.HotPath.backward_graphed = checked
sc = scene.build_scene(n_side=40, max_depth=16, seed=1234, cache_dir=os.path.join(os.environ.get("TMPDIR", "/tmp"), "gfnerf_amd_scene"))
kw = dict(procedural.LEGS["plugin"])
for a in sys.argv[4:]:
    k, v = a.split("=")
    kw[k] = type(getattr(hotpath.HotPathConfig(), k))(eval(v))
r = procedural.train_and_eval(sc, seed=seed, steps_global=sg, steps_focal=sf, detail=procedural.STUDY_DETAIL, lr_final=procedural.STUDY_LR_FINAL, holdout_every=8, **kw)
print(f"seed {seed}: global {r['psnr_global']:.3f} final {r['psnr']:.3f}; replayed backwards {stats['replayed']} ({stats['forms']}), differing from the eager backward "
      f"{stats['differ']} (first at {stats['first']}), worst relative difference {stats['worst']:.3e}; sync before the replay: {sync_before}", flush=True)

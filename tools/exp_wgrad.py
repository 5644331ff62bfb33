"""Scratch: time gfn_wgrad_x3 per layer shape at the global-stage size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import gfnerf_amd
from gfnerf_amd import ops
n = 524288
dev = "cuda"
fp = (torch.randn(n * 480, device=dev).half(), torch.randn(n * 480, device=dev).half())
bp = (torch.randn(n * 408, device=dev).half(), torch.randn(n * 408, device=dev).half())
fv = lambda k: (ops.plane_flat(fp[0], 480, k, ops.FWD_PLANE)[0], *ops.plane_flat(fp[1], 480, k, ops.FWD_PLANE))
bv = lambda k: (ops.plane_flat(bp[0], 408, k, ops.BWD_PLANE)[0], *ops.plane_flat(bp[1], 408, k, ops.BWD_PLANE))
gmax = torch.tensor([0.2], device=dev)
F, B = ops.FWD_PLANE, ops.BWD_PLANE
tot = 0
for name, dzk, ak, mv, shape in (("hW2", "dz3", "c2", 3, (3, 128)), ("hW1", "dz2", "c1", 128, (128, 128)), ("hW0", "dz1", "xp", 128, (128, 64)),
                                 ("bW1", "dh", "a1", 16, (16, 128)), ("bW0", "dza1", "feat", 128, (128, 32))):
    dW = torch.zeros(shape, device=dev); db = torch.zeros(shape[0], device=dev)
    fn = lambda: ops.wgrad_x3(bv(dzk), fv(ak), dW, db=db, gmax=gmax, m_valid=mv, tiled=True)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    gb = n * (B[dzk][1] + F[ak][1]) * 4 / 1e9
    t = float(np.median(ts)); tot += t
    print(f"{name}: {t:.3f} ms  {gb:.2f} GB -> {gb / t * 1e3:.0f} GB/s")
print("total", tot)

# all five layers in one launch (what the global-stage step runs), with partial vectors + reduce, and with atomics
grad = torch.zeros(40000, device=dev)
off = 0
layers = []
for name, dzk, ak, mv, shape in (("hW2", "dz3", "c2", 3, (3, 128)), ("hW1", "dz2", "c1", 128, (128, 128)), ("hW0", "dz1", "xp", 128, (128, 64)),
                                 ("bW1", "dh", "a1", 16, (16, 128)), ("bW0", "dza1", "feat", 128, (128, 32))):
    dW = grad[off:off + shape[0] * shape[1]].view(shape); off += shape[0] * shape[1]
    db = grad[off:off + shape[0]]; off += shape[0]
    layers.append((bv(dzk), fv(ak), dW, db, mv, None))
parts = ops.wgrad_parts(grad.numel(), dev)
for label, p in (("layers + partial vectors + reduce", parts), ("layers + atomics", None)):
    fn = lambda: ops.wgrad_x3_layers(layers, grad, n, gmax=gmax, parts=p)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = float(np.median(ts))
    print(f"{label}: {t:.3f} ms  {n * 888 * 4 / 1e9 / t * 1e3:.0f} GB/s")

"""Hash3DAnchored (gfnerf/hash_3d_anchored.py:17-140) on the HIP kernels.

`_NativeHash3DAnchored` mirrors the TorchScript class my_classes.Hash3DAnchored (bindings.cpp:300-357: ctor
(log2_table_size, n_volumes, lr), AnchoredQuery, GetParams, States, LoadStates, Reset, Zero,
SetFeatPoolRequireGrad, to, ReleaseResources) and Hash3DAnchored::Hash3DAnchored (Hash3DAnchored.cpp:17-75);
`Hash3DAnchored` is the nn.Module wrapper with the same hooks and state-dict keys (feat_pool, prime_pool,
bias_pool, n_volumes).  Saved tensors live in the autograd context (the reference stashes them on `this`,
Hash3DAnchored.cpp:104-106, so two queries before one backward corrupt gradients)."""
from typing import Any, Iterator, List, Mapping

import numpy as np
import torch
from torch import nn

from .. import ops

N_LEVELS, N_CHANNELS = 16, 2


def _random_primes(count, rng):
    """Hash3DAnchored.cpp:32-50: rejection-sample primes in [2^28, 2^30)."""
    out = []
    small = [p for p in range(2, 1 << 15) if all(p % q for q in range(2, int(p ** 0.5) + 1))]
    while len(out) < count:
        cand = rng.integers(1 << 28, 1 << 30, size=max(64, 24 * (count - len(out))))
        cand = cand[cand % 2 == 1]
        ok = np.ones(cand.shape, bool)
        for p in small:
            ok &= cand % p != 0
        out.extend(cand[ok].tolist())
    return np.asarray(out[:count], np.int32)


class _AnchoredQueryFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat_pool, native, points, anchors):
        out = ops.hash_fwd(feat_pool, native.prim_pool_, native.bias_pool_, points, anchors, native.n_volumes_,
                           native.log2_table_size_)
        ctx.native = native
        ctx.save_for_backward(points, anchors)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        points, anchors = ctx.saved_tensors
        nat = ctx.native
        # Hash3DAnchoredFunction::backward (Hash3DAnchored_cuda.cu:198-239): dense grad of the pool, none for the points
        grad = ops.hash_bwd(nat.prim_pool_, nat.bias_pool_, points, anchors, grad_out.contiguous(), nat.n_volumes_,
                            nat.log2_table_size_)
        return grad, None, None, None


class _NativeHash3DAnchored:
    def __init__(self, log2_table_size: int, n_volumes: int, lr: float = 1e-1, device="cuda", seed=None, allocate=True):
        self.log2_table_size_ = int(log2_table_size)
        self.n_volumes_ = int(n_volumes)
        self.learn_rate_ = lr
        self.pool_size_ = (1 << self.log2_table_size_) * N_LEVELS
        dev = torch.device(device)
        if not allocate:  # the caller aliases existing storage (an engine-backed field): nothing to draw or allocate
            self.feat_pool_ = self.prim_pool_ = self.bias_pool_ = torch.empty(0, device=dev)
            return
        rng = np.random.default_rng(seed)
        # Hash3DAnchored.cpp:26: (rand * .2 - 1) * 1e-4, overwritten by Reset()/Zero() by every caller
        self.feat_pool_ = ((torch.rand((self.pool_size_, N_CHANNELS), device=dev) * 0.2 - 1.0) * 1e-4).requires_grad_(True)
        self.prim_pool_ = torch.from_numpy(_random_primes(3 * N_LEVELS * self.n_volumes_, rng).reshape(N_LEVELS, self.n_volumes_, 3)).to(dev)
        self.bias_pool_ = torch.zeros((N_LEVELS * self.n_volumes_, 3), dtype=torch.float32, device=dev)  # rand_bias off

    def AnchoredQuery(self, points: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
        return _AnchoredQueryFn.apply(self.feat_pool_, self, points.contiguous(), anchors.contiguous())

    def GetParams(self) -> List[torch.Tensor]:
        return [self.feat_pool_]

    def States(self) -> List[torch.Tensor]:
        return [self.feat_pool_.data, self.prim_pool_.data, self.bias_pool_.data,
                torch.full((1,), self.n_volumes_, dtype=torch.int32)]

    def LoadStates(self, states: List[torch.Tensor], idx: int) -> int:
        dev = self.feat_pool_.device
        self.feat_pool_.data.copy_(states[idx]); idx += 1
        self.prim_pool_ = states[idx].clone().to(dev).contiguous(); idx += 1   # "The size may changed."
        self.bias_pool_ = states[idx].clone().to(dev).float().contiguous(); idx += 1
        self.n_volumes_ = int(states[idx].item()); idx += 1
        return idx

    def Reset(self):
        self.feat_pool_.data.uniform_(-1e-2, 1e-2)

    def Zero(self):
        self.feat_pool_.data.zero_()

    def SetFeatPoolRequireGrad(self, require_grad: bool):
        self.feat_pool_.requires_grad_(require_grad)

    def to(self, device: str):
        # Hash3DAnchored::to (Hash3DAnchored.cpp:184-200): moves only feat_pool
        rg = self.feat_pool_.requires_grad
        self.feat_pool_ = self.feat_pool_.detach().to("cpu" if device == "cpu" else self.prim_pool_.device).requires_grad_(rg)

    def ReleaseResources(self):
        self.feat_pool_ = torch.empty(0)
        self.prim_pool_ = torch.empty(0)
        self.bias_pool_ = torch.empty(0)
        torch.cuda.empty_cache()


class Hash3DAnchored(nn.Module):
    def __init__(self, log2_table_size: int, n_volumes: int, device="cuda", seed=None, allocate=True) -> None:
        super().__init__()
        self.hash_3d = _NativeHash3DAnchored(log2_table_size, n_volumes, 1e-1, device=device, seed=seed, allocate=allocate)
        self.register_parameter("feat_pool", None)
        self.register_buffer("prime_pool", None)
        self.register_buffer("bias_pool", None)
        self.register_buffer("n_volumes", None)
        self.hook_handles = [self._register_state_dict_hook(self.state_dict_hook)]

    def unregister_hooks(self):
        for h in self.hook_handles:
            h.remove()

    def __del__(self):
        """hash_3d_anchored.py:34-36: the reference announces the destruction (its tables die with the C++ object); here the
        tensors are ordinary torch storage and go with the last reference."""

    def state_dict_hook(self, *args):
        destination, prefix = args[1], args[2]
        feat_pool_, prime_pool_, bias_pool_, n_volume_ = self.states()
        destination[prefix + "feat_pool"] = feat_pool_
        destination[prefix + "prime_pool"] = prime_pool_
        destination[prefix + "bias_pool"] = bias_pool_
        destination[prefix + "n_volumes"] = n_volume_
        return destination

    def forward(self, input):
        points, anchors = input
        assert len(anchors.shape) == 1
        assert points.shape[1] == 3
        assert len(points.shape) == 2
        assert anchors.dtype == torch.int64
        return self.anchored_query(points, anchors)

    def parameters(self, recurse: bool = True) -> Iterator[nn.Parameter]:
        for p in self.hash_3d.GetParams():
            yield p

    def anchored_query(self, points, anchors):
        return self.hash_3d.AnchoredQuery(points, anchors)

    def load_state_dict(self, state_dict: Mapping[str, Any], strict: bool = True, prefix=""):
        pre = "" if prefix == "" else prefix + "."
        keys = [pre + k for k in ("feat_pool", "prime_pool", "bias_pool", "n_volumes")]
        states = [state_dict[k] for k in keys]
        for k in keys:
            del state_dict[k]
        self.load_states(states, idx=0)
        return None

    def set_require_grad(self, require_grad):
        assert type(require_grad) == bool
        self.hash_3d.SetFeatPoolRequireGrad(require_grad)

    def to(self, device):
        assert type(device) == str
        self.hash_3d.to(device)

    def load_states(self, states, idx: int) -> int:
        return self.hash_3d.LoadStates(states, idx)

    def states(self):
        return self.hash_3d.States()

    def get_params(self):
        return self.hash_3d.GetParams()

    def reset(self) -> None:
        self.hash_3d.Reset()

    def zero(self) -> None:
        self.hash_3d.Zero()

    def release_resources(self) -> None:
        self.hash_3d.ReleaseResources()

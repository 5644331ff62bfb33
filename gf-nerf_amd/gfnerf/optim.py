"""GfnAdam: torch.optim.Adam's update (gfnerf/config.py:130 `AdamOptimizerConfig(lr=1e-2, eps=1e-15)`; engine/optimizers.py:38-57)
as a torch.optim.Optimizer whose step runs on the HIP Adam kernels (gfn_adam_step / gfn_adam_step_ex, golden-pinned to
torch.optim.Adam in tests/test_gpu_parity.py::test_charbonnier_and_adam).

It is what the `gf-nerf` method config hands the reference's `Optimizers` (engine/optimizers.py:73-193) as
`AdamOptimizerConfig._target`, so the Trainer's `optimizers.optimizer_scaler_step_all(...)` / `zero_grad_all()` /
`scheduler_step_all()` work unchanged (param_groups, `lr` written by the LambdaLR scheduler, state_dict with `step` /
`exp_avg` / `exp_avg_sq` per parameter like torch.optim.Adam's).

For parameters that alias the engine's storage (the hash tables, the decoder weights, the appearance embedding: tagged
`_gfn_link` by GFNeRFField) the gradient is read where the HIP backward left it -- the f16 x128 table-gradient pool of
Hash3DAnchoredFunction::backward (Hash3DAnchored_cuda.cu:198-239), the flat decoder / embedding gradient vectors -- instead of
from a dense f32 `.grad` (268 MB per table at T = 2^21, which autograd would have to materialise every step), Adam touches
only the 17 T addressable elements of a table (the others never receive a gradient: dense Adam leaves them as they are),
and the kernel that steps a table also refreshes the f16 shadow the lookups read.  A parameter that does carry a `.grad`
(the piecewise Field API, any other module) is stepped from it by the same kernels."""
from dataclasses import dataclass, field
from typing import Type

import torch

from .. import ops
from ._host import AdamOptimizerConfig


class GfnAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("GfnAdam is plain Adam (the gf-nerf config uses weight_decay 0, no amsgrad: gfnerf/config.py:130)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._ids = {id(p) for g in self.param_groups for p in g["params"]}
        self._flat = {}  # engine id -> (m_flat, v_flat) of the decoder's flat parameter vector
        for g in self.param_groups:
            for p in g["params"]:
                link = getattr(p, "_gfn_link", None)
                if link is not None:
                    link[1].fused_optimizers.add(self)  # the autograd bridge stops materialising dense gradients for what this steps

    def owns(self, p):
        return id(p) in self._ids

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_ids"):
            self._ids |= {id(p) for p in self.param_groups[-1]["params"]}

    # ------------------------------------------------------------------------------------------ state
    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)  # like torch.optim.Adam (capturable=False): a host scalar
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _decoder_flat_state(self, hot, decoder):
        """The decoder's ten tensors are views of ONE flat vector (engine.mlp_flat): their moments are kept as views of two flat
        vectors laid out the same way, so the group steps in one launch.  States that came from load_state_dict (fresh
        tensors) are copied back into the flat layout."""
        key = id(hot)
        if key not in self._flat:
            self._flat[key] = (torch.zeros_like(hot.mlp_flat), torch.zeros_like(hot.mlp_flat))
        m_flat, v_flat = self._flat[key]
        base = hot.mlp_flat.data_ptr()
        for p in decoder:
            off = (p.data_ptr() - base) // 4
            st = self.state[p]
            want_m, want_v = m_flat[off:off + p.numel()].view(p.shape), v_flat[off:off + p.numel()].view(p.shape)
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
            for k, want in (("exp_avg", want_m), ("exp_avg_sq", want_v)):
                have = st.get(k)
                if have is None or have.data_ptr() != want.data_ptr():
                    if have is not None:
                        want.copy_(have)
                    st[k] = want
        return m_flat, v_flat

    # ------------------------------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._stepped = set()
        for group in self.param_groups:
            lr, (b1, b2), eps = float(group["lr"]), group["betas"], float(group["eps"])
            decoder = {}
            for p in group["params"]:
                link = getattr(p, "_gfn_link", None)
                if link is not None and link[0] == "mlp":
                    decoder.setdefault(id(link[1]), (link[1], []))[1].append(p)
                    continue
                self._step_one(p, link, lr, b1, b2, eps)
            for hot, ps in decoder.values():
                self._step_decoder(hot, ps, lr, b1, b2, eps)
        for hot in self._stepped:  # this iteration's launches are all enqueued: room for the host work the pipeline put off (the next batch)
            hot.run_deferred()
        return loss

    def _step_one(self, p, link, lr, b1, b2, eps):
        hot, name = (link[1], link[0]) if link is not None else (None, None)
        ready = hot is not None and name in hot.grad_ready
        if not ready and p.grad is None:
            return  # torch.optim.Adam skips parameters without a gradient
        st = self._state_of(p)
        st["step"] += 1
        t = int(st["step"])
        if hot is not None:
            hot.external_adam(name, st["exp_avg"], st["exp_avg_sq"], t, lr, b1, b2, eps, grad=None if ready else p.grad.contiguous())
            hot.grad_ready.discard(name)
            self._stepped.add(hot)
        else:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise NotImplementedError("GfnAdam steps contiguous float32 device parameters")
            ops.adam_step(p, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], t, lr=lr, beta1=b1, beta2=b2, eps=eps)

    def _step_decoder(self, hot, ps, lr, b1, b2, eps):
        ready = "mlp" in hot.grad_ready
        whole = len(ps) == 2 * len(hot.layers)
        if not ready and all(p.grad is None for p in ps):
            return
        if whole and ready:
            m_flat, v_flat = self._decoder_flat_state(hot, ps)
            for p in ps:
                self.state[p]["step"] += 1
            hot.external_adam("mlp", m_flat, v_flat, int(self.state[ps[0]]["step"]), lr, b1, b2, eps)
            hot.grad_ready.discard("mlp")
            self._stepped.add(hot)
            return
        for p in ps:  # a subset of the decoder, or dense .grad tensors: tensor by tensor
            g = hot.layer_grad_of(p) if ready else p.grad
            if g is None:
                continue
            st = self._state_of(p)
            st["step"] += 1
            ops.adam_step(p, g.contiguous(), st["exp_avg"], st["exp_avg_sq"], int(st["step"]), lr=lr, beta1=b1, beta2=b2, eps=eps)
        hot.decoder_changed()
        if ready:
            hot.grad_ready.discard("mlp")

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none)
        for g in self.param_groups:
            for p in g["params"]:
                link = getattr(p, "_gfn_link", None)
                if link is not None:
                    link[1].grad_ready.discard(link[0])


@dataclass
class GfnAdamOptimizerConfig(AdamOptimizerConfig):
    """AdamOptimizerConfig (engine/optimizers.py:38-57) whose `_target` is GfnAdam: same fields (lr, eps, max_norm,
    weight_decay), same `setup(params)`.

    Limits, refused rather than silently ignored: for engine-linked parameters the gradient stays in the engine's buffers and
    `p.grad` is None, so `Optimizers.optimizer_scaler_step_all`'s `clip_grad_norm_` (engine/optimizers.py:131-136) would clip nothing:
    `max_norm` must stay None (the gf-nerf method config's value, gfnerf/config.py:130).  For the same reason gradient accumulation over
    several forwards is not supported (the engine zeroes its buffers in every training forward: `gradient_accumulation_steps` must be 1) and
    neither is an enabled GradScaler (`mixed_precision` off: the HIP backward carries its own x128 / power-of-two loss scales)."""
    _target: Type = field(default_factory=lambda: GfnAdam)

    def setup(self, params):
        params = list(params)
        if self.max_norm is not None and any(getattr(p, "_gfn_link", None) is not None for g in params
                                             for p in (g["params"] if isinstance(g, dict) else [g])):
            raise NotImplementedError(
                "GfnAdamOptimizerConfig: max_norm is set, but engine-linked parameters (hash tables, decoder, appearance embedding) keep their "
                "gradients in the engine's buffers (p.grad is None), so Optimizers' clip_grad_norm_ would silently clip nothing.  Leave max_norm = None "
                "(gfnerf/config.py:130) or use torch.optim.Adam (AdamOptimizerConfig), which gets dense gradients.")
        return super().setup(params)

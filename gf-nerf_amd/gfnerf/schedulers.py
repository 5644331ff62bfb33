"""GFNerfExponentialDecayScheduler (nerfstudio/engine/schedulers.py:113-184 of the reference tree): exponential decay
from lr_init to lr_final over max_steps in the global stage, restarted for every focal sub-dataset, optional warm-up.
Same config fields and the same LambdaLR wiring as the reference, so Optimizers (engine/optimizers.py) can drive it;
`lr_multiplier` is the pure function for callers that step the learning rate themselves (HotPath.train_step)."""
import math
from dataclasses import dataclass
from typing import Optional


@dataclass
class GFNerfExponentialDecaySchedulerConfig:
    lr_pre_warmup: float = 1e-8
    lr_final: Optional[float] = None
    warmup_steps: int = 0
    max_steps: int = 100000
    ramp: str = "cosine"
    n_split_dataset: int = 1
    n_dataset_circles: int = 1
    steps_per_split_dataset: int = 1000
    steps_perssampler_init: int = 10000

    def setup(self):
        return GFNerfExponentialDecayScheduler(self)


def _relative_step(step: int, c: GFNerfExponentialDecaySchedulerConfig) -> int:
    """Steps since the current stage (global stage, or the current focal sub-dataset visit) began, plus one
    sub-dataset length per completed circle (:165-171)."""
    if c.steps_perssampler_init > 0 and step < c.steps_perssampler_init:
        return step
    since = step - c.steps_perssampler_init
    per_circle = c.steps_per_split_dataset * c.n_split_dataset
    circles, in_circle = divmod(since, per_circle)
    return in_circle % c.steps_per_split_dataset + circles * c.steps_per_split_dataset


def lr_multiplier(step: int, lr_init: float, c: GFNerfExponentialDecaySchedulerConfig) -> float:
    lr_final = lr_init if c.lr_final is None else c.lr_final
    if step < c.warmup_steps:
        frac = min(max(step / c.warmup_steps, 0.0), 1.0)
        if c.ramp == "cosine":  # the reference mixes towards 1 here, not towards lr_init (:150-153); kept
            lr = c.lr_pre_warmup + (1 - c.lr_pre_warmup) * math.sin(0.5 * math.pi * frac)
        else:
            lr = c.lr_pre_warmup + (lr_init - c.lr_pre_warmup) * step / c.warmup_steps
    else:
        t = (_relative_step(step, c) - c.warmup_steps) / (c.max_steps - c.warmup_steps)
        t = min(max(t, 0.0), 1.0)
        lr = math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return lr / lr_init


class GFNerfExponentialDecayScheduler:
    def __init__(self, config: GFNerfExponentialDecaySchedulerConfig):
        self.config = config

    def get_scheduler(self, optimizer, lr_init: float):
        from torch.optim import lr_scheduler

        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda step: lr_multiplier(step, lr_init, self.config))

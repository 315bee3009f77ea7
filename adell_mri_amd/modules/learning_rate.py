"""Learning-rate schedule of the training path (mirror of
adell_mri/modules/learning_rate.py:7-30,106-212). Host-side scalar arithmetic only."""
import math

from torch.optim.lr_scheduler import _LRScheduler


def float_to_epochs(v, max_epochs: int) -> int:
    """A float >= 1 is a number of epochs; a float < 1 a fraction of ``max_epochs``."""
    if isinstance(v, float):
        v = int(v) if v >= 1.0 else int(v * max_epochs)
    return v


class CosineAnnealingWithWarmupLR(_LRScheduler):
    """Linear warm-up to the base LR, flat until ``start_decay``, then cosine to ``eta_min``."""

    def __init__(self, optimizer, T_max: int, n_warmup_steps: int = 0, eta_min: int = 0,
                 last_epoch: int = -1, verbose: bool = False, start_decay: int = None):
        self.T_max = T_max
        self.eta_min = eta_min
        self.initial_lr = eta_min
        if start_decay is None:
            start_decay = n_warmup_steps
        self.n_warmup_steps = float_to_epochs(n_warmup_steps, T_max)
        self.start_decay = float_to_epochs(start_decay, T_max)
        self.last_lr = None
        self.verbose = verbose
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return self._get_closed_form_lr()

    def _get_closed_form_lr(self):
        le = self.last_epoch
        nws, ssd = float(self.n_warmup_steps), float(self.start_decay)
        if le < nws and nws > 0:
            return [(b - self.initial_lr) * ((le + 1) / nws) + self.eta_min for b in self.base_lrs]
        if le <= ssd:
            return list(self.base_lrs)
        r = max(nws, ssd)
        span = self.T_max - r
        return [self.eta_min + (b - self.eta_min) * (1 + math.cos(math.pi * (le - r) / span)) / 2
                for b in self.base_lrs]

    def step(self, step=None):
        self._step_count += 1
        self.last_epoch = self.last_epoch + 1 if step is None else step
        values = self._get_closed_form_lr()
        for group, lr in zip(self.optimizer.param_groups, values):
            group["lr"] = lr
        self._last_lr = [g["lr"] for g in self.optimizer.param_groups]

"""Fused optimisers over ONE flat fp32 parameter buffer per group.

``FlatParameters`` re-homes every parameter of a group into a single contiguous
device buffer (and ``.grad`` into a matching flat gradient buffer), so that an
optimiser step is one HIP launch per group and the data-parallel gradient
exchange (``parallel.GradSync``) is an all-reduce of that same buffer.
``FusedSGD`` follows ``torch.optim.SGD(momentum, nesterov, weight_decay)`` as
configured by the reference (adell_mri/modules/segmentation/pl.py:563-569);
``FusedAdamW`` follows ``torch.optim.AdamW`` (self_supervised/pl.py:245-250).
"""
import torch

from . import ops


class FlatParameters:
    """Flatten ``params`` (leaf fp32 Parameters on one CUDA device)."""

    def __init__(self, params):
        self.params = list(params)
        if not self.params:
            raise ValueError("FlatParameters: empty parameter list")
        dev = self.params[0].device
        self.offsets = []
        n = 0
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("FlatParameters: all parameters must be fp32 on one device")
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4  # keep every slice 16-byte aligned
        self.numel = n
        self.data = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.data[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):  # re-attach if a caller set them to None
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class _FusedBase(torch.optim.Optimizer):
    def _flat(self, group):
        flat = group.get("_flat")
        if flat is None:
            # frozen parameters (e.g. an EMA shadow) never receive gradients: torch.optim skips
            # them, so they stay out of the flat buffer
            flat = FlatParameters([p for p in group["params"] if p.requires_grad])
            group["_flat"] = flat
        return flat

    @property
    def flat_groups(self):
        return [self._flat(g) for g in self.param_groups]

    def zero_grad(self, set_to_none: bool = False):
        for g in self.param_groups:
            self._flat(g).zero_grad()


class FusedSGD(_FusedBase):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0,
                 nesterov=False):
        if dampening != 0.0:
            raise NotImplementedError("FusedSGD: dampening must be 0")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay,
                                      nesterov=nesterov, grad_scale=1.0))
        for g in self.param_groups:
            self._flat(g)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, g in enumerate(self.param_groups):
            flat = self._flat(g)
            st = self.state.setdefault(f"flat{gi}", {})
            first = "momentum_buffer" not in st
            if first and g["momentum"] != 0.0:
                st["momentum_buffer"] = torch.zeros_like(flat.data)
            ops.sgd_step(flat.data, flat.grad, st.get("momentum_buffer"), g["lr"], g["momentum"],
                         g["weight_decay"], g["nesterov"], first, g.get("grad_scale", 1.0))
        return loss


class FusedAdamW(_FusedBase):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      grad_scale=1.0))
        for g in self.param_groups:
            self._flat(g)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, g in enumerate(self.param_groups):
            flat = self._flat(g)
            st = self.state.setdefault(f"flat{gi}", {})
            if "step" not in st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(flat.data)
                st["exp_avg_sq"] = torch.zeros_like(flat.data)
            st["step"] += 1
            ops.adamw_step(flat.data, flat.grad, st["exp_avg"], st["exp_avg_sq"], g["lr"],
                           g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], st["step"],
                           g.get("grad_scale", 1.0))
        return loss

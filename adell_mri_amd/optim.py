"""Fused optimisers over ONE flat fp32 parameter buffer per group.

``FlatParameters`` re-homes every parameter of a group into a single contiguous
device buffer (and ``.grad`` into a matching flat gradient buffer), so that an
optimiser step is one HIP launch per group and the data-parallel gradient
exchange (``parallel.GradSync``) is an all-reduce of that same buffer.
``FusedSGD`` follows ``torch.optim.SGD(momentum, nesterov, weight_decay)`` as
configured by the reference (adell_mri/modules/segmentation/pl.py:563-569);
``FusedAdamW`` follows ``torch.optim.AdamW`` (self_supervised/pl.py:245-250).
"""
import numpy as np
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

    CHUNK = 16384

    def zero_grad(self, set_to_none=True):
        """Zero the flat gradient. ``set_to_none`` (torch's default) also detaches the
        parameters from it: autograd then hands each parameter its freshly computed gradient
        tensor (no per-parameter accumulate kernels) and ``collect()`` gathers them into the
        flat buffer with one launch. Otherwise ``p.grad`` stays a view of the flat buffer."""
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):
            p.grad = None if set_to_none else self.grad[o:o + p.numel()].view(p.shape)

    def _upload_and_copy(self, rows):
        """Stage the (pointer, offset, count) table through a small ring of pinned host buffers
        (asynchronous upload, no host synchronisation unless the ring wraps onto a copy that is
        still in flight) and launch the multi-copy."""
        n = len(rows)
        ring = getattr(self, "_ring", None)
        if ring is None or ring[0][0].shape[0] < n:
            cap = max(2 * n, 1024)
            ring = [(torch.empty((cap, 3), dtype=torch.int64).pin_memory(),
                     torch.empty((cap, 3), dtype=torch.int64, device=self.grad.device),
                     torch.cuda.Event()) for _ in range(4)]
            self._ring, self._ring_pos = ring, 0
        host, dev, ev = ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(ring)
        ev.synchronize()
        host[:n] = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64))
        dev[:n].copy_(host[:n], non_blocking=True)
        ev.record()
        ops.multi_copy(dev, n, self.grad)

    def collect(self):
        """Copy every parameter gradient that is not already a view of the flat buffer into its
        slot (one multi-copy launch), then point ``p.grad`` at the slots."""
        base = self.grad.data_ptr()
        todo, keep = [], []
        for p, o in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() == base + 4 * o:
                continue
            if g.dtype != torch.float32 or g.device != self.grad.device:
                raise ValueError("FlatParameters.collect: gradients must be fp32 on the GPU")
            g = g.contiguous()
            keep.append(g)
            todo.append((g.data_ptr(), o, g.numel()))
        if todo:
            # one row per CHUNK-element piece of every gradient, built without a Python loop
            ptr, off, num = np.array(todo, dtype=np.int64).T
            k = (num + self.CHUNK - 1) // self.CHUNK
            idx = np.repeat(np.arange(len(k)), k)
            start = (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k)) * self.CHUNK
            rows = np.stack([ptr[idx] + 4 * start, off[idx] + start,
                             np.minimum(self.CHUNK, num[idx] - start)], 1)
            self._upload_and_copy(rows)
        for p, o in zip(self.params, self.offsets):
            if p.grad is not None:
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

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            self._flat(g).zero_grad(set_to_none)

    def collect_grads(self):
        """Gather the parameters' gradients into the flat buffers (idempotent)."""
        for g in self.param_groups:
            self._flat(g).collect()


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
        self.collect_grads()
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
    decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      grad_scale=1.0))
        for g in self.param_groups:
            self._flat(g)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.collect_grads()
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
                           g.get("grad_scale", 1.0), decoupled=self.decoupled)
        return loss


class FusedAdam(FusedAdamW):
    """torch.optim.Adam: same kernel, weight decay added to the gradient (default 0)."""

    decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

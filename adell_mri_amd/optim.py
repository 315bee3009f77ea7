"""Fused optimisers over ONE flat fp32 parameter buffer per group.

``FlatParameters`` re-homes every parameter of a group into a single contiguous
device buffer (and ``.grad`` into a matching flat gradient buffer), so that an
optimiser step is one HIP launch per group and the data-parallel gradient
exchange (``parallel.GradSync``) is an all-reduce of slices of that same buffer.
``FusedSGD`` follows ``torch.optim.SGD(momentum, nesterov, weight_decay)`` as
configured by the reference (adell_mri/modules/segmentation/pl.py:563-569);
``FusedAdamW`` follows ``torch.optim.AdamW`` (self_supervised/pl.py:245-250).

Kept from ``torch.optim`` so that the reference's checkpoints interchange:

* ``state_dict()`` / ``load_state_dict()`` use the per-parameter layout of the torch
  optimisers (``momentum_buffer`` / ``exp_avg`` / ``exp_avg_sq`` / ``step`` keyed by
  parameter index); loading copies INTO the flat buffers, the parameters keep aliasing them.
* a parameter whose ``.grad`` is ``None`` is skipped by ``step()`` (no weight decay, no
  momentum update), and its state starts at the first step it does receive a gradient.
"""
import numpy as np
import torch

from . import ops


class FlatParameters:
    """Flatten ``params`` (leaf fp32 Parameters on one CUDA device)."""

    def __init__(self, params, device=None):
        # An empty list is legal (a parameter group whose parameters are all frozen, e.g. the SSL
        # encoder handed to the U-Net with lr_encoder == 0: entrypoints/segmentation/train.py:718-724):
        # zero-length buffers on ``device``; every loop below then has nothing to do.
        self.params = list(params)
        if not self.params and device is None:
            raise ValueError("FlatParameters: an empty parameter list needs a device")
        dev = self.params[0].device if self.params else torch.device(device)
        self.offsets = []
        n = 0
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("FlatParameters: all parameters must be fp32 on one device")
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4  # keep every slice 16-byte aligned
        self.numel = n
        self.ends = self.offsets[1:] + [n]   # padded end of every slice
        self.data = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        # parallel.GradSync: the slot holds this step's (reduced) gradient although ``p.grad`` is
        # detached from it while the collective runs
        self.reduced = [False] * len(self.params)
        # the per-step loops below run once per parameter on the host (a few hundred parameters:
        # a millisecond of a 20 ms step if they slice, view and compare tensors): slot views,
        # slot addresses and the "has a gradient" mask are made once / once per collect
        self._slots = [self.grad[o:o + p.numel()].view(p.shape)
                       for p, o in zip(self.params, self.offsets)]
        self._slot_ptr = [self.grad.data_ptr() + 4 * o for o in self.offsets]
        self._has = None
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                view = self.data[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self._slots[i]
                # functional._side_ok: gradients of these parameters are read only through
                # collect() / zero_grad(), which join the weight-gradient stream first
                p._adell_flat = True

    CHUNK = 16384

    def slot(self, i):
        return self._slots[i]

    def zero_grad(self, set_to_none=True):
        """Zero the flat gradient. ``set_to_none`` (torch's default) also detaches the
        parameters from it: autograd then hands each parameter its freshly computed gradient
        tensor (no per-parameter accumulate kernels) and ``collect()`` gathers them into the
        flat buffer with one launch. Otherwise ``p.grad`` stays a view of the flat buffer."""
        from . import functional as HF
        self.grad.zero_()
        self._has = None
        if set_to_none:
            for p in self.params:
                p.grad = None
        else:
            for p, v in zip(self.params, self._slots):
                p.grad = v
        HF.reset_uses(self.params)

    def _upload_and_copy(self, rows):
        """Stage the (pointer, offset, count) table through a small ring of pinned host buffers
        (asynchronous upload, no host synchronisation unless the ring wraps onto a copy that is
        still in flight) and launch the multi-copy."""
        n = len(rows)
        if self.grad.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a graph capture (trainer.StepRunner.enable_graph): the table is a node's static
            # input. Nothing may be allocated on the host here (pinning memory invalidates the
            # capture): one entry of the ring the eager steps built is taken out of it for good --
            # kept alive with the graph, never written again, no event to wait on later
            ring = getattr(self, "_ring", None)
            if ring is None or ring[0][0].shape[0] < n or len(ring) < 2:
                raise RuntimeError("FlatParameters.collect inside a graph capture: run eager steps "
                                   "first (they size the pinned staging ring)")
            host, dev, _ev = ring.pop(self._ring_pos % len(ring))
            self._ring_pos = self._ring_pos % len(ring)
            host[:n] = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64))
            dev[:n].copy_(host[:n], non_blocking=True)
            self._graph_tables = getattr(self, "_graph_tables", []) + [(host, dev)]
            ops.multi_copy(dev, n, self.grad)
            return
        ring = getattr(self, "_ring", None)
        if ring is None or ring[0][0].shape[0] < n:
            cap = max(2 * n, 1024)
            ring = [(torch.empty((cap, 3), dtype=torch.int64).pin_memory(),
                     torch.empty((cap, 3), dtype=torch.int64, device=self.grad.device),
                     torch.cuda.Event()) for _ in range(8)]
            self._ring, self._ring_pos = ring, 0
        host, dev, ev = ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(ring)
        ev.synchronize()
        host[:n] = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64))
        dev[:n].copy_(host[:n], non_blocking=True)
        ev.record()
        ops.multi_copy(dev, n, self.grad)

    def collect(self, indices=None, on_side_stream=False):
        """Copy every parameter gradient (of ``indices``, default all) that is not already a
        view of the flat buffer into its slot (one multi-copy launch), then point ``p.grad`` at
        the slots. Returns nothing; parameters without a gradient keep ``grad is None`` and a
        zero slot. ``on_side_stream``: the caller runs this on the weight-gradient stream itself
        (GradSync's buckets: in order behind the gradients, without stalling the main stream);
        otherwise the current stream first waits for that stream."""
        from . import functional as HF
        if not on_side_stream:
            HF.join_side_stream()     # weight gradients still running on the side stream
        todo, keep = [], []
        params, offsets, slots, slot_ptr = self.params, self.offsets, self._slots, self._slot_ptr
        idx = range(len(params)) if indices is None else indices
        dev, f32 = self.grad.device, torch.float32
        has = np.zeros(len(params), dtype=bool) if indices is None else None
        for i in idx:
            g = params[i].grad
            if g is None:
                continue
            if has is not None:
                has[i] = True
            ptr = g.data_ptr()
            if ptr == slot_ptr[i]:
                continue
            n = g.numel()
            if n == 0:
                continue
            if g.dtype != f32 or g.device != dev:
                raise ValueError("FlatParameters.collect: gradients must be fp32 on the GPU")
            if not g.is_contiguous():
                g = g.contiguous()
                ptr = g.data_ptr()
            if on_side_stream:       # (may have been produced on another stream: alive till the join)
                HF.side_keep(g)
            keep.append(g)
            todo.append((ptr, offsets[i], n))
        if todo:
            # one row per CHUNK-element piece of every gradient, built without a Python loop
            ptr, off, num = np.array(todo, dtype=np.int64).T
            k = (num + self.CHUNK - 1) // self.CHUNK
            ix = np.repeat(np.arange(len(k)), k)
            start = (np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k)) * self.CHUNK
            rows = np.stack([ptr[ix] + 4 * start, off[ix] + start,
                             np.minimum(self.CHUNK, num[ix] - start)], 1)
            self._upload_and_copy(rows)
        for i in idx:
            p = params[i]
            if p.grad is not None:
                p.grad = slots[i]
        self._has = has

    def has_grad(self):
        """Which parameters hold a gradient. After a full ``collect()`` its mask is the starting
        point, and every entry it marked is looked at again: a ``.grad`` set to None since then (a
        parameter frozen for this step, gradient filtering between the all-reduce and the step) is
        skipped, as ``torch.optim`` would skip it -- its slot still holds the stale gradient. A
        gradient ASSIGNED by hand after ``collect()`` needs another ``collect()`` anyway."""
        if self._has is not None:
            has, params = self._has.copy(), self.params
            for i in np.flatnonzero(has):
                if params[i].grad is None:
                    has[i] = False
            return has
        return np.fromiter((p.grad is not None for p in self.params), dtype=bool,
                           count=len(self.params))

    def runs(self, active, key=None):
        """Maximal runs of consecutive parameters with ``active[i]`` true and equal ``key[i]``:
        [(first index, element offset, element end)], ends padded so that a run is one
        contiguous 16-byte-aligned slice of the flat buffers."""
        out, i, n = [], 0, len(self.params)
        active = np.asarray(active)
        if n and active.all() and self.numel > 0:       # the usual step: ONE run, no scan
            k = None if key is None else np.asarray(key)
            if k is None or (k == k[0]).all():
                return [(0, self.offsets[0], self.ends[-1])]
        while i < n:
            if not active[i]:
                i += 1
                continue
            j = i
            while j + 1 < n and active[j + 1] and (key is None or key[j + 1] == key[i]):
                j += 1
            if self.ends[j] > self.offsets[i]:   # zero-element parameters make empty runs
                out.append((i, self.offsets[i], self.ends[j]))
            i = j + 1
        return out


class _FusedBase(torch.optim.Optimizer):
    _state_names = ()

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        # flat buffers live beside param_groups (not inside: torch pickles / deep-copies groups)
        self._flats = {}
        self._flat_state = {}
        for gi in range(len(self.param_groups)):
            self._flat(gi)

    def _flat(self, gi):
        flat = self._flats.get(gi)
        if flat is None:
            # frozen parameters (e.g. an EMA shadow) never receive gradients: torch.optim skips
            # them, so they stay out of the flat buffer
            group = self.param_groups[gi]
            dev = next((p.device for g in self.param_groups for p in g["params"]), None)
            flat = FlatParameters([p for p in group["params"] if p.requires_grad], device=dev)
            self._flats[gi] = flat
            self._flat_state[gi] = {"steps": np.zeros(len(flat.params), dtype=np.int64)}
        return flat

    @property
    def flat_groups(self):
        return [self._flat(gi) for gi in range(len(self.param_groups))]

    def zero_grad(self, set_to_none: bool = True):
        for flat in self.flat_groups:
            flat.zero_grad(set_to_none)

    def collect_grads(self):
        """Gather the parameters' gradients into the flat buffers (idempotent)."""
        for flat in self.flat_groups:
            flat.collect()

    def _buffer(self, gi, name):
        st = self._flat_state[gi]
        if name not in st:
            st[name] = torch.zeros_like(self._flat(gi).data)
        return st[name]

    # ---- torch.optim-compatible (de)serialisation ------------------------------------------
    def _hyper(self, group):
        return {k: v for k, v in group.items() if k != "params"}

    def state_dict(self):
        """Same layout as ``torch.optim.Optimizer.state_dict()`` of the optimiser this class
        replaces: ``state[index]`` holds per-parameter tensors (copies), ``param_groups`` the
        hyper-parameters and parameter indices. Frozen parameters have no state, as in torch."""
        state, groups, start = {}, [], 0
        for gi, group in enumerate(self.param_groups):
            flat, st = self._flat(gi), self._flat_state[gi]
            slot = {id(p): i for i, p in enumerate(flat.params)}
            ids = list(range(start, start + len(group["params"])))
            for pid, p in zip(ids, group["params"]):
                i = slot.get(id(p))
                if i is None or st["steps"][i] == 0:
                    continue
                o, n = flat.offsets[i], p.numel()
                state[pid] = self._param_state(st, i, o, n, p.shape)
            groups.append({**self._hyper(group), "params": ids})
            start += len(ids)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        groups = state_dict["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        start = 0
        for gi, (group, saved) in enumerate(zip(self.param_groups, groups)):
            if len(saved["params"]) != len(group["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't "
                                 "match the size of optimizer's group")
            keep_scale = group.get("grad_scale", 1.0)
            for k, v in saved.items():
                if k != "params":
                    group[k] = v
            group["grad_scale"] = keep_scale   # world-size dependent, not a checkpoint property
            flat, st = self._flat(gi), self._flat_state[gi]
            slot = {id(p): i for i, p in enumerate(flat.params)}
            st["steps"][:] = 0
            for name in self._state_names:
                if name in st:
                    st[name].zero_()
            for pid, p in zip(saved["params"], group["params"]):
                ent = state_dict["state"].get(pid, state_dict["state"].get(str(pid)))
                i = slot.get(id(p))
                if ent is None or i is None:
                    continue
                self._load_param_state(gi, st, i, flat.offsets[i], p.numel(), ent)
            start += len(group["params"])
        ops._weights_changed()


class FusedSGD(_FusedBase):
    _state_names = ("momentum_buffer",)

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0,
                 nesterov=False):
        if dampening != 0.0:
            raise NotImplementedError("FusedSGD: dampening must be 0")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening,
                                      weight_decay=weight_decay, nesterov=nesterov,
                                      grad_scale=1.0))

    def _param_state(self, st, i, o, n, shape):
        if "momentum_buffer" not in st:
            return {"momentum_buffer": None}
        return {"momentum_buffer": st["momentum_buffer"][o:o + n].view(shape).clone()}

    def _load_param_state(self, gi, st, i, o, n, ent):
        buf = ent.get("momentum_buffer")
        if buf is not None:
            self._buffer(gi, "momentum_buffer")[o:o + n].copy_(buf.reshape(-1))
        st["steps"][i] = 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.collect_grads()
        for gi, g in enumerate(self.param_groups):
            flat, st = self._flat(gi), self._flat_state[gi]
            buf = self._buffer(gi, "momentum_buffer") if g["momentum"] != 0.0 else None
            active = flat.has_grad()
            first = st["steps"] == 0
            # one launch per run of parameters that share "has a gradient" and "first step"
            # (normally ONE run: the whole buffer)
            for i, lo, hi in flat.runs(active, first):
                ops.sgd_step(flat.data[lo:hi], flat.grad[lo:hi],
                             None if buf is None else buf[lo:hi], g["lr"], g["momentum"],
                             g["weight_decay"], g["nesterov"], bool(first[i]),
                             g.get("grad_scale", 1.0))
            st["steps"][active] += 1
        return loss


class FusedAdamW(_FusedBase):
    decoupled = True
    _state_names = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      grad_scale=1.0))

    def _param_state(self, st, i, o, n, shape):
        return {"step": torch.tensor(float(st["steps"][i])),
                "exp_avg": st["exp_avg"][o:o + n].view(shape).clone(),
                "exp_avg_sq": st["exp_avg_sq"][o:o + n].view(shape).clone()}

    def _load_param_state(self, gi, st, i, o, n, ent):
        self._buffer(gi, "exp_avg")[o:o + n].copy_(ent["exp_avg"].reshape(-1))
        self._buffer(gi, "exp_avg_sq")[o:o + n].copy_(ent["exp_avg_sq"].reshape(-1))
        st["steps"][i] = int(float(ent["step"]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.collect_grads()
        for gi, g in enumerate(self.param_groups):
            flat, st = self._flat(gi), self._flat_state[gi]
            m, v = self._buffer(gi, "exp_avg"), self._buffer(gi, "exp_avg_sq")
            active = flat.has_grad()
            st["steps"][active] += 1
            # the bias correction depends on a parameter's own step count (torch keeps one per
            # parameter): one launch per run of equal counts (normally ONE run)
            for i, lo, hi in flat.runs(active, st["steps"]):
                ops.adamw_step(flat.data[lo:hi], flat.grad[lo:hi], m[lo:hi], v[lo:hi], g["lr"],
                               g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                               int(st["steps"][i]), g.get("grad_scale", 1.0),
                               decoupled=self.decoupled)
        return loss


class FusedAdam(FusedAdamW):
    """torch.optim.Adam: same kernel, weight decay added to the gradient (default 0)."""

    decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)


class _FusedScalarPlan(_FusedBase):
    """Shared driver of the optimisers whose step is ``ops.optim_step`` with host-computed
    per-step scalars: ``_kind``, ``_buffers`` (flat state names: torch state_dict keys) and
    ``_scalars(group, t, extra)`` -> the five floats of the kernel."""

    _kind = None
    _buffers = ()
    _extra_state = ()        # per-parameter host scalars kept beside ``steps`` (NAdam: mu_product)

    @property
    def _state_names(self):
        return self._buffers

    def _param_state(self, st, i, o, n, shape):
        out = {"step": torch.tensor(float(st["steps"][i]))}
        for name in self._extra_state:
            out[name] = torch.tensor(float(st[name][i]))
        for name in self._buffers:
            out[name] = st[name][o:o + n].view(shape).clone()
        return out

    def _load_param_state(self, gi, st, i, o, n, ent):
        for name in self._buffers:
            self._buffer(gi, name)[o:o + n].copy_(ent[name].reshape(-1))
        for name in self._extra_state:
            self._extras(gi, name)[i] = float(ent[name])
        st["steps"][i] = int(float(ent["step"]))

    def _extras(self, gi, name):
        st = self._flat_state[gi]
        if name not in st or torch.is_tensor(st[name]):
            st[name] = np.ones(len(self._flat(gi).params), dtype=np.float64)
        return st[name]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.collect_grads()
        for gi, g in enumerate(self.param_groups):
            flat, st = self._flat(gi), self._flat_state[gi]
            bufs = [self._buffer(gi, name) for name in self._buffers]
            extras = {name: self._extras(gi, name) for name in self._extra_state}
            active = flat.has_grad()
            st["steps"][active] += 1
            key = st["steps"] if not extras else np.stack(
                [st["steps"].astype(np.float64)] + [extras[n] for n in self._extra_state], 1)
            # one launch per run of parameters with equal step counts (normally ONE run)
            for i, lo, hi in flat.runs(active, [tuple(np.atleast_1d(k)) for k in key]):
                c5 = self._scalars(g, int(st["steps"][i]), {n: extras[n] for n in extras}, i, lo, hi,
                                   flat, active)
                ops.optim_step(self._kind, flat.data[lo:hi], flat.grad[lo:hi], bufs[0][lo:hi],
                               bufs[1][lo:hi] if len(bufs) > 1 else None, g["weight_decay"],
                               g["eps"], g.get("grad_scale", 1.0), c5)
        return loss


class FusedAdamax(_FusedScalarPlan):
    _kind, _buffers = "adamax", ("exp_avg", "exp_inf")

    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      grad_scale=1.0))

    def _scalars(self, g, t, extras, i, lo, hi, flat, active):
        b1, b2 = g["betas"]
        return [b1, b2, g["lr"] / (1.0 - b1 ** t)]


class FusedAdagrad(_FusedScalarPlan):
    _kind, _buffers = "adagrad", ("sum",)

    def __init__(self, params, lr=1e-2, lr_decay=0.0, weight_decay=0.0,
                 initial_accumulator_value=0.0, eps=1e-10):
        if initial_accumulator_value != 0.0:
            raise NotImplementedError("FusedAdagrad: initial_accumulator_value must be 0")
        super().__init__(params, dict(lr=lr, lr_decay=lr_decay, eps=eps, weight_decay=weight_decay,
                                      initial_accumulator_value=0.0, grad_scale=1.0))

    def _scalars(self, g, t, extras, i, lo, hi, flat, active):
        return [g["lr"] / (1.0 + (t - 1) * g["lr_decay"])]


class FusedNAdam(_FusedScalarPlan):
    _kind, _buffers, _extra_state = "nadam", ("exp_avg", "exp_avg_sq"), ("mu_product",)

    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 momentum_decay=4e-3):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      momentum_decay=momentum_decay, grad_scale=1.0))

    def _scalars(self, g, t, extras, i, lo, hi, flat, active):
        b1, b2 = g["betas"]
        md = g["momentum_decay"]
        mu = b1 * (1.0 - 0.5 * 0.96 ** (t * md))
        mu_next = b1 * (1.0 - 0.5 * 0.96 ** ((t + 1) * md))
        # every parameter of the run shares the step count and the product so far
        j = i
        while j < len(flat.params) and flat.offsets[j] < hi:
            if active[j]:
                extras["mu_product"][j] *= mu
            j += 1
        prod = extras["mu_product"][i]
        return [b1, b2, 1.0 - b2 ** t, g["lr"] * (1.0 - mu) / (1.0 - prod),
                g["lr"] * mu_next / (1.0 - prod * mu_next)]


class FusedRAdam(_FusedScalarPlan):
    _kind, _buffers = "radam", ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      grad_scale=1.0))

    def _scalars(self, g, t, extras, i, lo, hi, flat, active):
        b1, b2 = g["betas"]
        bc2 = 1.0 - b2 ** t
        rho_inf = 2.0 / (1.0 - b2) - 1.0
        rho_t = rho_inf - 2.0 * t * (b2 ** t) / bc2
        rect = -1.0
        if rho_t > 5.0:
            rect = (((rho_t - 4) * (rho_t - 2) * rho_inf)
                    / ((rho_inf - 4) * (rho_inf - 2) * rho_t)) ** 0.5 * bc2 ** 0.5
        return [b1, b2, g["lr"] / (1.0 - b1 ** t), rect]


class FusedRMSprop(_FusedScalarPlan):
    _kind, _buffers = "rmsprop", ("square_avg",)

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0, momentum=0.0,
                 centered=False):
        if momentum != 0.0 or centered:
            raise NotImplementedError("FusedRMSprop: momentum 0 and centered=False only")
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay,
                                      momentum=0.0, centered=False, grad_scale=1.0))

    def _scalars(self, g, t, extras, i, lo, hi, flat, active):
        return [g["alpha"], g["lr"]]

"""Data-parallel plumbing: one process per GPU, RCCL over xGMI through
``torch.distributed`` (backend "nccl" is RCCL on ROCm; "gloo" in CPU tests).

The reference gets data parallelism from ``lightning.Trainer(strategy="ddp")``
(adell_mri/entrypoints/segmentation/train.py:799-819, utils/pl_utils.py:424-458):
torch DDP's bucketed sum-all-reduce of gradients, launched from backward hooks and divided
by the world size. Here the optimiser already owns ONE flat gradient buffer per parameter
group (``optim.FlatParameters``), so a bucket is a contiguous slice of that buffer:

* the flat buffer is cut into a few buckets of consecutive parameters (xGMI rings are
  per-link bound: few, large messages -- 4 buckets of >= 4 MB by default);
* a post-accumulate-grad hook on every parameter counts arrivals; when the last parameter
  of a bucket has its gradient (backward visits the buckets last-to-first), the bucket's
  gradients are gathered into their slots by one multi-copy launch and an ASYNC all-reduce
  of the slice is issued on the process group's stream -- it overlaps the rest of backward;
* ``all_reduce()`` (called before ``optimizer.step``) sends whatever has not gone out yet
  (parameters without a gradient leave their bucket incomplete) and waits for every handle;
* the division by the world size is folded into the fused optimiser kernel
  (``grad_scale``), not a separate pass.
"""
import os
import statistics
import time

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from torchrun-style env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # ADELL_DIST_BACKEND=gloo: rehearsal of the N > 1 path without RCCL
            backend = os.environ.get("ADELL_DIST_BACKEND") or (
                "nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_indices(n_items, rank, world):
    """Indices of the units (volumes) rank ``rank`` owns out of ``n_items``."""
    return list(range(rank, n_items, world))


def all_reduce_flat(buffers, chunk_elems, async_op=False):
    """Sum-all-reduce each 1-D buffer in chunks of ``chunk_elems`` elements."""
    if world_size() == 1:
        return
    handles = []
    for g in buffers:
        for o in range(0, g.numel(), chunk_elems):
            h = dist.all_reduce(g[o:o + chunk_elems], op=dist.ReduceOp.SUM, async_op=async_op)
            if async_op:
                handles.append(h)
    for h in handles:
        h.wait()


def plan_buckets(sizes, n_buckets=4, min_elems=1 << 20):
    """Cut consecutive parameters (element counts ``sizes``, in flat-buffer order) into at most
    ``n_buckets`` runs of roughly equal element count, none smaller than ``min_elems`` unless
    it is the only one. Returns [(first index, last index + 1)], in buffer order."""
    total = int(sum(sizes))
    n = max(1, min(int(n_buckets), total // max(int(min_elems), 1) or 1, len(sizes)))
    out, start, acc, left, remaining = [], 0, 0, n, total
    for i, s in enumerate(sizes):
        acc += int(s)
        # cut when this bucket has its share of what is left, keeping >= 1 parameter for each
        # of the buckets still to come
        if left > 1 and acc >= remaining / left and len(sizes) - (i + 1) >= left - 1:
            out.append((start, i + 1))
            start, remaining, left, acc = i + 1, remaining - acc, left - 1, 0
    out.append((start, len(sizes)))
    return out


class _Bucket:
    __slots__ = ("flat", "lo", "hi", "e0", "e1", "pending", "sent", "handle", "again", "issued")

    def __init__(self, flat, lo, hi):
        self.flat, self.lo, self.hi = flat, lo, hi
        self.e0, self.e1 = flat.offsets[lo], flat.ends[hi - 1]
        self.reset()

    def reset(self):
        self.pending = self.hi - self.lo   # parameters whose gradient has not arrived yet
        self.sent = False                  # the slice went out in this step
        self.handle = None                 # ... and this is its collective
        self.again = False                 # a later backward of the same step brought more
        self.issued = None                 # GradSync.profile: (event at the collective's issue, host time)


class GradSync:
    """Sum-all-reduce the flat gradient buffers of a fused optimiser.

    ``overlap`` (default: on when the world size is > 1, ``ADELL_DDP_OVERLAP=0`` turns it off)
    issues the collectives bucket by bucket from backward hooks; otherwise the whole buffer
    goes out in ``chunk_mb`` messages when ``all_reduce()`` is called. The bucket count grows
    with the message: ``n_buckets`` (default 4, ``ADELL_DDP_BUCKETS``) is a minimum, and a flat
    buffer is cut into at least one bucket per ``max_bucket_mb`` (default 32 MB: the 135-167 MB
    of gradients of the ViT / ConvNeXt / ResNet-encoder configurations go out in 5-6 pieces
    that start while backward still runs, instead of four 40 MB ones).

    A step is everything between two ``all_reduce()`` calls (``optimizer.zero_grad()`` also
    closes one: a step abandoned after a NaN leaves no stale state). Several backward passes
    per step are legal:

    * inside ``no_sync()`` nothing is sent: gradients accumulate locally, exactly as under
      ``DistributedDataParallel.no_sync()`` (Lightning's ``accumulate_grad_batches``);
    * without it, a bucket that already went out is not touched by the later backward: once
      sent, its parameters are detached from the flat slice (their next gradients arrive as
      fresh tensors), and ``all_reduce()`` reduces that second contribution separately and adds
      it to the first -- the result is the sum over ranks of ALL local gradients either way.

    ``find_unused_parameters``: a parameter that got no gradient on this rank but did on
    another still has to be stepped here, or the replicas drift (torch DDP writes the reduced
    gradient on every rank). With the flag on, ``all_reduce()`` also MAX-reduces a has-gradient
    bitmap and attaches the reduced slot to such parameters; it costs one host synchronisation
    per step, so it is off unless the model has rank-dependent control flow (DDP's own flag).
    """

    def __init__(self, optimizer, chunk_mb=64, async_op=False, overlap=None, n_buckets=None,
                 min_bucket_elems=None, max_bucket_mb=32, find_unused_parameters=False,
                 _force_overlap=False):
        self.optimizer = optimizer
        self.chunk = int(chunk_mb * 1024 * 1024 // 4)
        self.async_op = async_op
        self.world = world_size()
        self.find_unused = bool(find_unused_parameters)
        for g in optimizer.param_groups:
            g["grad_scale"] = 1.0 / self.world
        if overlap is None:
            overlap = self.world > 1 and os.environ.get("ADELL_DDP_OVERLAP", "1") != "0"
        # (_force_overlap: the single-rank RCCL test runs the hook -> async all-reduce -> wait path
        # on a one-process group; a production world of 1 has nothing to exchange)
        self.overlap = bool(overlap) and (self.world > 1 or (_force_overlap and dist.is_initialized()))
        self.buckets, self._hooks = [], []
        self._sync = True
        # exchange profile (bench.py's N > 1 line): with ``profile`` on, every step records when
        # each bucket's collective was issued and how long ``all_reduce()`` kept the main stream
        # waiting for handles -- see ``exchange_stats()``
        self.profile = False
        self._prof = []
        # functional._side_ok: with a process group up, weight gradients may be produced on the
        # side stream only for parameters whose exchange goes through this object (every read of a
        # gradient here is behind FlatParameters.collect, which joins that stream)
        for flat in getattr(optimizer, "flat_groups", ()):
            for p in getattr(flat, "params", ()):
                p._adell_gradsync = True
        if self.overlap:
            nb = int(os.environ.get("ADELL_DDP_BUCKETS", "4")) if n_buckets is None else n_buckets
            me = (1 << 20) if min_bucket_elems is None else min_bucket_elems
            self._install(nb, me, int(max_bucket_mb * 1024 * 1024 // 4))
            self._wrap_zero_grad()

    # ---- bucketed, overlapped path --------------------------------------------------------
    def _install(self, n_buckets, min_elems, max_elems):
        for flat in self.optimizer.flat_groups:
            if not flat.params:          # a group of frozen parameters: nothing to exchange
                continue
            sizes = [e - o for o, e in zip(flat.offsets, flat.ends)]
            nb = max(int(n_buckets), -(-int(sum(sizes)) // max(max_elems, 1)))
            for lo, hi in plan_buckets(sizes, nb, min_elems):
                b = _Bucket(flat, lo, hi)
                self.buckets.append(b)
                for i in range(lo, hi):
                    self._hooks.append(
                        flat.params[i].register_post_accumulate_grad_hook(self._make_hook(b)))
                    # (functional.side_run: these hooks join the weight-gradient stream before
                    # they read a gradient)
                    flat.params[i]._adell_gradsync = True

    def _wrap_zero_grad(self):
        """``optimizer.zero_grad()`` starts a new step: drop whatever an abandoned one left."""
        inner = getattr(self.optimizer, "zero_grad", None)
        if inner is None:
            return

        def zero_grad(*args, **kwargs):
            self.reset()
            return inner(*args, **kwargs)

        self.optimizer.zero_grad = zero_grad

    def _make_hook(self, bucket):
        def hook(_param):
            if not self._sync:
                return
            if bucket.sent:              # a second backward in this step (see the class notes)
                bucket.again = True
                return
            bucket.pending -= 1
            if bucket.pending == 0:
                self._send(bucket)
        return hook

    def _send(self, b):
        # With weight gradients in flight on their own stream (functional.side_run) the gather and
        # the collective queue BEHIND them on that stream: the main stream -- the backward-data
        # chain -- is not held up for a bucket (it waits for the handle before the optimiser).
        side = None
        if b.flat.grad.is_cuda and os.environ.get("ADELL_DDP_SEND_MAIN") is None:
            from . import functional as HF
            side = HF.side_stream_behind(torch.cuda.current_stream(b.flat.grad.device))
        if side is not None:
            with torch.cuda.stream(side):
                b.flat.collect(range(b.lo, b.hi), on_side_stream=True)
                if b.e1 > b.e0:
                    b.handle = dist.all_reduce(b.flat.grad[b.e0:b.e1], op=dist.ReduceOp.SUM,
                                               async_op=True)
        else:
            b.flat.collect(range(b.lo, b.hi))
            if b.e1 > b.e0:
                b.handle = dist.all_reduce(b.flat.grad[b.e0:b.e1], op=dist.ReduceOp.SUM,
                                           async_op=True)
        if self.profile:
            ev = None
            if b.flat.grad.is_cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(side if side is not None else torch.cuda.current_stream(b.flat.grad.device))
            b.issued = (ev, time.perf_counter())
        b.sent = True
        # detach the parameters from the slice while the collective may be running: a later
        # backward must not accumulate in place into memory the collective reads and writes
        for i in range(b.lo, b.hi):
            p = b.flat.params[i]
            if p.grad is not None:
                b.flat.reduced[i] = True
                p.grad = None

    def _finish(self, b):
        """After the bucket's collective: fold in what a later backward of the step produced,
        then hand the parameters their reduced gradients."""
        if b.handle is not None:
            b.handle.wait()
            b.handle = None
        flat = b.flat
        late = [i for i in range(b.lo, b.hi) if flat.params[i].grad is not None]
        if late:
            first = flat.grad[b.e0:b.e1].clone()      # sum over ranks of the first contribution
            flat.grad[b.e0:b.e1].zero_()
            flat.collect(late)
            if self.world > 1 or dist.is_initialized():
                dist.all_reduce(flat.grad[b.e0:b.e1], op=dist.ReduceOp.SUM)
            flat.grad[b.e0:b.e1].add_(first)
        for i in range(b.lo, b.hi):
            if flat.reduced[i] or flat.params[i].grad is not None:
                flat.params[i].grad = flat.slot(i)
            flat.reduced[i] = False

    def reset(self):
        """Forget the current step (waits for collectives still in flight)."""
        for b in self.buckets:
            if b.handle is not None:
                b.handle.wait()
            for i in range(b.lo, b.hi):
                b.flat.reduced[i] = False
            b.reset()

    def no_sync(self):
        """Context manager: backward passes inside it exchange nothing (their gradients
        accumulate locally and go out with the first backward outside it, or at
        ``all_reduce()``)."""
        outer = self

        class _NoSync:
            def __enter__(self):
                self.prev, outer._sync = outer._sync, False

            def __exit__(self, *exc):
                outer._sync = self.prev
                return False

        return _NoSync()

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for flat in self.optimizer.flat_groups:
            for p in flat.params:
                if getattr(p, "_adell_gradsync", False):
                    del p._adell_gradsync

    # ---- public ---------------------------------------------------------------------------
    def broadcast_parameters(self, src=0, module=None):
        """Rank ``src``'s parameters everywhere; with ``module`` also the tensors that are
        not in the optimiser's flat buffers (frozen parameters, buffers), as DDP does at
        construction."""
        if self.world == 1:
            return
        for flat in self.optimizer.flat_groups:
            if flat.data.numel() > 0:
                dist.broadcast(flat.data, src=src)
        if module is not None:
            flat_ids = {id(p) for f in self.optimizer.flat_groups for p in f.params}
            for t in list(module.parameters()) + list(module.buffers()):
                if id(t) not in flat_ids and t.numel() > 0:
                    dist.broadcast(t.data, src=src)
        from . import ops

        ops._weights_changed()

    def _attach_remote_grads(self):
        """find_unused_parameters: a parameter another rank produced a gradient for gets its
        (reduced) slot here too."""
        flats = [f for f in self.optimizer.flat_groups if f.params]
        if not flats:
            return
        dev = flats[0].grad.device
        have = torch.tensor([p.grad is not None for f in flats for p in f.params],
                            dtype=torch.int32, device=dev)
        dist.all_reduce(have, op=dist.ReduceOp.MAX)
        have = have.cpu().tolist()
        k = 0
        for f in flats:
            for i, p in enumerate(f.params):
                if have[k] and p.grad is None:
                    p.grad = f.slot(i)
                k += 1

    def all_reduce(self):
        """Complete the gradient exchange of this step: after it returns (stream-ordered), every
        flat gradient buffer holds the sum over ranks and every parameter that produced a
        gradient has ``p.grad`` pointing at its slice of it."""
        prof = self._profile_begin() if self.profile else None
        if self.overlap:
            late = 0
            for b in self.buckets:           # buckets that a gradient-less parameter (or
                if not b.sent:               # no_sync) held back
                    self._send(b)
                    late += 1
            issued = [b.issued for b in self.buckets]
            for b in self.buckets:
                self._finish(b)
                b.reset()
            if prof is not None:
                self._profile_end(prof, issued, late)
        else:
            collect = getattr(self.optimizer, "collect_grads", None)
            if collect is not None:
                collect()
            all_reduce_flat([f.grad for f in self.optimizer.flat_groups], self.chunk,
                            self.async_op)
            if prof is not None:
                self._profile_end(prof, [], 0)
        if self.find_unused and (self.world > 1):
            self._attach_remote_grads()


    # ---- exchange profile ---------------------------------------------------------------------
    def _profile_begin(self):
        """End of backward as the MAIN stream sees it (an event) and as the host sees it."""
        dev = next((f.grad.device for f in self.optimizer.flat_groups if f.grad.numel()), None)
        ev = None
        if dev is not None and dev.type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))
        return dev, ev, time.perf_counter()

    def _profile_end(self, begin, issued, late):
        dev, ev0, t0 = begin
        ev1 = None
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(torch.cuda.current_stream(dev))
        self._prof.append((ev0, ev1, t0, time.perf_counter(), issued, late))

    def exchange_stats(self):
        """What the profiled steps saw (medians over steps; call after a device synchronisation):
        ``exposed_comm_ms`` -- how long ``all_reduce()`` held the main stream between the end of
        backward and the optimiser (HIP events on that stream: waits for bucket handles, late
        sends, the second-contribution fold), ``exposed_comm_host_ms`` -- the same interval on the
        host clock (a blocking backend such as gloo shows here), ``bucket_mb`` -- message sizes
        in issue order of the flat buffer, ``issue_before_backward_end_ms`` -- per bucket, how long
        before the end of backward its collective was issued on the device time line (negative:
        it went out from ``all_reduce()`` itself, nothing left to hide behind) and
        ``buckets_sent_late`` -- buckets no hook completed."""
        if not self._prof:
            return None
        exp_dev, exp_host, lead, late = [], [], [], []
        for ev0, ev1, t0, t1, issued, nlate in self._prof:
            exp_host.append(1e3 * (t1 - t0))
            late.append(nlate)
            if ev0 is not None and ev1 is not None:
                ev1.synchronize()
                exp_dev.append(ev0.elapsed_time(ev1))
            row = []
            for it in issued:
                if it is None:
                    row.append(None)
                elif it[0] is not None and ev0 is not None:
                    it[0].synchronize()
                    row.append(it[0].elapsed_time(ev0))
                else:
                    row.append(1e3 * (t0 - it[1]))
            lead.append(row)
        med = statistics.median
        nb = max((len(r) for r in lead), default=0)
        out = {"steps_profiled": len(self._prof),
               "overlap": bool(self.overlap),
               "exposed_comm_ms": None if not exp_dev else round(med(exp_dev), 4),
               "exposed_comm_host_ms": round(med(exp_host), 4),
               "bucket_mb": [round(4.0 * (b.e1 - b.e0) / 2 ** 20, 3) for b in self.buckets],
               "issue_before_backward_end_ms": [
                   (lambda v: None if not v else round(med(v), 4))(
                       [r[k] for r in lead if len(r) > k and r[k] is not None]) for k in range(nb)],
               "buckets_sent_late": med(late) if late else 0}
        if not self.overlap:
            out["bucket_mb"] = [round(4.0 * f.grad.numel() / 2 ** 20, 3)
                                for f in self.optimizer.flat_groups]
        return out

    def clear_profile(self):
        self._prof = []


def reduce_max(value, device):
    """MAX over ranks of a host scalar (bench timing contract)."""
    if world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

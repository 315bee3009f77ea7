"""Data-parallel plumbing: one process per GPU, RCCL over xGMI through
``torch.distributed`` (backend "nccl" is RCCL on ROCm; "gloo" in CPU tests).

The reference gets data parallelism from ``lightning.Trainer(strategy="ddp")``
(adell_mri/entrypoints/segmentation/train.py:799-819, utils/pl_utils.py:424-458):
bucketed sum-all-reduce of gradients divided by the world size. Here the
optimiser already owns ONE flat gradient buffer per parameter group
(``optim.FlatParameters``), so the exchange is an all-reduce of that buffer in a
few large chunks (xGMI rings are per-link bound: few large messages), and the
division by the world size is folded into the fused optimiser kernel
(``grad_scale``), not a separate pass.
"""
import os

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


class GradSync:
    """Sum-all-reduce the flat gradient buffers of a fused optimiser.

    ``chunk_mb`` bounds one collective's payload; config 2's whole gradient
    (33 MB) goes out as a single message.
    """

    def __init__(self, optimizer, chunk_mb=64, async_op=False):
        self.optimizer = optimizer
        self.chunk = int(chunk_mb * 1024 * 1024 // 4)
        self.async_op = async_op
        self.world = world_size()
        for g in optimizer.param_groups:
            g["grad_scale"] = 1.0 / self.world

    def broadcast_parameters(self, src=0, module=None):
        """Rank ``src``'s parameters everywhere; with ``module`` also the tensors that are
        not in the optimiser's flat buffers (frozen parameters, buffers), as DDP does at
        construction."""
        if self.world == 1:
            return
        for flat in self.optimizer.flat_groups:
            dist.broadcast(flat.data, src=src)
        if module is not None:
            flat_ids = {id(p) for f in self.optimizer.flat_groups for p in f.params}
            for t in list(module.parameters()) + list(module.buffers()):
                if id(t) not in flat_ids:
                    dist.broadcast(t.data, src=src)
        from . import ops

        ops._weights_changed()

    def all_reduce(self):
        collect = getattr(self.optimizer, "collect_grads", None)
        if collect is not None:
            collect()
        all_reduce_flat([f.grad for f in self.optimizer.flat_groups], self.chunk, self.async_op)


def reduce_max(value, device):
    """MAX over ranks of a host scalar (bench timing contract)."""
    if world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

"""N>1 path on CPU: world_size-2 gloo processes exercise the same GradSync /
all_reduce_flat / shard code the RCCL run uses (flat gradient buffer, chunked sum
all-reduce, 1/world folded into grad_scale, broadcast of rank-0 parameters)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from adell_mri_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Flat:
    def __init__(self, n, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.data = torch.randn(n, generator=g)
        self.grad = torch.randn(n, generator=g)


class _Opt:
    """Stand-in with the two attributes GradSync uses (the fused optimisers need a GPU)."""

    def __init__(self, rank):
        self.param_groups = [{"lr": 0.1}, {"lr": 0.2}]
        self._flats = [_Flat(1000, rank), _Flat(37, rank)]

    @property
    def flat_groups(self):
        return self._flats


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    opt = _Opt(rank)
    local = [f.grad.clone() for f in opt.flat_groups]
    sync = parallel.GradSync(opt, chunk_mb=0.001, overlap=False)  # 262 elements per chunk
    assert all(abs(g["grad_scale"] - 1.0 / world) < 1e-12 for g in opt.param_groups)
    sync.broadcast_parameters(src=0)
    sync.all_reduce()
    mx = parallel.reduce_max(float(rank + 1), torch.device("cpu"))
    torch.save({"grads": [f.grad for f in opt.flat_groups], "local": local,
                "data": [f.data for f in opt.flat_groups], "max": mx,
                "shard": parallel.shard_indices(7, rank, world)}, f"{out}/r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradsync_two_ranks_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    for gi in range(2):
        total = res[0]["local"][gi] + res[1]["local"][gi]
        for r in range(world):
            assert torch.allclose(res[r]["grads"][gi], total)
            assert torch.equal(res[r]["data"][gi], res[0]["data"][gi])  # rank-0 broadcast
    assert res[0]["max"] == res[1]["max"] == 2.0
    assert sorted(res[0]["shard"] + res[1]["shard"]) == list(range(7))


def test_single_process_is_a_noop():
    opt = _Opt(0)
    before = [f.grad.clone() for f in opt.flat_groups]
    s = parallel.GradSync(opt)
    s.broadcast_parameters()
    s.all_reduce()
    assert all(torch.equal(a, f.grad) for a, f in zip(before, opt.flat_groups))
    assert parallel.reduce_max(3.5, torch.device("cpu")) == 3.5


# ---- bucketed all-reduce issued from backward hooks (the overlap path) ---------------------------
class _CpuFlat:
    """FlatParameters' interface on CPU tensors (the real one gathers with a HIP launch): lets
    the hook / bucket / async-handle logic of GradSync run under gloo."""

    def __init__(self, params):
        self.params = list(params)
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.ends = self.offsets[1:] + [n]
        self.data, self.grad = torch.zeros(n), torch.zeros(n)
        for p, o in zip(self.params, self.offsets):
            v = self.data[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
        self.collected = []
        self.reduced = [False] * len(self.params)

    def slot(self, i):
        p, o = self.params[i], self.offsets[i]
        return self.grad[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:
            p.grad = None

    def collect(self, indices=None):
        idx = list(range(len(self.params)) if indices is None else indices)
        self.collected.append(idx)
        for i in idx:
            p = self.params[i]
            if p.grad is not None and p.grad.data_ptr() != self.slot(i).data_ptr():
                self.slot(i).copy_(p.grad)
                p.grad = self.slot(i)


class _CpuSGD:
    def __init__(self, params, lr, momentum, wd):
        self.param_groups = [dict(lr=lr, momentum=momentum, weight_decay=wd, grad_scale=1.0)]
        self._flat = _CpuFlat(params)
        self.buf = None

    @property
    def flat_groups(self):
        return [self._flat]

    def collect_grads(self):
        self._flat.collect()

    def step(self):
        g, f = self.param_groups[0], self._flat
        has = torch.zeros_like(f.grad, dtype=torch.bool)
        for p, o in zip(f.params, f.offsets):
            if p.grad is not None:
                has[o:o + p.numel()] = True
        d = f.grad * g["grad_scale"] + g["weight_decay"] * f.data
        self.buf = d.clone() if self.buf is None else g["momentum"] * self.buf + d
        f.data[has] -= g["lr"] * (d + g["momentum"] * self.buf)[has]


def _net():
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 6, 3, padding=1), torch.nn.SiLU(),
                              torch.nn.Conv2d(6, 6, 3, padding=1), torch.nn.SiLU(),
                              torch.nn.Conv2d(6, 1, 1))
    net.unused = torch.nn.Parameter(torch.ones(5))     # never receives a gradient
    return net


def _data():
    g = torch.Generator().manual_seed(5)
    return torch.randn(2, 2, 12, 12, generator=g), torch.randn(2, 1, 12, 12, generator=g)


def _overlap_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_distributed(backend="gloo")
    net = _net()
    if rank == 1:                       # ranks start different: the broadcast must fix that
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    opt = _CpuSGD(net.parameters(), 0.1, 0.9, 0.01)
    sync = parallel.GradSync(opt, n_buckets=3, min_bucket_elems=1)
    assert sync.overlap and len(sync.buckets) == 3
    sync.broadcast_parameters(module=net)
    x, y = _data()
    sent_during_backward = []
    for step in range(2):
        opt._flat.zero_grad()
        loss = ((net(x[rank:rank + 1]) - y[rank:rank + 1]) ** 2).mean()
        loss.backward()
        sent_during_backward.append(sum(b.sent for b in sync.buckets))
        sync.all_reduce()
        assert all(b.handle is None and not b.sent and b.pending == b.hi - b.lo
                   for b in sync.buckets)
        # every parameter that produced a gradient reads its reduced slice again
        f = opt._flat
        assert all(p.grad is None or p.grad.data_ptr() == f.slot(i).data_ptr()
                   for i, p in enumerate(f.params))
        assert sum(p.grad is not None for p in f.params) == len(f.params) - 1
        if step == 0:
            grads = opt._flat.grad.clone() * opt.param_groups[0]["grad_scale"]
        opt.step()
    torch.save({"grads": grads, "data": opt._flat.data.clone(), "sent": sent_during_backward,
                "buckets": [(b.lo, b.hi) for b in sync.buckets],
                "collected": opt._flat.collected}, f"{out}/o{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_overlap_equals_single_process_batch(tmp_path):
    """Two ranks, one item each, hooks + async bucket all-reduces == one process, batch of two
    (torch DDP semantics under train.py:799-819): averaged gradients and parameters after two
    momentum-SGD steps."""
    world, port = 2, _free_port()
    mp.spawn(_overlap_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"o{r}.pt") for r in range(world)]
    net = _net()
    opt = _CpuSGD(net.parameters(), 0.1, 0.9, 0.01)
    x, y = _data()
    for step in range(2):
        opt._flat.zero_grad()
        # mean over ranks of the per-rank mean loss == mean over the batch of two
        loss = ((net(x) - y) ** 2).mean()
        loss.backward()
        opt.collect_grads()
        if step == 0:
            ref_grads = opt._flat.grad.clone()
        opt.step()
    for r in range(world):
        assert torch.allclose(res[r]["grads"], ref_grads, rtol=1e-5, atol=1e-7)
        assert torch.allclose(res[r]["data"], opt._flat.data, rtol=1e-5, atol=1e-7)
        # the buckets whose parameters all had gradients went out DURING backward; the bucket
        # holding the gradient-less parameter waited for all_reduce()
        assert res[r]["sent"] == [2, 2], res[r]["sent"]
        # reverse execution order: the last bucket's parameters were gathered first
        first = res[r]["collected"][0]
        assert first == list(range(*res[r]["buckets"][-2])) or first == list(range(*res[r]["buckets"][-1]))
    assert torch.equal(res[0]["data"], res[1]["data"])
    # the unused parameter was left alone (no weight decay on a gradient-less parameter)
    assert torch.equal(net.unused.data, torch.ones(5))


def _accum_worker(rank, world, port, out, mode):
    """Two backward passes per step (gradient accumulation): ``plain`` = no
    precaution at all, ``no_sync`` = the first backward inside GradSync.no_sync(), ``abandon`` =
    a step given up after its backward (zero_grad without all_reduce), then a normal one."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_distributed(backend="gloo")
    net = _net()
    opt = _CpuSGD(net.parameters(), 0.1, 0.9, 0.01)
    opt.zero_grad = opt._flat.zero_grad          # what a step starts with (GradSync wraps it)
    sync = parallel.GradSync(opt, n_buckets=3, min_bucket_elems=1)
    sync.broadcast_parameters(module=net)
    x, y = _data()
    xs, ys = x[rank:rank + 1], y[rank:rank + 1]
    for step in range(2):
        opt.zero_grad()
        if mode == "abandon" and step == 0:
            ((net(xs) - ys) ** 2).mean().backward()
            assert any(b.sent for b in sync.buckets)
            continue                              # e.g. a NaN loss: no all_reduce, no step
        if mode == "no_sync":
            with sync.no_sync():
                ((net(xs) - ys) ** 2).mean().backward()
                assert not any(b.sent for b in sync.buckets)
        elif mode == "plain":
            ((net(xs) - ys) ** 2).mean().backward()
        (0.5 * (net(xs * 2.0) - ys) ** 2).mean().backward()
        sync.all_reduce()
        assert all(b.handle is None and not b.sent and not b.again for b in sync.buckets)
        opt.step()
    torch.save({"data": opt._flat.data.clone()}, f"{out}/a{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("mode", ["plain", "no_sync", "abandon"])
def test_two_backward_passes_per_step(tmp_path, mode):
    world, port = 2, _free_port()
    mp.spawn(_accum_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"a{r}.pt") for r in range(world)]
    net = _net()
    opt = _CpuSGD(net.parameters(), 0.1, 0.9, 0.01)
    x, y = _data()
    for step in range(2):
        opt._flat.zero_grad()
        if mode == "abandon" and step == 0:
            continue
        if mode != "abandon":
            ((net(x) - y) ** 2).mean().backward()
        (0.5 * (net(x * 2.0) - y) ** 2).mean().backward()
        opt.collect_grads()
        opt.step()
    for r in range(world):
        assert torch.allclose(res[r]["data"], opt._flat.data, rtol=1e-5, atol=1e-7), mode
    assert torch.equal(res[0]["data"], res[1]["data"])


def test_plan_buckets():
    assert parallel.plan_buckets([8, 108, 8, 324, 8, 8, 4], 3, 1) == [(0, 4), (4, 6), (6, 7)]
    for sizes, n, m in [([5] * 10, 3, 1), ([0, 0, 8], 3, 1), ([100, 200, 300], 4, 1 << 20),
                        ([7, 1, 1, 1, 90], 4, 2)]:
        b = parallel.plan_buckets(sizes, n, m)
        assert b[0][0] == 0 and b[-1][1] == len(sizes) and len(b) <= n
        assert all(a[1] == c[0] for a, c in zip(b, b[1:]))

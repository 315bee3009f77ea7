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
    sync = parallel.GradSync(opt, chunk_mb=0.001)  # 262 elements per chunk: several messages
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

"""Rank program of tests/test_zz_multirank_gpu.py::test_single_rank_rccl_overlap_path (not a test
module): ONE rank on the one card with backend "nccl" (= RCCL on ROCm). GradSync is forced onto its
overlapped path (post-accumulate-grad hooks -> multi-copy gather -> async all-reduce of the bucket
on ProcessGroupNCCL's stream -> wait before the optimiser), which a world of one otherwise skips:
this exercises RCCL initialisation, the stream ordering between autograd's stream, the gather
launch and the collective, and the handle waits -- everything except the wire. The parameters after
two steps must equal the no-exchange run bit for bit (a sum over one rank is the identity)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def steps(net, batch, n, **sync_kw):
    from adell_mri_amd.parallel import GradSync
    from adell_mri_amd.trainer import StepRunner

    opt = net.configure_optimizers()["optimizer"]
    sync = GradSync(opt, **sync_kw)
    runner = StepRunner(net, opt, sync)
    losses = [float(runner.train_step(batch)) for _ in range(n)]
    torch.cuda.synchronize()
    return sync, losses, {k: p.detach().cpu().clone() for k, p in net.named_parameters()}


def ddp_steps(device, batch, side_stream):
    """Two steps of the small U-Net wrapped in torch DistributedDataParallel (the reference's
    Lightning strategy="ddp" route) with the fused optimiser: DDP's Reducer copies every gradient
    into its bucket from a C++ hook on the main stream as soon as the node returns, so no weight
    gradient of a DDP-managed module may be produced on the side stream (functional._side_ok).
    Returns (losses, parameters, number of side_run calls)."""
    import ddp_worker
    from torch.nn.parallel import DistributedDataParallel as DDP

    from adell_mri_amd import functional as HF

    HF.FLAGS["wgrad_stream"] = side_stream
    calls = []
    real = HF.side_run
    HF.side_run = lambda fn, reads: calls.append(1) or real(fn, reads)
    try:
        net = ddp_worker.build(device)
        opt = net.configure_optimizers()["optimizer"]
        ddp = DDP(net, device_ids=[0])
        losses = []
        for _ in range(2):
            opt.zero_grad()
            out = ddp(batch["image"])
            prob = out[0] if isinstance(out, tuple) else out
            loss = ((prob - batch["mask"]) ** 2).mean()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
    finally:
        HF.side_run = real
        HF.FLAGS["wgrad_stream"] = True
    return losses, {k: p.detach().cpu().clone() for k, p in net.named_parameters()}, len(calls)


def main():
    import ddp_worker

    out = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    device = torch.device("cuda", 0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "unet3d_cfg2_small.npz"))
    batch = {"image": torch.from_numpy(g["x"]).to(device), "mask": torch.from_numpy(g["y"]).to(device)}
    sync, losses, params = steps(ddp_worker.build(device), batch, 2, overlap=True,
                                 _force_overlap=True, n_buckets=3, min_bucket_elems=1)
    assert sync.overlap and len(sync.buckets) == 3 and sync.world == 1
    # two backward passes in one step on the RCCL path as well (the late contribution is reduced
    # separately and added): gradient accumulation without no_sync()
    net2 = ddp_worker.build(device)
    opt2 = net2.configure_optimizers()["optimizer"]
    from adell_mri_amd.parallel import GradSync
    sync2 = GradSync(opt2, overlap=True, _force_overlap=True, n_buckets=2, min_bucket_elems=1)
    opt2.zero_grad()
    for _ in range(2):
        loss = net2.training_step(batch, 0)
        loss.backward()
    sync2.all_reduce()
    torch.cuda.synchronize()
    flat = opt2.flat_groups[0]
    twice = flat.grad.detach().cpu().clone()
    plain_sync, plain_losses, plain = steps(ddp_worker.build(device), batch, 2, overlap=False)
    assert not plain_sync.overlap
    # reference for the accumulation: one backward, gradient doubled
    net3 = ddp_worker.build(device)
    opt3 = net3.configure_optimizers()["optimizer"]
    opt3.zero_grad()
    net3.training_step(batch, 0).backward()
    opt3.collect_grads()
    once = opt3.flat_groups[0].grad.detach().cpu().clone()
    ddp_on = ddp_steps(device, batch, True)
    ddp_off = ddp_steps(device, batch, False)
    torch.save({"params": params, "plain": plain, "losses": losses, "plain_losses": plain_losses,
                "twice": twice, "once": once, "ddp_on": ddp_on, "ddp_off": ddp_off},
               os.path.join(out, "rccl.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

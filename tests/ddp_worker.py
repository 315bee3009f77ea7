"""Rank program of tests/test_zz_multirank_gpu.py::test_two_rank_unet_equals_single_process_batch
(not a test module): one item of the ``unet3d_cfg2_small`` fixture batch per rank, two training
steps of the real small U-Net through StepRunner + GradSync (bucketed all-reduce from backward
hooks), parameters and first-step gradients saved per rank."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(device):
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,
                                                           binary_generalized_dice_loss)
    from adell_mri_amd.modules.segmentation.pl import UNetPL
    from cases import UNET_CASES
    from oracle.weights import tensor_for

    kw = dict(UNET_CASES["unet3d_cfg2_small"])
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    kw["dropout_param"] = 0.0
    loss = CompoundLoss([(binary_generalized_dice_loss, {"smooth": 1e-5, "eps": 1e-6}),
                         (binary_focal_loss, {"gamma": 1.0, "eps": 1e-6})])
    net = UNetPL(image_key="image", label_key="mask", learning_rate=5e-2, weight_decay=5e-3,
                 loss_fn=loss, **kw)
    net.load_state_dict({k: torch.from_numpy(tensor_for(k, v.shape))
                         for k, v in net.state_dict().items()})
    return net.to(device).train()


def run(net, batch, steps, n_buckets=None):
    from adell_mri_amd.parallel import GradSync
    from adell_mri_amd.trainer import StepRunner

    opt = net.configure_optimizers()["optimizer"]
    sync = GradSync(opt, n_buckets=n_buckets, min_bucket_elems=1)
    runner = StepRunner(net, opt, sync)
    grads = None
    for s in range(steps):
        runner.train_step(batch)
        if s == 0:
            flat = opt.flat_groups[0]
            grads = (flat.grad * opt.param_groups[0]["grad_scale"]).detach().cpu().clone()
    torch.cuda.synchronize()
    return grads, {k: p.detach().cpu().clone() for k, p in net.named_parameters()}, sync


def main():
    from adell_mri_amd.parallel import init_distributed

    out = sys.argv[1]
    rank, world, _ = init_distributed()
    device = torch.device("cuda", 0)    # rehearsal: both ranks share the one card (gloo)
    torch.cuda.set_device(device)
    g = np.load(os.path.join(ROOT, "tests", "golden", "unet3d_cfg2_small.npz"))
    x = torch.from_numpy(g["x"])[rank:rank + 1].to(device)
    y = torch.from_numpy(g["y"])[rank:rank + 1].to(device)
    net = build(device)
    if rank == 1:   # the broadcast at StepRunner construction must undo this
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.5)
    grads, params, sync = run(net, {"image": x, "mask": y}, 2, n_buckets=3)
    assert sync.overlap and len(sync.buckets) == 3
    torch.save({"grads": grads, "params": params}, os.path.join(out, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

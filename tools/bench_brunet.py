#!/usr/bin/env python
"""Timing of the multi-branch U-Net (reference unet.py:846-1253): the
BASELINE config-2 network (depth [32,32,64,128,256], instance norm, swish, transposed-conv decoder,
residual links) with one encoder per MRI sequence -- two 1-channel 128^3 inputs instead of one
2-channel input -- merged by concurrent squeeze-and-excite gates. One step = forward + dice/focal
loss + backward + fused SGD-Nesterov through BrUNetPL.training_step. Not the headline bench."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--branches", type=int, default=2)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    from adell_mri_amd import ops
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,
                                                           binary_generalized_dice_loss)
    from adell_mri_amd.modules.segmentation.pl import BrUNetPL
    from adell_mri_amd.trainer import StepRunner

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    keys = [f"seq{i}" for i in range(args.branches)]
    loss_fn = CompoundLoss([(binary_generalized_dice_loss, {"smooth": 1e-5, "eps": 1e-6}),
                            (binary_focal_loss, {"gamma": 1.0, "eps": 1e-6})])
    net = BrUNetPL(image_keys=keys, label_key="mask", loss_fn=loss_fn, learning_rate=5e-4,
                   weight_decay=5e-3, spatial_dimensions=3, n_input_branches=args.branches,
                   conv_type="regular", link_type="residual", upscale_type="transpose",
                   norm_type="instance", padding=1, dropout_param=0.15,
                   activation_fn=activation_factory["swish"], in_channels=1, n_classes=2,
                   depth=[32, 32, 64, 128, 256], kernel_sizes=[3] * 5, strides=[2] * 5).to(dev)
    net.train()
    runner = StepRunner(net, net.configure_optimizers()["optimizer"])
    g = torch.Generator().manual_seed(1)
    S = args.size
    batch = {k: torch.rand((args.batch, 1, S, S, S), generator=g).to(dev) for k in keys}
    for k in keys:
        batch[k + "_weight"] = torch.ones(args.batch, device=dev)
    batch["mask"] = (torch.rand((args.batch, 1, S, S, S), generator=g) > 0.9).float().to(dev)
    for _ in range(args.warmup):
        runner.train_step(batch)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    ops.KERNEL_TIMER = ops.KernelTimer()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = runner.train_step(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    print(json.dumps({
        "workload": f"BrUNet {args.branches} x 1-channel {S}^3 branches, batch {args.batch}, "
                    "cfg-2 depths, dice+focal, SGD-Nesterov",
        "params": sum(p.numel() for p in net.parameters()),
        "ms_per_step": 1e3 * dt / args.steps, "volumes_per_s": args.batch * args.steps / dt,
        "loss": float(loss.detach().cpu()),
        "max_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30, "kernels": timer.summary()}))


if __name__ == "__main__":
    main()

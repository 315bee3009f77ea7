#!/usr/bin/env python
"""Timing of the SWIN-UNet training step at BASELINE configs[4] size (unet-swin.yaml:
256x256x128, 2 channels, patch 4^3, window 8^3, shifts [0,1], embedding 32/64/128/256,
8 heads, conv links, transposed-conv decoder). Not the headline bench (bench.py)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--size", type=str, default="256,256,128")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--layers", action="store_true", help="per (kernel, shape) table instead of the JSON line")
    args = ap.parse_args()
    from adell_mri_amd import ops
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.unetr import SWINUNet
    from adell_mri_amd.optim import FusedSGD
    from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,
                                                           binary_generalized_dice_loss)

    dev = torch.device("cuda", 0)
    size = [int(s) for s in args.size.split(",")]
    torch.manual_seed(0)
    net = SWINUNet(image_size=size, patch_size=[4, 4, 4], window_size=[8, 8, 8],
                   shift_sizes=[[0, 1]] * 4, embedding_size=[32, 64, 128, 256], n_heads=8,
                   dropout_rate=args.dropout, embed_method="convolutional", mlp_structure=4.0,
                   spatial_dimensions=3, conv_type="regular", link_type="conv",
                   upscale_type="transpose", norm_type="instance", padding="same",
                   dropout_param=0.0, activation_fn=activation_factory["leaky_relu"],
                   in_channels=2, n_classes=2, depth=[32, 64, 128, 256], kernel_sizes=[3] * 4,
                   strides=[[2, 2, 1], [2, 2, 1], 2, 2]).to(dev).train()
    loss_fn = CompoundLoss([(binary_generalized_dice_loss, {"smooth": 1e-5, "eps": 1e-6}),
                            (binary_focal_loss, {"gamma": 0.0, "eps": 1e-6})])
    opt = FusedSGD(net.parameters(), lr=5e-3, momentum=0.99, weight_decay=0.05, nesterov=True)
    g = torch.Generator().manual_seed(1)
    x = torch.rand((args.batch, 2, *size), generator=g).to(dev)
    y = (torch.rand((args.batch, 1, *size), generator=g) > 0.9).float().to(dev)

    def step():
        opt.zero_grad()
        prob, _ = net(x)
        loss = torch.stack([t.mean() for t in loss_fn(prob, y)]).mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ops.KERNEL_TIMER = ops.KernelTimer()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    if args.layers:
        tags = timer.by_tag()
        tot = sum(v["ms"] for v in tags.values())
        print(f"timed kernels {tot / args.steps:.2f} ms/step")
        for (name, tag), v in sorted(tags.items(), key=lambda kv: -kv[1]["ms"])[:45]:
            print(f"{v['ms'] / args.steps:7.3f} ms {100 * v['ms'] / tot:5.1f}% {v['tflops']:7.1f} TF "
                  f"x{v['launches'] // args.steps:3d}  {name.replace('adell_', '')}  {tag}")
        return
    print(json.dumps({"workload": f"SWIN-UNet {args.size} batch {args.batch}",
                      "params": sum(p.numel() for p in net.parameters()),
                      "ms_per_step": 1e3 * dt / args.steps,
                      "volumes_per_s": args.batch * args.steps / dt,
                      "loss": float(loss.detach()),
                      "max_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
                      "kernels": timer.summary()}))


if __name__ == "__main__":
    main()

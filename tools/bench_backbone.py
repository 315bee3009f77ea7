#!/usr/bin/env python
"""Timing of BASELINE config 2b: the config-2 U-Net with a ResNet backbone encoder
(sample_configs/ssl-resnet.yaml:5-6: [[64,64,5,2],[128,128,3,2],[256,256,3,2],[512,512,3,2]],
pools [[2,2,1],[2,2,1],[2,2,2],[2,2,2]]; 41.8 M parameters, 7.3 TFLOP forward per 128^3 volume,
SURVEY.md 8(a) row a12), assembled through the SSL -> U-Net hand-off. Not the headline bench."""
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
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    from adell_mri_amd import ops
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,
                                                           binary_generalized_dice_loss)
    from adell_mri_amd.modules.segmentation.unet import UNet
    from adell_mri_amd.optim import FusedSGD
    from adell_mri_amd.utils.handoff import unet_encoder_from_ssl

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ssl_cfg = dict(backbone_args=dict(
        spatial_dim=3, in_channels=2,
        structure=[[64, 64, 5, 2], [128, 128, 3, 2], [256, 256, 3, 2], [512, 512, 3, 2]],
        maxpool_structure=[[2, 2, 1], [2, 2, 1], [2, 2, 2], [2, 2, 2]], res_type="resnet",
        adn_fn=get_adn_fn(3, "batch", "swish", 0.0)),
        projection_head_args=dict(in_channels=512, structure=[1024, 512, 256],
                                  adn_fn=get_adn_fn(1, "batch", "swish", 0.0)))
    base = dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
                upscale_type="transpose", norm_type="instance", padding=1, dropout_param=0.1,
                activation_fn=activation_factory["swish"], in_channels=2, n_classes=2)
    cfg, enc, _ = unet_encoder_from_ssl(base, ssl_cfg)
    net = UNet(encoding_operations=enc[0], **cfg).to(dev).train()
    loss_fn = CompoundLoss([(binary_generalized_dice_loss, {"smooth": 1e-5, "eps": 1e-6}),
                            (binary_focal_loss, {"gamma": 1.0, "eps": 1e-6})])
    opt = FusedSGD(net.parameters(), lr=5e-4, momentum=0.99, weight_decay=5e-3, nesterov=True)
    g = torch.Generator().manual_seed(1)
    x = torch.rand((args.batch, 2, 128, 128, 128), generator=g).to(dev)
    y = (torch.rand((args.batch, 1, 128, 128, 128), generator=g) > 0.9).float().to(dev)

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
    ks = timer.summary()
    if os.environ.get("ADELL_BENCH_LAYERS"):
        tags = timer.by_tag()
        tot = sum(v["ms"] for v in tags.values())
        for (name, tag), v in sorted(tags.items(), key=lambda kv: -kv[1]["ms"])[:40]:
            print(f"{v['ms'] / args.steps:7.3f} ms {100 * v['ms'] / tot:5.1f}% {v['tflops']:7.1f} TF "
                  f"x{v['launches'] // args.steps:2d}  {name.replace('adell_', '')}  {tag}")
    print(json.dumps({"workload": f"U-Net + ResNet backbone (config 2b) 128^3 batch {args.batch}",
                      "params": sum(p.numel() for p in net.parameters()),
                      "depth": cfg["depth"], "strides": cfg["strides"],
                      "ms_per_step": 1e3 * dt / args.steps,
                      "volumes_per_s": args.batch * args.steps / dt, "loss": float(loss.detach()),
                      "kernels": {k: {"ms_per_step": v["ms"] / args.steps,
                                      "tflops": v.get("tflops", 0.0)} for k, v in ks.items()}}))


if __name__ == "__main__":
    main()

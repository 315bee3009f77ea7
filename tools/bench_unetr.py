#!/usr/bin/env python
"""Timing of the UNETR training step at BASELINE configs[2] size (unetr.yaml with
image_size 96^3, patch 16^3 as SURVEY.md 8(d) prescribes). Not the headline bench."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)       # unetr.yaml:25
    ap.add_argument("--dropout-rate", type=float, default=0.1)  # unetr.yaml:16
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", action="store_true", help="per (kernel, shape) table instead of the JSON line")
    ap.add_argument("--ab", default=None, help="functional.FLAGS name: alternate blocks of steps with the flag "
                                               "off / on in this one process and print both medians")
    args = ap.parse_args()
    from adell_mri_amd import ops
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,
                                                           binary_generalized_dice_loss)
    from adell_mri_amd.modules.segmentation.unetr import UNETR
    from adell_mri_amd.optim import FusedSGD

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    # sample_configs/unetr.yaml:1-31 with image 96^3 / patch 16^3 (BASELINE.json configs[2]);
    # attention_dim / hidden_dim default to embedding_size as the reference's parser leaves them
    kw = dict(image_size=[96, 96, 96], patch_size=[16, 16, 16], number_of_blocks=8,
              attention_dim=512, hidden_dim=512, embedding_size=512, n_heads=8,
              return_at=[2, 4, 6], mlp_structure=[1024], dropout_rate=args.dropout_rate,
              embed_method="linear", spatial_dimensions=3, conv_type="regular",
              link_type="residual", upscale_type="transpose", norm_type="instance", padding=1,
              dropout_param=0.0, activation_fn=activation_factory["leaky_relu"], in_channels=1,
              n_classes=2, depth=[16, 32, 64, 128], kernel_sizes=[3, 3, 3, 3])
    net = UNETR(**kw).to(dev).train()
    loss_fn = CompoundLoss([(binary_generalized_dice_loss, {"smooth": 1e-5, "eps": 1e-6}),
                            (binary_focal_loss, {"gamma": 0.0, "eps": 1e-6})])
    opt = FusedSGD(net.parameters(), lr=5e-3, momentum=0.99, weight_decay=5e-4, nesterov=True)
    g = torch.Generator().manual_seed(1)
    x = torch.rand((args.batch, 1, 96, 96, 96), generator=g).to(dev)
    y = (torch.rand((args.batch, 1, 96, 96, 96), generator=g) > 0.9).float().to(dev)

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
    if args.ab:
        import statistics
        from adell_mri_amd import functional as HF
        res = {False: [], True: []}
        for _ in range(6):
            for flag in (False, True):
                HF.FLAGS[args.ab] = flag
                step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                res[flag].append(1e3 * (time.perf_counter() - t0) / args.steps)
        HF.FLAGS[args.ab] = False
        print(json.dumps({"flag": args.ab, "ms_per_step_off": round(statistics.median(res[False]), 3),
                          "ms_per_step_on": round(statistics.median(res[True]), 3),
                          "blocks_off": [round(v, 2) for v in res[False]],
                          "blocks_on": [round(v, 2) for v in res[True]]}))
        return
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
        print(f"{1e3 * dt / args.steps:.2f} ms/step, timed kernels {tot / args.steps:.2f} ms/step")
        for (name, tag), v in sorted(tags.items(), key=lambda kv: -kv[1]["ms"])[:60]:
            print(f"{v['ms'] / args.steps:7.3f} ms {100 * v['ms'] / tot:5.1f}% {v['tflops']:7.1f} TF "
                  f"x{v['launches'] // args.steps:2d}  {name.replace('adell_', '')}  {tag}")
        return
    print(json.dumps({"workload": f"UNETR 96^3 patch 16 batch {args.batch}",
                      "params": sum(p.numel() for p in net.parameters()),
                      "ms_per_step": 1e3 * dt / args.steps,
                      "volumes_per_s": args.batch * args.steps / dt,
                      "loss": float(loss.detach()), "kernels": timer.summary()}))


if __name__ == "__main__":
    main()

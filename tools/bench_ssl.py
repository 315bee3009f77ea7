#!/usr/bin/env python
"""Timing of the VICReg / 3-D ConvNeXt training step (BASELINE configs[3] at full size:
structure of sample_configs/ssl-2d-convnext.yaml in 3-D, 64^3 crops). Not the headline
bench (bench.py); prints ms/step, crops/s and the per-kernel breakdown."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ema", action="store_true")
    ap.add_argument("--layers", action="store_true", help="per (kernel, shape) table instead of the JSON line")
    ap.add_argument("--ab", default=None, help="launch-plan switch of adell_set_tuning (or hf:<flag> of functional.FLAGS): "
                                               "alternate blocks of steps with it off / on in this process")
    args = ap.parse_args()
    from adell_mri_amd import ops
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.self_supervised.pl import SelfSLConvNeXtPL
    from adell_mri_amd.trainer import StepRunner
    from adell_mri_amd.utils import ExponentialMovingAverage

    dev = torch.device("cuda", 0)
    adn = get_adn_fn(1, "layer", "gelu", 0.0)
    torch.manual_seed(0)
    net = SelfSLConvNeXtPL(
        aug_image_key_1="a", aug_image_key_2="b", ssl_method="vicreg", stop_gradient=False,
        learning_rate=0.005, weight_decay=0.001, n_epochs=100, batch_size=args.batch,
        ema=None,
        backbone_args=dict(spatial_dim=3, in_channels=1,
                           structure=[[96, 384, 7, 3], [192, 768, 7, 3], [384, 1536, 7, 9],
                                      [768, 3072, 3, 3]],
                           maxpool_structure=[[2, 2, 2]] * 4),
        projection_head_args=dict(in_channels=768, structure=[1024, 2048, 1024], adn_fn=adn),
        prediction_head_args=dict(in_channels=1024, structure=[2048, 1024], adn_fn=adn)).to(dev)
    if args.ema:
        net.ema = ExponentialMovingAverage(0.99)
        net.ema.update(net)
    net.train()
    runner = StepRunner(net)
    g = torch.Generator().manual_seed(1)
    shape = (args.batch, 1, args.size, args.size, args.size)
    batch = {"a": torch.rand(shape, generator=g).to(dev), "b": torch.rand(shape, generator=g).to(dev)}
    for _ in range(args.warmup):
        runner.train_step(batch)
    torch.cuda.synchronize()
    if args.ab:
        import statistics
        from adell_mri_amd import _lib, functional as HF

        def set_switch(v):
            if args.ab.startswith("hf:"):
                HF.FLAGS[args.ab[3:]] = bool(v)
            else:
                _lib.lib().adell_set_tuning(args.ab.encode(), int(v))
            torch.cuda.synchronize()
            HF._ADN_PLAN.clear()

        res = {0: [], 1: []}
        for _ in range(6):
            for v in (0, 1):
                set_switch(v)
                runner.train_step(batch)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    runner.train_step(batch)
                torch.cuda.synchronize()
                res[v].append(1e3 * (time.perf_counter() - t0) / args.steps)
        set_switch(0)
        print(json.dumps({"switch": args.ab, "ms_per_step_off": round(statistics.median(res[0]), 3),
                          "ms_per_step_on": round(statistics.median(res[1]), 3),
                          "blocks_off": [round(v, 2) for v in res[0]],
                          "blocks_on": [round(v, 2) for v in res[1]]}))
        return
    ops.KERNEL_TIMER = ops.KernelTimer()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = runner.train_step(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    if args.layers:
        tags = timer.by_tag()
        tot = sum(v["ms"] for v in tags.values())
        print(f"{1e3 * dt / args.steps:.2f} ms/step, timed kernels {tot / args.steps:.2f} ms/step")
        for (name, tag), v in sorted(tags.items(), key=lambda kv: -kv[1]["ms"])[:40]:
            print(f"{v['ms'] / args.steps:7.3f} ms {100 * v['ms'] / tot:5.1f}% {v['tflops']:7.1f} TF "
                  f"x{v['launches'] // args.steps:3d}  {name.replace('adell_', '')}  {tag}")
        return
    print(json.dumps({"workload": f"VICReg ConvNeXt-3D {args.size}^3 batch {args.batch}",
                      "params": sum(p.numel() for p in net.parameters() if p.requires_grad),
                      "ms_per_step": 1e3 * dt / args.steps,
                      "crops_per_s": 2 * args.batch * args.steps / dt, "loss": float(loss),
                      "kernels": timer.summary()}))


if __name__ == "__main__":
    main()

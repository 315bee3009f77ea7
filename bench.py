#!/usr/bin/env python
"""Headline benchmark: training volumes/s of the 3-D U-Net of BASELINE config 2
(sample_configs/u-net-3d-resnet.yaml: regular-conv encoder, residual links,
instance norm, swish, transposed-conv decoder) on synthetic 2-channel 128^3
volumes; one step = forward + dice/focal loss + backward + SGD-Nesterov step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job
volumes/s, plus `roofline` for the dominant kernel (per-launch HIP-event timing
inside the timed region) and `cpu_baseline` (the torch-CPU oracle on a bounded
sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CFG2 = dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
            upscale_type="transpose", norm_type="instance", interpolation="bilinear", padding=1,
            dropout_param=float(os.environ.get("ADELL_BENCH_DROPOUT", "0.15")),  # config value 0.15
            in_channels=2, n_classes=2, depth=[32, 32, 64, 128, 256],
            kernel_sizes=[3] * 5, strides=[2] * 5)
LOSS = dict(smooth=1e-5, dice_eps=1e-6, gamma=1.0, focal_eps=1e-6)
LR, WD = 5e-4, 5e-3
# /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)" and "Peak BF16/FP16 MFMA ~2.5 PF dense".
# The f16x3 kernels execute 3 f16 MFMA FLOPs per algorithmic (fp32-equivalent) FLOP, so the
# algorithmic ceiling of that path is 2500 / 3.
FP32_MFMA_PEAK_TFLOPS = 157.3
F16_MFMA_PEAK_TFLOPS = 2500.0
F16X3_ALGORITHMIC_PEAK_TFLOPS = F16_MFMA_PEAK_TFLOPS / 3.0


def build_module(device, size):
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,
                                                           binary_generalized_dice_loss)
    from adell_mri_amd.modules.segmentation.pl import UNetPL

    loss = CompoundLoss([
        (binary_generalized_dice_loss, {"smooth": LOSS["smooth"], "eps": LOSS["dice_eps"]}),
        (binary_focal_loss, {"gamma": LOSS["gamma"], "eps": LOSS["focal_eps"]}),
    ])
    torch.manual_seed(0)
    net = UNetPL(image_key="image", label_key="mask", optimizer_str="sgd", learning_rate=LR,
                 weight_decay=WD, batch_size=1, n_epochs=100, loss_fn=loss,
                 activation_fn=activation_factory["swish"], **CFG2)
    return net.to(device)


def _shape(size):
    return (size, size, size) if isinstance(size, int) else tuple(size)


def synthetic_batch(batch, size, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand((batch, 2, *_shape(size)), generator=g)
    y = (torch.rand((batch, 1, *_shape(size)), generator=g) > 0.9).float()
    return {"image": x.to(device), "mask": y.to(device)}


def pmc_traffic(kernel):
    """Mean HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json, made by tools/pmc_traffic.py from separate
    ``--pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` runs of this script; gfx950 correction
    2*FETCH + WRITE as MI355X_MICROARCH.md prescribes). None when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            k = json.load(fh)["kernels"].get(kernel)
        return None if k is None else float(k["traffic_bytes_mean"])
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(size, threads):
    """One training step of the stock-torch CPU oracle (oracle/torch_ref) at size^3."""
    from oracle.torch_ref.unet import UNetOracle, compound_loss
    from oracle.weights import tensor_for
    from adell_mri_amd.modules.segmentation.unet import UNet

    torch.set_num_threads(threads)
    keys = {k: tuple(v.shape) for k, v in
            UNet(activation_fn=torch.nn.SiLU, **CFG2).state_dict().items()}
    sd = {k: torch.from_numpy(tensor_for(k, s)) for k, s in keys.items()}
    cfg = dict(depth=CFG2["depth"], kernel_sizes=CFG2["kernel_sizes"], strides=CFG2["strides"],
               padding=1, norm_type="instance", activation="swish", link_type="residual",
               n_classes=2, dropout_param=CFG2["dropout_param"])
    net = UNetOracle(sd, cfg).requires_grad_(True)
    net.training = True
    opt = torch.optim.SGD(net.parameters(), lr=LR, momentum=0.99, weight_decay=WD, nesterov=True)
    b = synthetic_batch(1, size, "cpu", 42)
    t0 = time.perf_counter()
    opt.zero_grad()
    prob = net.forward(b["image"], return_logits=False)
    loss = compound_loss(prob, b["mask"])
    loss.backward()
    opt.step()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--shape", type=str, default=None,
                    help="D,H,W of the synthetic volumes (e.g. 256,256,128); overrides --size")
    ap.add_argument("--batch", type=int, default=2,
                    help="volumes per GPU per step (u-net-3d-resnet.yaml:16 ships batch_size: 2; "
                         "SURVEY.md 8(d): B per GPU in {1, 2})")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-size", type=int, default=128,
                    help="edge of the one volume the CPU oracle is timed on (128: the workload's own "
                         "volume size, ~10-15 s on 16 threads)")
    args = ap.parse_args()

    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops
    from adell_mri_amd.parallel import GradSync, init_distributed, reduce_max
    from adell_mri_amd.trainer import StepRunner

    rank, world, local_rank = init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # ADELL_SINGLE_GPU_REHEARSAL=1 (with ADELL_DIST_BACKEND=gloo): all ranks on device 0, to run the
    # multi-rank code path on a one-GPU box; never set by the driver
    device = torch.device("cuda", 0 if os.environ.get("ADELL_SINGLE_GPU_REHEARSAL") else local_rank)
    torch.cuda.set_device(device)

    shape = _shape(args.size) if args.shape is None else tuple(int(v) for v in args.shape.split(","))
    shape_str = "x".join(str(v) for v in shape) if len(set(shape)) > 1 else f"{shape[0]}^3"
    net = build_module(device, shape)
    net.train()
    opt = net.configure_optimizers()["optimizer"]
    runner = StepRunner(net, opt, GradSync(opt))
    batch = synthetic_batch(args.batch, shape, device, 42 + rank)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # warm-up (untimed): every MFMA kernel family is event-timed to find the dominant one; the
    # timed region then carries events for that family only (fewer markers in the stream)
    ops.KERNEL_TIMER = ops.KernelTimer()
    for _ in range(args.warmup):
        runner.train_step(batch)
    barrier()
    warm, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    dom_warm = warm.dominant() if args.warmup > 0 else None
    warm_summary = warm.summary() if args.warmup > 0 else {}
    ops.KERNEL_TIMER = ops.KernelTimer(only=None if dom_warm is None else {dom_warm[0]})
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = runner.train_step(batch)
    barrier()
    dt = time.perf_counter() - t0
    timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    dt = reduce_max(dt, device)
    loss_value = float(loss.detach().cpu())

    if rank != 0:
        return
    vols = args.batch * world * args.steps
    out = {
        "metric": f"volumes/sec 3D U-Net {shape_str} 2-ch seg (train step: fwd+loss+bwd+SGD)",
        "value": vols / dt, "unit": "volumes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if HF.CONV_PRECISION == "fp32" else "f32 (f16x3 split MFMA, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: u-net-3d-resnet.yaml U-Net, 2x{shape_str}, "
                               f"batch {args.batch}/GPU, dice+focal, SGD-Nesterov",
                   "per_gpu_batch": args.batch, "size": list(shape), "parallelism": f"dp{world}"},
        "final_loss": loss_value,
    }
    dom = timer.dominant()
    if dom is not None:
        name, flops, ms, launches = dom
        achieved = flops / (ms * 1e-3) / 1e12
        f16 = "f16" in name
        peak = F16X3_ALGORITHMIC_PEAK_TFLOPS if f16 else FP32_MFMA_PEAK_TFLOPS
        out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": achieved,
                           "peak": peak, "unit": "TFLOP/s",
                           "frac": achieved / peak, "traffic": pmc_traffic(name),
                           "traffic_unit": "bytes per launch (mean; rocprofv3 PMC 2*FETCH_SIZE + "
                                           "WRITE_SIZE, profiles/r01_pmc_traffic.json)",
                           "algorithmic_bytes_per_launch":
                               timer.summary()[name]["algorithmic_bytes"] / launches,
                           "peak_basis": ("2.5 PFLOP/s dense f16 MFMA / 3 MFMAs per fp32 product"
                                          if f16 else "fp32 matrix peak"),
                           "executed_mfma_tflops": achieved * (3.0 if f16 else 1.0),
                           "fp32_mfma_peak": FP32_MFMA_PEAK_TFLOPS,
                           "launches": launches, "avg_launch_ms": ms / launches,
                           "kernel_time_share": timer.share(name, 1e3 * dt),
                           "all_kernels_warmup": warm_summary}
    if world == 1 and not args.no_cpu_baseline:
        threads = max(1, min(os.cpu_count() or 1, 16))
        t = cpu_baseline(args.cpu_size, threads)
        scale = args.cpu_size ** 3 / float(shape[0] * shape[1] * shape[2])
        out["cpu_baseline"] = {
            "value": scale / t, "unit": "volumes/s", "cores": threads, "kind": "port",
            "sample": f"1 training step (fwd + loss + bwd + SGD) of the stock-torch CPU oracle on one "
                      f"2-channel {args.cpu_size}^3 volume ({t:.2f} s)"
                      + ("" if scale == 1.0 else f", scaled by voxel count to {shape_str}")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

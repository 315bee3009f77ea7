#!/usr/bin/env python
"""Headline benchmark: training volumes/s of the 3-D U-Net of BASELINE config 2
(configs/u-net-3d-resnet.yaml = the reference's sample_configs/u-net-3d-resnet.yaml: regular-conv
encoder, residual links, instance norm, swish, transposed-conv decoder) on synthetic 2-channel
128^3 volumes; one step = forward + dice/focal loss + backward + SGD-Nesterov step. The module
is built the way the reference's entrypoint builds it: YAML -> ``parse_config_unet`` ->
``get_segmentation_network("unet", ...)``.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Started without torchrun and with --gpus N > 1, it launches the N ranks itself (child
``torch.distributed.run`` process, before anything touches the GPU) and relays rank 0's line.

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job volumes/s over the
timed K steps, plus

* ``roofline``: the dominant MFMA kernel family (per-launch HIP-event timing inside the timed
  region, on every ``--event-every``-th step: an event pair idles the stream ~6 us, so timing
  all ~85 launches of every step would cost the step it measures ~1 ms), with ``hbm`` = the
  dominant HBM-bound family (the fused norm/dropout/activation kernels; event-timed on 2 steps
  directly after the timed region) against 8 TB/s and ``step_frac`` = whole-step algorithmic
  FLOPs / step time / ceiling;
* ``cpu_baseline``: the stock-torch CPU oracle (rank 0, N = 1 only), 1 warm-up + 3 timed
  training steps on the host's physical cores, and a one-thread figure on a 64^3 volume;
* ``fp32_mfma``: the same step on the bit-exact fp32-MFMA kernels (secondary figure);
* ``secondary``: north_star's other workloads on the same module -- batch-1 128^3 and 256 x 256 x 128
  volumes (3 warm-up + 5 timed steps each, with their own ``roofline.frac``);
* ``median_ms_per_step``: median of per-step HIP-event times; ``value`` keeps
  the contract's definition (all K steps between two barriers).
"""
import argparse
import gc
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIG = os.path.join(ROOT, "configs", "u-net-3d-resnet.yaml")
KEYS = ["image", "image_1"]          # two image keys x in_channels 1 = the 2-channel input
# /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)" and "Peak BF16/FP16 MFMA ~2.5 PF dense".
# The f16x3 kernels execute 3 f16 MFMA FLOPs per algorithmic (fp32-equivalent) FLOP, so the
# algorithmic ceiling of that path is 2500 / 3.
FP32_MFMA_PEAK_TFLOPS = 157.3
F16_MFMA_PEAK_TFLOPS = 2500.0
F16X3_ALGORITHMIC_PEAK_TFLOPS = F16_MFMA_PEAK_TFLOPS / 3.0
HBM_PEAK_TBS = 8.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps (default 100: ~3.5 s of GPU work, long enough for a driver-side "
                         "utilisation sampler to see the GPU busy)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--shape", type=str, default=None,
                    help="D,H,W of the synthetic volumes (e.g. 256,256,128); overrides --size")
    ap.add_argument("--batch", type=int, default=None,
                    help="volumes per GPU per step (default: batch_size of the YAML = 2; "
                         "SURVEY.md 8(d): B per GPU in {1, 2})")
    ap.add_argument("--config", type=str, default=CONFIG)
    ap.add_argument("--event-every", type=int, default=20,
                    help="event-time the dominant kernel family on every M-th timed step "
                         "(1: every step; each event pair idles the stream ~6 us)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32-MFMA secondary figure")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the other north_star workloads (256x256x128 and batch-1 128^3)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip BASELINE configs 3-5 (UNETR 96^3, VICReg ConvNeXt 64^3, SWIN-UNet "
                         "256x256x128) in the `secondary` object")
    ap.add_argument("--cpu-size", type=int, default=128,
                    help="edge of the volume the CPU oracle is timed on with all cores")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N > 1 without torchrun: start the ranks as a child process (never re-exec this
    one) and pass rank 0's JSON line through."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
           str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def roofline_entry(name, flops, ms, launches, nbytes):
    """The roofline object of one kernel family: the bound is the side of the ridge its arithmetic
    intensity (algorithmic FLOPs / algorithmic bytes over the family's launches) falls on --
    f16x3 MFMA: 833 TFLOP/s / 8 TB/s = 104 FLOP/B, fp32 MFMA: 157.3 / 8 = 20 FLOP/B -- and
    `achieved` / `peak` / `frac` are quoted against THAT ceiling."""
    f16 = "f16" in name
    peak_tf = F16X3_ALGORITHMIC_PEAK_TFLOPS if f16 else FP32_MFMA_PEAK_TFLOPS
    ridge = peak_tf / HBM_PEAK_TBS                    # FLOP per byte
    tf = flops / (ms * 1e-3) / 1e12
    out = {"kernel": name, "launches": launches, "avg_launch_ms": ms / max(launches, 1),
           "arithmetic_intensity": None if not nbytes else flops / nbytes,
           "ridge_flop_per_byte": ridge, "mfma_tflops": tf, "mfma_frac": tf / peak_tf,
           "peak_basis": ("2.5 PFLOP/s dense f16 MFMA / 3 MFMAs per fp32 product" if f16
                          else "fp32 matrix peak")}
    if nbytes and flops / nbytes < ridge:
        tbs = nbytes / (ms * 1e-3) / 1e12
        out.update({"bound": "hbm", "achieved": tbs * 1e3, "peak": HBM_PEAK_TBS * 1e3,
                    "unit": "GB/s", "frac": tbs / HBM_PEAK_TBS})
    else:
        out.update({"bound": "mfma", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": tf / peak_tf})
    return out


def _shape(size):
    return (size, size, size) if isinstance(size, int) else tuple(size)


def build_module(device, config=CONFIG):
    """YAML -> parse_config_unet -> get_segmentation_network, as entrypoints/segmentation/
    train.py:736 does (no SSL backbone: encoding_operations = [None])."""
    import torch

    from adell_mri_amd.modules.config_parsing import parse_config_unet
    from adell_mri_amd.utils.network_factories import get_segmentation_network

    cfg, loss_keys = parse_config_unet(config, len(KEYS), 2)
    if "ADELL_BENCH_DROPOUT" in os.environ:     # profiling aid; the YAML ships 0.15
        cfg["dropout_param"] = float(os.environ["ADELL_BENCH_DROPOUT"])
    torch.manual_seed(0)
    net = get_segmentation_network(
        net_type="unet", network_config=cfg, bottleneck_classification=False,
        clinical_feature_keys=[], all_aux_keys=[], clinical_feature_params=None,
        clinical_feature_key_net=None, aux_key_net=None, max_epochs=100,
        encoding_operations=[None], picai_eval=False, lr_encoder=None, encoder_checkpoint=None,
        res_config_file=None, deep_supervision=False, n_classes=2, keys=KEYS,
        optimizer_str="sgd")
    return net.to(device), loss_keys


def build_cfg2b(device=None, config=CONFIG):
    """BASELINE config 2b: the same YAML with the ResNet of configs/ssl-resnet.yaml (the reference's
    sample_configs/ssl-resnet.yaml) as encoder -- `--res_config_file`, assembled as
    entrypoints/segmentation/train.py:672-734 does: parse_config_ssl -> ResNet -> depth / strides
    from the backbone, stem / stages / pools as encoding_operations (utils/handoff.py) ->
    get_segmentation_network. 41.8 M parameters, 7.3 TFLOP forward per 128^3 volume; the padded
    max-pools give odd 65^3 / 33x33x65 / 17x17x65 / 9x9x33 maps, so crop_to_size runs in every
    decoder level."""
    import torch

    from adell_mri_amd.modules.config_parsing import parse_config_ssl, parse_config_unet
    from adell_mri_amd.utils.handoff import unet_encoder_from_ssl
    from adell_mri_amd.utils.network_factories import get_segmentation_network

    res_config = os.path.join(ROOT, "configs", "ssl-resnet.yaml")
    cfg, loss_keys = parse_config_unet(config, len(KEYS), 2)
    _, cfg_ssl = parse_config_ssl(res_config, 0.0, len(KEYS))
    torch.manual_seed(0)
    cfg, enc, _ = unet_encoder_from_ssl(cfg, cfg_ssl)
    net = get_segmentation_network(
        net_type="unet", network_config=cfg, bottleneck_classification=False,
        clinical_feature_keys=[], all_aux_keys=[], clinical_feature_params=None,
        clinical_feature_key_net=None, aux_key_net=None, max_epochs=100,
        encoding_operations=enc, picai_eval=False, lr_encoder=None, encoder_checkpoint=None,
        res_config_file=res_config, deep_supervision=False, n_classes=2, keys=KEYS,
        optimizer_str="sgd")
    return (net if device is None else net.to(device)), loss_keys


def synthetic_batch(batch, size, device, seed):
    import torch

    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand((batch, 2, *_shape(size)), generator=g)
    y = (torch.rand((batch, 1, *_shape(size)), generator=g) > 0.9).float()
    return {"image": x.to(device), "mask": y.to(device)}


def pmc_traffic(kernel, workload):
    """Mean HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes
    (tools/pmc_traffic.py: separate ``--pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` runs of this
    script; gfx950 correction 2*FETCH + WRITE as MI355X_MICROARCH.md prescribes). Offline
    figure: returned only when the profile was taken on the same workload, else None."""
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if not (name.startswith("r") and name.endswith("_pmc_traffic.json")):
            continue
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                d = json.load(fh)
            if d.get("workload") != workload:
                continue
            k = d["kernels"].get(kernel)
            return None if k is None else (float(k["traffic_bytes_mean"]), name)
        except (OSError, ValueError, KeyError):
            continue
    return None


def host_cpu():
    """(model name, logical CPUs visible, physical cores usable by this process)."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    physical = min(len(cores) or usable, usable)
    # a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) bounds what the process can
    # really run in parallel: more threads than that only oversubscribe
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
            quota = None if q == "max" else float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, \
                    open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, period = float(fq.read()), float(fp.read())
                quota = None if q <= 0 else q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        physical = min(physical, max(1, int(quota)))
    return model, os.cpu_count() or 1, max(1, physical)


def cpu_training_steps(cfg_kwargs, size, threads, warmup, steps):
    """Median time of one training step (fwd + dice/focal + bwd + SGD-Nesterov) of the
    stock-torch CPU oracle (oracle/torch_ref) on one 2-channel size^3 volume."""
    import torch

    from adell_mri_amd.modules.segmentation.unet import UNet
    from oracle.torch_ref.unet import UNetOracle, compound_loss
    from oracle.weights import tensor_for

    torch.set_num_threads(threads)
    keys = {k: tuple(v.shape) for k, v in UNet(**cfg_kwargs).state_dict().items()}
    sd = {k: torch.from_numpy(tensor_for(k, s)) for k, s in keys.items()}
    cfg = dict(depth=cfg_kwargs["depth"], kernel_sizes=cfg_kwargs["kernel_sizes"],
               strides=cfg_kwargs["strides"], padding=cfg_kwargs["padding"],
               norm_type=cfg_kwargs["norm_type"], activation="swish",
               link_type=cfg_kwargs["link_type"], n_classes=2,
               dropout_param=cfg_kwargs["dropout_param"])
    net = UNetOracle(sd, cfg).requires_grad_(True)
    net.training = True
    opt = torch.optim.SGD(net.parameters(), lr=5e-4, momentum=0.99, weight_decay=5e-3,
                          nesterov=True)
    b = synthetic_batch(1, size, "cpu", 42)
    times = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        prob = net.forward(b["image"], return_logits=False)
        loss = compound_loss(prob, b["mask"])
        loss.backward()
        opt.step()
        if i >= warmup:
            times.append(time.perf_counter() - t0)
    return statistics.median(times), torch.get_num_threads()


# per-step deltas of the last timed_steps(per_step_events=True) call: (host s, device allocs, device
# frees, allocator retries, reserved bytes after the step, gc gen0 / gen1 / gen2 passes)
LAST_DIAG = []


def step_record(per, diag):
    """Self-describing per-step record of a timed region: every step's GPU time, where the slowest
    one sits and what the host saw during it."""
    if not per:
        return None
    srt = sorted(per)
    worst = max(range(len(per)), key=lambda i: per[i])
    rec = {"per_step_ms": [round(v, 2) for v in per], "max_ms": round(srt[-1], 2),
           "p95_ms": round(srt[min(len(srt) - 1, int(0.95 * len(srt)))], 2), "argmax": worst,
           "excess_over_median_ms": round(sum(per) - statistics.median(per) * len(per), 2)}
    if diag and len(diag) == len(per):
        rec["host_ms"] = [round(1e3 * d[0], 2) for d in diag]
        rec["device_allocs"] = [d[1] for d in diag]
        rec["device_frees"] = [d[2] for d in diag]
        rec["alloc_retries"] = [d[3] for d in diag]
        rec["reserved_GB"] = [round(d[4] / 2**30, 2) for d in diag]
        rec["gc_passes"] = [list(d[5:8]) for d in diag]
    return rec


def timed_steps(runner, batch, steps, barrier, per_step_events=True, timer=None, event_every=1):
    """K training steps between two barriers. ``timer``: the per-launch event timer of the
    dominant kernel family, switched on for every ``event_every``-th step only (each event pair
    idles the stream ~6 us; sampling keeps the instrument from slowing what it measures). The
    instrumented steps keep every launch on ONE stream (functional.FLAGS['wgrad_stream'] off): a
    launch duration taken while the weight-gradient stream shares the chip says nothing about the
    kernel -- so the timed region mixes overlapped steps with a few slower serial ones."""
    import torch

    from adell_mri_amd import functional as HF
    overlap = HF.FLAGS["wgrad_stream"]

    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step_events else None
    # per-step host-side record (cheap: the host runs ~15 ms ahead of the GPU): host time of the
    # step's enqueue, the allocator's device-allocation / retry counters, reserved bytes, and the
    # cyclic collector's pass counts -- what a stalled step is attributed with
    diag = [] if per_step_events else None

    def probe():
        st = torch.cuda.memory_stats()
        g = gc.get_stats()
        return (time.perf_counter(), st.get("num_device_alloc", 0), st.get("num_device_free", 0),
                st.get("num_alloc_retries", 0), st.get("reserved_bytes.all.current", 0),
                g[0]["collections"], g[1]["collections"], g[2]["collections"])

    t0 = time.perf_counter()
    loss = None
    for i in range(steps):
        if evs:
            evs[i].record()
            diag.append(probe())
        if timer is not None:
            timer.active = event_every > 0 and i % event_every == 0
            HF.FLAGS["wgrad_stream"] = overlap and not timer.active
        loss = runner.train_step(batch)
    HF.FLAGS["wgrad_stream"] = overlap
    if timer is not None:
        timer.active = True
    if evs:
        evs[steps].record()
        diag.append(probe())
    barrier()
    dt = time.perf_counter() - t0
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)] if evs else []
    if diag:
        LAST_DIAG[:] = [tuple(b[j] - a[j] if j != 4 else b[j] for j in range(8))
                        for a, b in zip(diag[:-1], diag[1:])]
    return dt, loss, per



CONFIGS_DIR = os.path.join(ROOT, "configs")


# tools/native_ops.py: a callable (key, runner, batch) -> entry run INSTEAD of the timed steps of a
# secondary workload (after its warm-up steps)
PROBE = None


def other_config_runs(device, rank, world, barrier, reduce_max):
    """BASELINE configs 3, 4 and 5 at the sizes SURVEY.md 8(d) gives them, each built from its YAML
    through the factory the entrypoint uses (as tests/test_fullsize_configs_gpu.py does): 3 warm-up
    steps, pool head-room, 5 timed steps with the per-step record, then one instrumented step for
    the dominant FLOP-carrying kernel family of THAT workload (its own roofline)."""
    import torch

    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops
    from adell_mri_amd.modules.config_parsing import parse_config_ssl, parse_config_unet
    from adell_mri_amd.parallel import GradSync
    from adell_mri_amd.trainer import StepRunner
    from adell_mri_amd.utils.network_factories import get_segmentation_network

    def seg(net_type, yaml_name, keys, size, patch=None):
        cfg, _ = parse_config_unet(os.path.join(CONFIGS_DIR, yaml_name), len(keys), 2)
        if patch is not None:
            cfg["patch_size"] = patch
        torch.manual_seed(0)
        return get_segmentation_network(net_type, cfg, False, [], [], None, None, None, 100, [None],
                                        False, None, None, None, False, 2, keys,
                                        random_crop_size=size)

    def seg_batch(n, c, size, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.rand((n, c, *size), generator=g).to(device)
        y = (torch.rand((n, 1, *size), generator=g) > 0.9).float().to(device)
        return {"image": x, "mask": y}

    def build_cfg2b_entry():
        net, _ = build_cfg2b()
        return net, seg_batch(1, 2, (128, 128, 128), 242 + rank), 1, "volumes/s", \
            "BASELINE config 2b: u-net-3d-resnet.yaml + --res_config_file ssl-resnet.yaml (ResNet-backbone " \
            "encoder, 41.8 M parameters, odd 65^3 .. 9x9x33 maps), UNetPL, 2x128^3, batch 1/GPU"

    def build_cfg3():
        net = seg("unetr", "unetr.yaml", ["image"], [96, 96, 96], patch=[16, 16, 16])
        return net, seg_batch(4, 1, (96, 96, 96), 342 + rank), 4, "volumes/s", \
            "BASELINE configs[2]: unetr.yaml UNETRPL, 1x96^3, patch 16^3, batch 4/GPU"

    def build_cfg4():
        from adell_mri_amd.modules.self_supervised.pl import SelfSLConvNeXtPL
        _, cfg = parse_config_ssl(os.path.join(CONFIGS_DIR, "ssl-3d-convnext.yaml"), 0.0, 1)
        cfg.pop("batch_size", None)
        cfg["vic_reg_loss_params"] = {}
        cfg["backbone_args"] = {k: v for k, v in cfg["backbone_args"].items() if k != "res_type"}
        torch.manual_seed(0)
        net = SelfSLConvNeXtPL(aug_image_key_1="augmented_image_1",
                               aug_image_key_2="augmented_image_2", ssl_method="vicreg",
                               stop_gradient=False, n_epochs=100, **cfg)
        g = torch.Generator().manual_seed(442 + rank)
        x1 = torch.randn((32, 1, 64, 64, 64), generator=g).to(device)
        x2 = (x1 + 0.3 * torch.randn(x1.shape, generator=g).to(device)).flip(2)
        return net, {"augmented_image_1": x1, "augmented_image_2": x2}, 64, "crops/s", \
            "BASELINE configs[3]: ssl-3d-convnext.yaml SelfSLConvNeXtPL (VICReg), 2 views x 32 crops of 64^3 per GPU, AdamW"

    def build_cfg5():
        net = seg("swin", "unet-swin.yaml", ["image", "image_1"], [256, 256, 128])
        return net, seg_batch(1, 2, (256, 256, 128), 542 + rank), 1, "volumes/s", \
            "BASELINE configs[4]: unet-swin.yaml SWINUNetPL, 2x256x256x128, batch 1/GPU"

    out = {}
    # config 3 is ~800 launches of 10-60 us per step: eager, the host's enqueue time is as long as the
    # step (round 4's driver record: host 24.9 ms of a 26.8 ms step). With ADELL_BENCH_GRAPH=1 the same
    # step is also replayed from ONE captured HIP graph after its eager timed steps
    # (StepRunner.enable_graph: forward + loss + backward + gradient gather; the optimiser launch stays
    # eager) and recorded as `hip_graph_replay`: 2 ms of host time per step, ~1 ms MORE GPU time than an
    # eager step the host keeps up with (21.3 vs 20.1-20.4 ms, profiles/r05a_bench_line.json) -- `value`
    # is the eager figure either way. NOT run by default: hipStreamEndCapture of a whole training step
    # segfaults inside the ROCm runtime on some boxes of this pool (config 4's step always; config 3's
    # replayed fine all morning and crashed on four boxes in a row in the afternoon, on the very commit
    # that had recorded it), and a segfault would take the headline line down with it.
    graphed = ({"cfg3_unetr_96"} if world == 1 and os.environ.get("ADELL_BENCH_GRAPH")
               and not os.environ.get("ADELL_BENCH_NO_GRAPH") else set())
    for key, build in (("cfg2b_resnet_backbone_128", build_cfg2b_entry),
                       ("cfg3_unetr_96", build_cfg3), ("cfg4_vicreg_convnext_64", build_cfg4),
                       ("cfg5_swinunet_256x256x128", build_cfg5)):
        only = os.environ.get("ADELL_BENCH_ONLY")       # debugging aid: a comma list of the keys to run
        if only and key not in only.split(","):
            continue
        try:
            net, batch, units, unit, workload = build()
            net = net.to(device).train()
            opt = net.configure_optimizers()["optimizer"]
            runner = StepRunner(net, opt, GradSync(opt))
            for _ in range(4):
                runner.train_step(batch)
            barrier()
            if PROBE is not None:
                out[key] = PROBE(key, runner, batch)
                del runner, opt, net, batch
                torch.cuda.empty_cache()
                continue
            runner.reserve_memory()
            dt, loss, per = timed_steps(runner, batch, 5, barrier)
            rec = step_record(per, list(LAST_DIAG))
            dt = reduce_max(dt, device)
            replay = None
            if key in graphed:
                runner.enable_graph(batch, warmup=1)
                barrier()
                gdt, _, gper = timed_steps(runner, batch, 5, barrier)
                grec = step_record(gper, list(LAST_DIAG))
                runner.disable_graph()
                barrier()
                replay = {"ms_per_step": 1e3 * gdt / 5, "median_ms_per_step": statistics.median(gper),
                          "value": units * world * 5 / gdt, "unit": unit,
                          "host_ms": grec.get("host_ms"), "per_step_ms": grec.get("per_step_ms"),
                          "captured": "zero_grad + training_step + backward + gradient gather; the "
                                      "optimiser launch stays eager"}
            ops.KERNEL_TIMER = ops.KernelTimer()
            overlap = HF.FLAGS["wgrad_stream"]
            HF.FLAGS["wgrad_stream"] = False       # (instrumented step: one stream, see timed_steps)
            runner.train_step(batch)
            barrier()
            HF.FLAGS["wgrad_stream"] = overlap
            timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
            entry = {"workload": workload, "value": units * world * 5 / dt, "unit": unit,
                     "ms_per_step": 1e3 * dt / 5, "median_ms_per_step": statistics.median(per),
                     "steps": 5, "warmup": 4, "final_loss": float(loss.detach().cpu()),
                     "step_record": rec,
                     "params": sum(p.numel() for p in net.parameters())}
            if replay is not None:
                entry["hip_graph_replay"] = replay
            dom = timer.dominant()
            if dom is not None:
                name, flops, ms, launches = dom
                summ = timer.summary()
                roof = roofline_entry(name, flops, ms, launches, summ[name]["algorithmic_bytes"])
                roof["kernel_time_share_of_step"] = ms / (1e3 * dt / 5)
                roof["families_ms_per_step"] = {k: round(v["ms"], 3) for k, v in summ.items()}
                entry["roofline"] = roof
            out[key] = entry
            del runner, opt, net, batch
        except Exception as exc:      # a secondary workload must not take the headline line down
            out[key] = {"error": f"{type(exc).__name__}: {exc}"[:400]}
        torch.cuda.empty_cache()
    return out


def cpu_cfg1_step(threads):
    """BASELINE configs[0]: the 2-D U-Net of the reference's testing/test_unet.py:63-72 (depth
    16 / 32 / 64, transposed upscaling, 140 748 parameters) on the stock-torch CPU oracle: median of
    3 training steps on a batch of 4 x 1 x 128 x 128, seconds per step."""
    import torch

    from oracle.torch_ref import unet2d as o2

    torch.set_num_threads(threads)
    return o2.training_step_seconds(batch=4, size=128, steps=3, warmup=1)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))

    import torch

    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops
    from adell_mri_amd.parallel import GradSync, init_distributed, reduce_max
    from adell_mri_amd.trainer import StepRunner

    rank, world, local_rank = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # ADELL_SINGLE_GPU_REHEARSAL=1 (with ADELL_DIST_BACKEND=gloo): all ranks on device 0, to run the
    # multi-rank code path on a one-GPU box; never set by the driver
    device = torch.device("cuda", 0 if os.environ.get("ADELL_SINGLE_GPU_REHEARSAL") else local_rank)
    torch.cuda.set_device(device)

    shape = _shape(args.size) if args.shape is None else tuple(int(v) for v in args.shape.split(","))
    shape_str = "x".join(str(v) for v in shape) if len(set(shape)) > 1 else f"{shape[0]}^3"
    net, loss_keys = build_module(device, args.config)
    net.train()
    per_gpu_batch = args.batch if args.batch is not None else int(net.batch_size)
    opt = net.configure_optimizers()["optimizer"]
    sync = GradSync(opt)
    runner = StepRunner(net, opt, sync)
    batch = synthetic_batch(per_gpu_batch, shape, device, 42 + rank)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # warm-up (untimed): every instrumented kernel family is event-timed to find the dominant
    # MFMA one; the timed region then carries events for that family and the HBM-bound
    # norm/activation family only (fewer markers in the stream)
    # (instrumented steps keep every launch on one stream, see timed_steps; a few plain steps
    # after them warm the weight-gradient stream and its memory pool)
    ops.KERNEL_TIMER = ops.KernelTimer()
    overlap = HF.FLAGS["wgrad_stream"]
    HF.FLAGS["wgrad_stream"] = False
    for _ in range(args.warmup):
        runner.train_step(batch)
    barrier()
    warm, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    HF.FLAGS["wgrad_stream"] = overlap
    warm_extra = min(3, args.warmup) if overlap else 0
    for _ in range(warm_extra):
        runner.train_step(batch)
    barrier()
    # no device allocation inside a timed region: the pools get their head-room now (round 3's
    # driver record had a 64 ms and a 112 ms step where the pools grew by 4.3 / 7 GB)
    reserved_extra = runner.reserve_memory() if args.warmup > 0 else (0, 0)
    dom_warm = warm.dominant() if args.warmup > 0 else None
    warm_summary = warm.summary() if args.warmup > 0 else {}
    ops.KERNEL_TIMER = ops.KernelTimer(only=None if dom_warm is None else {dom_warm[0]})
    every = max(1, args.event_every)
    sampled = len(range(0, args.steps, every))
    barrier()
    dt, loss, per_step = timed_steps(runner, batch, args.steps, barrier, timer=ops.KERNEL_TIMER,
                                     event_every=every)
    main_record = step_record(per_step, list(LAST_DIAG))
    timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    # the HBM-bound norm / dropout / activation family: its own instrumented steps right after the
    # timed region (clocks still in their loaded state), so that its ~140 event pairs per step do
    # not sit inside the region `value` is measured over
    hbm_steps = 2 if args.warmup > 0 else 0
    ops.KERNEL_TIMER = ops.KernelTimer(only={ops.NORM_ACT_FAMILY})
    overlap = HF.FLAGS["wgrad_stream"]
    HF.FLAGS["wgrad_stream"] = False          # (instrumented steps: one stream, see timed_steps)
    for _ in range(hbm_steps):
        runner.train_step(batch)
    barrier()
    hbm_timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    # the same step with every launch on one stream, for the record (a few steps)
    serial_ms = None
    if overlap and args.warmup > 0:
        nser = max(3, min(args.steps, 8))
        dts, _, _ = timed_steps(runner, batch, nser, barrier, per_step_events=False)
        serial_ms = 1e3 * reduce_max(dts, device) / nser
    HF.FLAGS["wgrad_stream"] = overlap
    # N > 1: what the gradient exchange costs the step, on a few profiled steps after the timed
    # region (GradSync.exchange_stats: exposed wait of the main stream, bucket sizes, how long before
    # the end of backward each bucket went out) -- the line that explains a scaling curve
    exchange = None
    if world > 1 and args.warmup > 0:
        sync.profile = True
        timed_steps(runner, batch, 3, barrier, per_step_events=False)
        sync.profile = False
        exchange = sync.exchange_stats()
        sync.clear_profile()
    dt = reduce_max(dt, device)
    loss_value = float(loss.detach().cpu())

    f16 = HF.CONV_PRECISION != "fp32"
    fp32_line = None
    if not args.no_fp32 and f16:
        # secondary figure: the same step on the bit-exact fp32-MFMA conv kernels
        HF.set_conv_precision("fp32")
        for _ in range(2):
            runner.train_step(batch)
        barrier()
        n32 = max(3, min(args.steps, 5))
        dt32, _, _ = timed_steps(runner, batch, n32, barrier, per_step_events=False)
        dt32 = reduce_max(dt32, device)
        HF.set_conv_precision("f16x3")
        fp32_line = {"value": per_gpu_batch * world * n32 / dt32, "unit": "volumes/s",
                     "ms_per_step": 1e3 * dt32 / n32, "steps": n32,
                     "dtype": "f32 (v_mfma_f32_32x32x2_f32, bit-exact fp32 FMA chains)"}

    # the other workloads north_star names, on the same module (every rank runs them: the gradient
    # exchange inside a step is collective): 256 x 256 x 128 volumes and batch-1 128^3, 3 warm-up +
    # 5 timed steps each, then two instrumented steps for the dominant kernel family's roofline
    secondary = {}
    if not args.no_secondary and args.shape is None and args.size == 128 and args.batch is None:
        # (the smaller workload first: it reuses the main run's memory pool; 3 warm-up steps, a new
        # shape's first steps grow the pool and repack nothing else)
        for key, sshape, sbatch in (("128^3_batch1", (128, 128, 128), 1),
                                    ("256x256x128_batch1", (256, 256, 128), 1)):
            sb = synthetic_batch(sbatch, sshape, device, 142 + rank)
            for _ in range(3):
                runner.train_step(sb)
            barrier()
            runner.reserve_memory()
            sdt, _, sper = timed_steps(runner, sb, 5, barrier)
            srec = step_record(sper, list(LAST_DIAG))
            sdt = reduce_max(sdt, device)
            # its roofline: two instrumented (one-stream) steps after the timed ones
            ops.KERNEL_TIMER = ops.KernelTimer(only=None if dom_warm is None else {dom_warm[0]})
            timed_steps(runner, sb, 2, barrier, per_step_events=False, timer=ops.KERNEL_TIMER,
                        event_every=1)
            stimer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
            sec = {"value": sbatch * world * 5 / sdt, "unit": "volumes/s",
                   "ms_per_step": 1e3 * sdt / 5, "median_ms_per_step": statistics.median(sper),
                   "step_record": srec,
                   "steps": 5, "warmup": 3, "per_gpu_batch": sbatch, "size": list(sshape)}
            sdom = stimer.dominant()
            if sdom is not None:
                sname, sflops, sms, slaunches = sdom
                sec["roofline"] = roofline_entry(sname, sflops, sms, slaunches,
                                                 stimer.summary()[sname]["algorithmic_bytes"])
            secondary[key] = sec
            del sb
        torch.cuda.empty_cache()
    other = {}
    if (not args.no_secondary and not args.no_other_configs and args.shape is None
            and args.size == 128 and args.batch is None and os.path.abspath(args.config) == CONFIG):
        # release the headline model's pools first: cfg 5 alone reserves ~45 GB
        del batch
        torch.cuda.empty_cache()
        other = other_config_runs(device, rank, world, barrier, reduce_max)
        secondary.update(other)

    if rank != 0:
        return
    vols = per_gpu_batch * world * args.steps
    yaml_name = os.path.basename(args.config)
    opt_str = getattr(net, "optimizer_str", "sgd")
    opt_desc = "SGD-Nesterov" if opt_str == "sgd" else opt_str
    baseline_cfg = "BASELINE configs[1]: " if os.path.abspath(args.config) == CONFIG else ""
    workload = (f"{baseline_cfg}{yaml_name} {type(net).__name__}, 2x{shape_str}, "
                f"batch {per_gpu_batch}/GPU, {'+'.join(loss_keys)}, {opt_desc}")
    out = {
        "metric": f"volumes/sec 3D U-Net {shape_str} 2-ch seg (train step: fwd+loss+bwd+SGD)",
        "value": vols / dt, "unit": "volumes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (f16x3 split MFMA, fp32 accumulate)" if f16 else "f32",
        "data": "synthetic",
        "config": {"workload": workload, "per_gpu_batch": per_gpu_batch, "size": list(shape),
                   "parallelism": f"dp{world}", "yaml": os.path.relpath(args.config, ROOT),
                   "built_by": "parse_config_unet -> get_segmentation_network('unet')",
                   "gradient_exchange": ("none (1 rank)" if world == 1 else
                                         f"{len(sync.buckets)} bucket all-reduces issued from "
                                         f"backward hooks" if sync.overlap else
                                         "one all-reduce after backward")},
        "final_loss": loss_value,
    }
    if per_step:
        med = statistics.median(per_step)
        out["median_ms_per_step"] = med
        out["value_median"] = per_gpu_batch * world / (med * 1e-3)
        out["step_record"] = main_record
    out["streams"] = {"weight_gradient_stream": bool(overlap),
                      "ms_per_step_one_stream": serial_ms,
                      "untimed_steps_after_warmup": warm_extra,
                      "pool_headroom_bytes": list(reserved_extra),
                      "note": "the weight-gradient kernels run on a second HIP stream beside the "
                              "backward-data chain; the event-timed steps (roofline) keep one stream"}
    if exchange is not None:
        exchange["gradient_mb"] = round(sum(exchange["bucket_mb"]), 3)
        exchange["exposed_share_of_step"] = (
            None if exchange["exposed_comm_ms"] is None
            else exchange["exposed_comm_ms"] / (1e3 * dt / args.steps))
        exchange["backend"] = torch.distributed.get_backend()
        exchange["measured"] = "3 profiled steps on rank 0 directly after the timed region"
        out["gradient_exchange"] = exchange
    if fp32_line is not None:
        out["fp32_mfma"] = fp32_line
    if secondary:
        out["secondary"] = secondary
    dom = timer.dominant()
    if dom is not None:
        name, flops, ms, launches = dom
        kf16 = "f16" in name
        peak = F16X3_ALGORITHMIC_PEAK_TFLOPS if kf16 else FP32_MFMA_PEAK_TFLOPS
        traffic = pmc_traffic(name, workload)
        roof = roofline_entry(name, flops, ms, launches, timer.summary()[name]["algorithmic_bytes"])
        achieved = roof["mfma_tflops"]
        roof.update({
            "traffic": None if traffic is None else traffic[0],
            "traffic_source": None if traffic is None else
            f"offline rocprofv3 PMC passes of this workload (profiles/{traffic[1]}): mean bytes "
            f"per launch, 2*FETCH_SIZE + WRITE_SIZE",
            "algorithmic_bytes_per_launch": timer.summary()[name]["algorithmic_bytes"] / launches,
            "executed_mfma_tflops": achieved * (3.0 if kf16 else 1.0),
            "fp32_mfma_peak": FP32_MFMA_PEAK_TFLOPS,
            "event_timed_steps": f"{sampled} of the {args.steps} timed steps (every "
                                 f"{every}{'st' if every == 1 else 'th'})",
            "kernel_time_share": timer.share(name, 1e3 * dt * sampled / args.steps)})
        # whole step: algorithmic FLOPs of every instrumented MFMA / conv family (from the
        # warm-up census, per step) over the measured step time, against the same ceiling
        if args.warmup > 0:
            step_flops = sum(v["flops"] for v in warm_summary.values()) / args.warmup
            roof["step_frac"] = step_flops / (dt / args.steps) / 1e12 / peak
            roof["step_algorithmic_tflop"] = step_flops / 1e12
        hb = hbm_timer.summary().get(ops.NORM_ACT_FAMILY) if hbm_steps else None
        if hb is not None and hb["ms"] > 0:
            tbs = hb["algorithmic_bytes"] / (hb["ms"] * 1e-3) / 1e12
            roof["hbm"] = {"bound": "hbm", "kernel": ops.NORM_ACT_FAMILY + " (fused norm -> dropout "
                           "-> activation, forward + backward)", "achieved": tbs * 1e3,
                           "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s", "frac": tbs / HBM_PEAK_TBS,
                           "launches": hb["launches"],
                           "ms_per_step": hb["ms"] / hbm_steps,
                           "time_share": hb["ms"] / hbm_steps / (1e3 * dt / args.steps),
                           "algorithmic_bytes_per_step": hb["algorithmic_bytes"] / hbm_steps,
                           "measured": f"{hbm_steps} instrumented steps directly after the timed "
                                       f"region"}
        roof["all_kernels_warmup"] = warm_summary
        out["roofline"] = roof
    if world == 1 and not args.no_cpu_baseline:
        from adell_mri_amd.modules.config_parsing import parse_config_unet

        cfg, _ = parse_config_unet(args.config, len(KEYS), 2)
        kw = {k: cfg[k] for k in ("spatial_dimensions", "conv_type", "link_type", "upscale_type",
                                  "norm_type", "interpolation", "padding", "dropout_param",
                                  "activation_fn", "in_channels", "depth", "kernel_sizes",
                                  "strides")}
        model, logical, physical = host_cpu()
        t_all, used = cpu_training_steps(kw, args.cpu_size, physical, 1, 3)
        small = min(64, args.cpu_size)
        t_one, _ = cpu_training_steps(kw, small, 1, 0, 1)
        vox = float(shape[0] * shape[1] * shape[2])
        out["cpu_baseline"] = {
            "value": (args.cpu_size ** 3 / vox) / t_all, "unit": "volumes/s", "cores": used,
            "kind": "port", "cpu_model": model, "os_cpu_count": logical,
            "torch_num_threads": used,
            "sample": f"stock-torch CPU oracle (oracle/torch_ref), one 2-channel {args.cpu_size}^3 "
                      f"volume per step: 1 warm-up + 3 timed training steps (fwd + dice/focal + bwd "
                      f"+ SGD-Nesterov), median {t_all:.2f} s/step on {used} threads"
                      + ("" if args.cpu_size ** 3 == vox else f", scaled by voxel count to {shape_str}"),
            "one_thread": {"value": (small ** 3 / vox) / t_one, "unit": "volumes/s", "cores": 1,
                           "sample": f"1 training step on one {small}^3 volume ({t_one:.2f} s), "
                                     f"scaled by voxel count to {shape_str}"}}
        # "... at cfg 2 (B = 1) and cfg 1" -- BASELINE configs[0], the reference's
        # own CPU-runnable case (testing/test_unet.py:63-72), on the same host cores
        t_cfg1 = cpu_cfg1_step(physical)
        out["cpu_baseline"]["cfg1"] = {
            "value": 4.0 / t_cfg1, "unit": "images/s", "cores": used, "kind": "port",
            "sample": f"BASELINE configs[0]: 2-D U-Net 128x128 1-ch (140 748 parameters, BatchNorm2d "
                      f"+ PReLU), stock-torch CPU oracle (oracle/torch_ref/unet2d.py, pinned to the "
                      f"reference fixture unet2d_cfg1): 1 warm-up + 3 timed training steps on a "
                      f"batch of 4, median {1e3 * t_cfg1:.1f} ms/step"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

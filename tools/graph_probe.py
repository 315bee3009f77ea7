"""Feasibility probe: capture one whole training step of BASELINE config 3 (UNETR 96^3, batch 4) in a
HIP graph (torch.cuda.CUDAGraph) and replay it; prints what broke or the eager / replayed step times."""
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd.modules.config_parsing import parse_config_unet  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402
from adell_mri_amd.utils.network_factories import get_segmentation_network  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("PROBE_UNET_FIRST"):
    unet, _ = bench.build_module(dev)
    unet.train()
    o = unet.configure_optimizers()["optimizer"]
    r = StepRunner(unet, o, GradSync(o))
    b = bench.synthetic_batch(2, (128, 128, 128), dev, 42)
    for _ in range(3):
        r.train_step(b)
    torch.cuda.synchronize()
    del unet, o, r, b
    torch.cuda.empty_cache()
cfg, _ = parse_config_unet(os.path.join(bench.CONFIGS_DIR, "unetr.yaml"), 1, 2)
cfg["patch_size"] = [16, 16, 16]
if os.environ.get("PROBE_NODROP"):
    cfg["dropout_rate"] = 0.0
torch.manual_seed(0)
net = get_segmentation_network("unetr", cfg, False, [], [], None, None, None, 100, [None], False, None,
                               None, None, False, 2, ["image"], random_crop_size=[96, 96, 96])
net = net.to(dev).train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
g = torch.Generator().manual_seed(342)
batch = {"image": torch.rand((4, 1, 96, 96, 96), generator=g).to(dev),
         "mask": (torch.rand((4, 1, 96, 96, 96), generator=g) > 0.9).float().to(dev)}


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


out = {}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(4):
        runner.train_step(batch)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
out["eager_ms"] = timed(lambda: runner.train_step(batch), 8)
if os.environ.get("PROBE_RESERVE"):
    runner.reserve_memory()
try:
    loss = runner.enable_graph(batch, warmup=1)
    out["captured"] = True
    out["replay_ms"] = timed(lambda: runner.train_step(batch), 8)
    out["loss_after_replays"] = float(loss)
except Exception as exc:      # noqa: BLE001
    out["captured"] = False
    out["error"] = f"{type(exc).__name__}: {exc}"[:600]
    traceback.print_exc()
print(json.dumps(out))

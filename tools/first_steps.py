"""Wall time of each of the first N training steps of a fresh process (memory-pool growth, stream
creation, first-use packing): python tools/first_steps.py [N=40] [shape=128,128,128] [batch=2]   (NO_STEP_SYNC=1: no per-step synchronisation)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
shape = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (128, 128, 128)
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
SYNC = os.environ.get("NO_STEP_SYNC") is None
dev = torch.device("cuda:0")
net, _ = bench.build_module(dev, bench.CONFIG)
net.train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
batch = bench.synthetic_batch(nb, shape, dev, 42)
ms, allocs = [], []
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
for i in range(n):
    if SYNC:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[i].record()
    runner.train_step(batch)
    if SYNC:
        torch.cuda.synchronize()
    ms.append(round((time.perf_counter() - t0) * 1e3, 1))
    allocs.append(torch.cuda.memory_stats().get("num_device_alloc"))
evs[n].record()
torch.cuda.synchronize()
if not SYNC:      # host run-ahead kept: GPU time per step from events
    ms = [round(evs[i].elapsed_time(evs[i + 1]), 1) for i in range(n)]
print(json.dumps({"ms": ms, "device_allocs": allocs, "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2),
                  "allocated_peak_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))

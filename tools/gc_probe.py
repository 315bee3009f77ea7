"""How long Python's cyclic garbage collector stops the host during training steps (the host runs
~15 ms ahead of the GPU: a longer stop idles the chip): every collection with its generation and
duration over N steps.  python tools/gc_probe.py [steps=120]"""
import gc
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda:0")
net, _ = bench.build_module(dev, bench.CONFIG)
net.train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
batch = bench.synthetic_batch(2, (128, 128, 128), dev, 42)
for _ in range(5):
    runner.train_step(batch)
torch.cuda.synchronize()
events, t0 = [], [0.0]


def cb(phase, info):
    if phase == "start":
        t0[0] = time.perf_counter()
    else:
        events.append((info["generation"], round((time.perf_counter() - t0[0]) * 1e3, 2), info["collected"]))


gc.callbacks.append(cb)
host = []
for _ in range(n):
    t = time.perf_counter()
    runner.train_step(batch)
    host.append(round((time.perf_counter() - t) * 1e3, 1))
torch.cuda.synchronize()
gc.callbacks.remove(cb)
by_gen = {}
for g, ms, _ in events:
    by_gen.setdefault(g, []).append(ms)
print(json.dumps({"objects": len(gc.get_objects()), "collections": {g: {"n": len(v), "max_ms": max(v), "sum_ms": round(sum(v), 1)}
                                                       for g, v in by_gen.items()},
                  "host_ms_per_step_median": sorted(host)[len(host) // 2], "host_ms_max": max(host)}))

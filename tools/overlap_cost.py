"""What the bucketed gradient exchange costs the bench step on ONE rank (backend nccl = RCCL,
world 1, GradSync forced onto its overlapped path: hooks -> gather -> async all-reduce per bucket):
everything of the N > 1 step except the wire, against the plain one-rank step, alternating.
    python tools/overlap_cost.py [rounds=4] [steps=12]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
net, _ = bench.build_module(dev, bench.CONFIG)
net.train()
opt = net.configure_optimizers()["optimizer"]
plain = GradSync(opt, overlap=False)
runner = StepRunner(net, opt, plain)
batch = bench.synthetic_batch(2, (128, 128, 128), dev, 42)


def timed(n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        runner.train_step(batch)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)


out = {"plain": [], "overlap": []}
for _ in range(6):
    runner.train_step(batch)
for r in range(rounds):
    out["plain"].append(timed(steps))
    plain.remove_hooks()
    runner.sync = GradSync(opt, overlap=True, _force_overlap=True)
    for _ in range(3):
        runner.train_step(batch)
    out["overlap"].append(timed(steps))
    out["buckets"] = len(runner.sync.buckets)
    runner.sync.remove_hooks()
    runner.sync = plain
    for _ in range(2):
        runner.train_step(batch)
print(json.dumps(out))
dist.destroy_process_group()

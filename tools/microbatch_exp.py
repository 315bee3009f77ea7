"""EXPERIMENT: the two volumes of a step as two micro-batches on two HIP streams (forward + backward
per volume, gradients accumulated, one optimiser step), so that one volume's HBM-bound norm / dropout /
activation passes run beside the other's MFMA-bound convolutions. Alternating blocks of steps against
the normal batch-2 step in one process; prints ms/step of both and the loss trajectories (the two
must agree: instance norm makes the volumes independent, the loss is a mean over volumes).
usage: microbatch_exp.py [rounds=4] [steps per block=8]"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")


def make():
    net, _ = bench.build_module(dev, bench.CONFIG)
    net.train()
    opt = net.configure_optimizers()["optimizer"]
    return net, opt, StepRunner(net, opt, GradSync(opt))


batch = bench.synthetic_batch(2, (128, 128, 128), dev, 42)
items = [{k: v[i:i + 1].contiguous() for k, v in batch.items()} for i in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def micro_step(net, opt, runner):
    main = torch.cuda.current_stream()
    opt.zero_grad(set_to_none=True)
    losses = []
    for s in streams:
        s.wait_stream(main)
    for it, s in zip(items, streams):          # both forwards first, then both backwards
        with torch.cuda.stream(s):
            losses.append(net.training_step(it, runner.step_idx))
    for l, s in zip(losses, streams):
        with torch.cuda.stream(s):
            (l * 0.5).backward()
    for s in streams:
        main.wait_stream(s)
    HF.join_side_stream()
    runner.sync.all_reduce()
    opt.step()
    runner.step_idx += 1
    return (losses[0].detach() + losses[1].detach()) * 0.5


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = [fn() for _ in range(n)]
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, [float(v) for v in out]


HF.FLAGS["no_adn_fuse"] = False
torch.manual_seed(0)
A = make()          # normal step
torch.manual_seed(0)
B = make()          # micro-batched step (same initial weights: same seed in build_module)
for _ in range(4):
    A[2].train_step(batch)
    micro_step(*B)
torch.cuda.synchronize()
res = {"batch2": [], "micro": []}
traj = {"batch2": [], "micro": []}
for r in range(rounds):
    t, l = timed(lambda: A[2].train_step(batch), steps)
    res["batch2"].append(round(t, 3)); traj["batch2"] += l
    t, l = timed(lambda: micro_step(*B), steps)
    res["micro"].append(round(t, 3)); traj["micro"] += l
print(json.dumps({"ms_per_step": {k: statistics.median(v) for k, v in res.items()}, "blocks": res,
                  "loss_first": {k: [round(x, 5) for x in v[:4]] for k, v in traj.items()},
                  "loss_last": {k: [round(x, 5) for x in v[-4:]] for k, v in traj.items()}}))

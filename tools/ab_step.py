"""A/B of one launch-plan switch on the bench workload, alternating blocks of steps inside ONE
process on ONE box (box-to-box and clock-ramp noise is +-0.3 ms of a 40 ms step, more than most
single changes). usage: ab_step.py switch [value] [rounds] [steps per block]   (switch: a launch-plan switch of
adell_set_tuning, or py:<name> for a dispatch flag of ops.FLAGS)"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import _lib  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

switch = sys.argv[1]
value = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda:0")
net, _ = bench.build_module(dev, bench.CONFIG)
net.train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
batch = bench.synthetic_batch(int(net.batch_size), (128, 128, 128), dev, 42)
PY = switch.startswith(("py:", "hf:"))   # "py:no_cinfold": a dispatch flag of ops.FLAGS; "hf:no_adn_fuse": of functional.FLAGS
if PY:
    from adell_mri_amd import functional as HF  # noqa: E402
    from adell_mri_amd import ops  # noqa: E402

    FLAGSET = HF.FLAGS if switch.startswith("hf:") else ops.FLAGS


def get_switch():
    return int(bool(FLAGSET[switch[3:]])) if PY else _lib.lib().adell_get_tuning(switch.encode())


def set_switch(v):
    if PY:
        FLAGSET[switch[3:]] = bool(v)
    else:
        _lib.lib().adell_set_tuning(switch.encode(), v)
    # launch plans cached on the Python side (partial-row counts of the fused site epilogue) belong
    # to the switch they were made under: a stale row count is an out-of-bounds write
    from adell_mri_amd import functional as _HF
    torch.cuda.synchronize()
    _HF._ADN_PLAN.clear()


base = get_switch()
for _ in range(6):
    runner.train_step(batch)
torch.cuda.synchronize()
res = {0: [], 1: []}
for r in range(rounds):
    for which in (0, 1):
        set_switch(value if which else base)
        runner.train_step(batch)          # settle (weights repacked, allocator)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            runner.train_step(batch)
        e1.record()
        torch.cuda.synchronize()
        res[which].append(e0.elapsed_time(e1) / steps)
set_switch(base)
print(json.dumps({"switch": switch, "base_value": base, "test_value": value,
                  "ms_per_step_base": round(statistics.median(res[0]), 3),
                  "ms_per_step_test": round(statistics.median(res[1]), 3),
                  "blocks_base": [round(v, 2) for v in res[0]],
                  "blocks_test": [round(v, 2) for v in res[1]]}))

"""Which tensor-library (at::native) launches the config-2b step still makes, by operator and
shape: torch.profiler over two training steps (python tools/cfg2b_native_ops.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import ops  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
net, _ = bench.build_cfg2b(dev)
net.train()
batch = bench.synthetic_batch(1, (128, 128, 128), dev, 1)
runner = StepRunner(net)
for _ in range(2):
    runner.train_step(batch)
torch.cuda.synchronize()

real = ops.ndhwc
seen = {}


def logged(x):
    xp = x.permute(0, 2, 3, 4, 1)
    if not xp.is_contiguous():
        k = (tuple(x.shape), tuple(x.stride()))
        seen[k] = seen.get(k, 0) + 1
    return real(x)


ops.ndhwc = logged
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    runner.train_step(batch)
    torch.cuda.synchronize()
ops.ndhwc = real
print("ndhwc copies:")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(" ", v, k)
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    if e.key.startswith("aten::") and t > 50:
        rows.append((t, e.key, e.count, str(e.input_shapes)[:110]))
for t, k, c, s in sorted(rows, reverse=True)[:40]:
    print(f"{t / 1e3:8.3f} ms  x{c:<4d} {k:32s} {s}")

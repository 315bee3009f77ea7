"""Which tensor-library (aten) launches a BASELINE config's training step still makes, by operator
and input shape, and which tensors reach ops.ndhwc in another layout (a copy each): torch.profiler
over one step after bench.py's warm-up.  ADELL_BENCH_ONLY=<key,...> python tools/native_ops.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import ops  # noqa: E402


def probe(key, runner, batch):
    real, seen = ops.ndhwc, {}

    def logged(x):
        if x.dim() == 5 and not x.permute(0, 2, 3, 4, 1).is_contiguous():
            k = (tuple(x.shape), tuple(x.stride()))
            seen[k] = seen.get(k, 0) + 1
        return real(x)

    ops.ndhwc = logged
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        runner.train_step(batch)
        torch.cuda.synchronize()
    ops.ndhwc = real
    print(f"==== {key}")
    for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
        print("  ndhwc copy x%d" % v, k)
    rows, total = [], 0.0
    for e in prof.key_averages(group_by_input_shape=True):
        t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
        if e.key.startswith("aten::") and t > 0:
            rows.append((t, e.key, e.count, str(e.input_shapes)[:100]))
            total += t
    print(f"  aten device time {total / 1e3:.3f} ms per step in {sum(r[2] for r in rows)} calls")
    for t, k, c, s in sorted(rows, reverse=True)[:25]:
        print(f"  {t / 1e3:8.3f} ms  x{c:<4d} {k:30s} {s}")
    return {"aten_ms": total / 1e3}


bench.PROBE = probe
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bench.other_config_runs(dev, 0, 1, torch.cuda.synchronize, lambda v, d: float(v))

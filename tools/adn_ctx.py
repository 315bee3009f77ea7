"""Why do the norm->dropout->activation kernels run at half their isolated rate inside the step?
Times the 2 x 128^3 x 32 forward kernel (a) back to back on one buffer pair, (b) rotating over
fresh buffer pairs (no cache / TLB reuse), (c) directly after a 40 ms burst of f16x3 conv launches
(clock / power state), (d) interleaved conv, adn, conv, adn as the step does."""
import json
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
shape = (2, 32, 128, 128, 128)
n, c = 2, 32
mean = torch.zeros(n, c, device=dev)
rstd = torch.ones(n, c, device=dev)
pairs = [ops.ndhwc(torch.randn(*shape, device=dev)) for _ in range(12)]
nbytes = pairs[0].numel() * 4
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
b = torch.randn(32, device=dev)
wp = ops.pack_weight_f16x3(w, 0)


def adn(i):
    return ops.norm_act_fwd(pairs[i % len(pairs)], mean, rstd, "swish", drop_p=0.15, seed=1)


def conv(i):
    return ops.conv3d_fwd(pairs[i % len(pairs)], wp, b, 32, 3, 1, 1, want_stats=True)


def ev():
    return torch.cuda.Event(enable_timing=True)


def report(name, us):
    us = sorted(us)
    med = us[len(us) // 2]
    print(json.dumps({"case": name, "median_us": round(med, 1), "min_us": round(us[0], 1),
                      "max_us": round(us[-1], 1), "TBps": round(2 * nbytes / med / 1e6, 2)}))


for _ in range(3):
    adn(0); conv(0)
torch.cuda.synchronize()

# (a) same buffer, each launch timed on its own
us = []
for i in range(20):
    e0, e1 = ev(), ev(); e0.record(); adn(0); e1.record(); torch.cuda.synchronize()
    us.append(e0.elapsed_time(e1) * 1e3)
report("same buffer", us)
# (b) rotating buffers
us = []
for i in range(24):
    e0, e1 = ev(), ev(); e0.record(); adn(i); e1.record(); torch.cuda.synchronize()
    us.append(e0.elapsed_time(e1) * 1e3)
report("rotating buffers", us)
# (c) after a conv burst
us = []
for i in range(10):
    for k in range(14):
        conv(k)
    e0, e1 = ev(), ev(); e0.record(); adn(i); e1.record(); torch.cuda.synchronize()
    us.append(e0.elapsed_time(e1) * 1e3)
report("after 14-conv burst", us)
# (d) interleaved, no host sync between
evs = []
for i in range(40):
    y = conv(i)
    e0, e1 = ev(), ev(); e0.record()
    ops.norm_act_fwd(y[0] if isinstance(y, tuple) else y, mean, rstd, "swish", drop_p=0.15, seed=1)
    e1.record(); evs.append((e0, e1))
torch.cuda.synchronize()
report("interleaved conv/adn on the conv output", [a.elapsed_time(b_) * 1e3 for a, b_ in evs[5:]])
# (e) conv timing in the same loop for reference
evs = []
for i in range(40):
    e0, e1 = ev(), ev(); e0.record(); y = conv(i); e1.record(); evs.append((e0, e1))
    adn(i)
torch.cuda.synchronize()
cu = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in evs[5:])
print(json.dumps({"case": "conv 32->32 in the interleaved loop", "median_us": round(cu[len(cu) // 2], 1)}))

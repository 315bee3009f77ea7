"""Bandwidth of the fused norm/dropout/activation kernels on one 32 x 128^3 activation,
feature by feature (what costs what)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import ops
dev = torch.device("cuda:0")
C, S = 32, 128
x = ops.ndhwc(torch.randn(1, C, S, S, S, device=dev))
g = ops.ndhwc(torch.randn(1, C, S, S, S, device=dev))
mean = torch.randn(1, C, device=dev) * 0.1
rstd = torch.rand(1, C, device=dev) + 0.5
nbytes = x.numel() * 4
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x)); print(f"torch copy                     : {2*nbytes/t/1e12:5.2f} TB/s  {t*1e6:7.1f} us")
for name, kw in [("identity, no norm", dict(mean=None, rstd=None, act="identity")),
                 ("norm only", dict(mean=mean, rstd=rstd, act="identity")),
                 ("norm + relu", dict(mean=mean, rstd=rstd, act="relu")),
                 ("norm + swish", dict(mean=mean, rstd=rstd, act="swish")),
                 ("norm + dropout + swish", dict(mean=mean, rstd=rstd, act="swish", drop_p=0.15, seed=1)),
                 ("norm + gelu", dict(mean=mean, rstd=rstd, act="gelu"))]:
    t = timeit(lambda: ops.norm_act_fwd(x, **kw))
    tb = timeit(lambda: ops.norm_act_bwd(x, g, **kw))
    print(f"fwd {name:27s}: {2*nbytes/t/1e12:5.2f} TB/s  {t*1e6:7.1f} us | bwd (partials+apply, 5 passes): {5*nbytes/tb/1e12:5.2f} TB/s {tb*1e6:7.1f} us")

# the same with 8 distinct inputs in rotation (2.1 GB > the 256 MB Infinity Cache): cold reads
xs = [ops.ndhwc(torch.randn(1, C, S, S, S, device=dev)) for _ in range(8)]
ys = [torch.empty_like(x) for _ in range(8)]
state = {"i": 0}
def rot_copy():
    i = state["i"] = (state["i"] + 1) % 8
    ys[i].copy_(xs[i])
def rot_fwd():
    i = state["i"] = (state["i"] + 1) % 8
    ops.norm_act_fwd(xs[i], mean=mean, rstd=rstd, act="swish", drop_p=0.15, seed=1)
def rot_bwd():
    i = state["i"] = (state["i"] + 1) % 8
    ops.norm_act_bwd(xs[i], xs[(i + 3) % 8], mean=mean, rstd=rstd, act="swish", drop_p=0.15, seed=1)
t = timeit(rot_copy, 16); print(f"rotating torch copy            : {2*nbytes/t/1e12:5.2f} TB/s  {t*1e6:7.1f} us")
t = timeit(rot_fwd, 16); print(f"rotating fwd norm+drop+swish   : {2*nbytes/t/1e12:5.2f} TB/s  {t*1e6:7.1f} us")
t = timeit(rot_bwd, 16); print(f"rotating bwd (5 passes)        : {5*nbytes/t/1e12:5.2f} TB/s  {t*1e6:7.1f} us")

"""Forward norm / activation pass at the activation shapes of configs 2 and 3, inputs in rotation
(cold reads): is the 16-channel UNETR tensor slower per byte than the 32-channel U-Net one?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=16):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for N, C, S, act in [(2, 32, 128, "swish"), (4, 16, 96, "leaky_relu"), (4, 16, 96, "swish"), (4, 32, 48, "leaky_relu"),
                     (2, 16, 128, "leaky_relu"), (4, 32, 96, "leaky_relu")]:
    xs = [ops.ndhwc(torch.randn(N, C, S, S, S, device=dev)) for _ in range(6)]
    mean = torch.randn(N, C, device=dev) * 0.1
    rstd = torch.rand(N, C, device=dev) + 0.5
    nbytes = xs[0].numel() * 4
    st = {"i": 0}
    def fwd():
        i = st["i"] = (st["i"] + 1) % 6
        ops.norm_act_fwd(xs[i], mean=mean, rstd=rstd, act=act, act_p=0.01, stats_per_item=1)
    t = timeit(fwd)
    print(f"[{N},{C},{S}^3] {act:10s}: {2 * nbytes / t / 1e12:5.2f} TB/s {t * 1e6:7.1f} us  ({nbytes / 1e6:.0f} MB)")
    del xs

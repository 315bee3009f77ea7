"""Where the norm -> dropout -> activation kernels spend their time: the 2 x 128^3 x 32 layer of the
bench workload with / without dropout and with swish / relu, beside a plain device copy of the
same bytes (the practical HBM ceiling of this box). Prints one JSON line per case."""
import json
import sys

import torch

sys.path.insert(0, ".")
from adell_mri_amd import ops  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    dev = torch.device("cuda:0")
    shapes = [(2, 32, 128, 128, 128), (2, 64, 64, 64, 64), (2, 128, 32, 32, 32)]
    for shape in shapes:
        n, c = shape[0], shape[1]
        x = ops.ndhwc(torch.randn(*shape, device=dev))
        g = ops.ndhwc(torch.randn(*shape, device=dev))
        y = torch.empty_like(x)
        mean = torch.zeros(n, c, device=dev)
        rstd = torch.ones(n, c, device=dev)
        nbytes = x.numel() * 4
        t = timed(lambda: y.copy_(x))
        print(json.dumps({"shape": shape, "case": "copy", "us": round(t, 1),
                          "TBps": round(2 * nbytes / t / 1e6, 2)}))
        for act in ("swish", "relu"):
            for p in (0.0, 0.15):
                tf = timed(lambda: ops.norm_act_fwd(x, mean, rstd, act, drop_p=p, seed=1))
                tb = timed(lambda: ops.norm_act_bwd(x, g, mean, rstd, act, drop_p=p, seed=1))
                print(json.dumps({"shape": shape, "case": f"{act} p={p}", "fwd_us": round(tf, 1),
                                  "fwd_TBps": round(2 * nbytes / tf / 1e6, 2),
                                  "bwd_us": round(tb, 1),
                                  "bwd_TBps": round(5 * nbytes / tb / 1e6, 2)}))


if __name__ == "__main__":
    main()

"""dW of the 7^3 stem conv (2 -> 64 at 128^3, res_net.py:55) by the vector-ALU kernel and through the
x-tap fold (16 -> 64, k = 7 x 7 x 1 on the f16x3 weight-gradient kernel): values and time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adell_mri_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
Cin, Cout, K = 2, 64, 7
size = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 128, 128)
g = torch.Generator(device=dev).manual_seed(0)
x = ops.ndhwc(torch.randn(1, Cin, *size, device=dev, generator=g))
dy = ops.ndhwc(torch.randn(1, Cout, *size, device=dev, generator=g))


def direct():
    return ops.conv3d_bwd_weight(x, dy, K, 1, K // 2, want_db=True, f16x3=True)


def folded():
    xf = ops.fold_x_taps(x, K, K // 2)
    dwf, db = ops.conv3d_bwd_weight(xf, dy, (K, K, 1), 1, (K // 2, K // 2, 0), want_db=True, f16x3=True)
    dw = dwf[:, :K * Cin, :, :, 0].reshape(Cout, K, Cin, K, K).permute(0, 2, 3, 4, 1).contiguous()
    return dw, db


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


a, da = direct()
b, db = folded()
print("max |a - b| / max |a|:", float((a - b).abs().max() / a.abs().max()), float((da - db).abs().max() / da.abs().max()))
print("direct %.1f us   folded %.1f us" % (timed(direct), timed(folded)))

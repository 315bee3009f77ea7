"""Which ADN sites of the bench step write split rows and which convs read them (one step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import ops  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

dev = torch.device("cuda:0")
net, _ = bench.build_module(dev, bench.CONFIG)
net.train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
batch = bench.synthetic_batch(2, (128, 128, 128), dev, 42)
runner.train_step(batch)
log = []
r_na, r_cf, r_wg = ops.norm_act_fwd, ops.conv3d_fwd, ops.conv3d_bwd_weight


def na(x, *a, **k):
    log.append(("adn", tuple(x.shape), k.get("split_exp")))
    return r_na(x, *a, **k)


def cf(x0, wp, bias, Cout, kernel, stride, padding, x1=None, **k):
    log.append(("conv", tuple(x0.shape), None if x1 is None else x1.shape[1], Cout, tuple(kernel), tuple(stride),
                k.get("rows0") is not None, k.get("rows1") is not None))
    return r_cf(x0, wp, bias, Cout, kernel, stride, padding, x1=x1, **k)


def wg(x0, dy, kernel, stride, padding, x1=None, **k):
    log.append(("wgrad", tuple(x0.shape), None if x1 is None else x1.shape[1], dy.shape[1],
                k.get("rows0") is not None, k.get("rows1") is not None))
    return r_wg(x0, dy, kernel, stride, padding, x1=x1, **k)


ops.norm_act_fwd, ops.conv3d_fwd, ops.conv3d_bwd_weight = na, cf, wg
before = ops.ROWS_FALLBACKS[0]
runner.train_step(batch)
torch.cuda.synchronize()
for e in log:
    print(e)
print("fallbacks", ops.ROWS_FALLBACKS[0] - before)

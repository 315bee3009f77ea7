"""How well conditioned are the gradients of the 2-D SWIN-UNet fixture? The same step twice, the
second time with the input perturbed at the fp32 rounding level (relative 1e-7): largest relative
gradient change per parameter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_swin as T
cuda = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "swinunet2d_small"
g = np.load(os.path.join(T.GOLD, name + ".npz"))
net = T.build(name).to(cuda).eval()
x0 = torch.from_numpy(g["x"]).to(cuda)
y = torch.from_numpy(g["y"]).to(cuda)
def grads(x):
    for p in net.parameters():
        p.grad = None
    prob, _ = net(x)
    T.compound_loss(prob, y).backward()
    return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
a = grads(x0)
torch.manual_seed(0)
if len(sys.argv) > 2 and sys.argv[2] == "ln":
    # perturb every LayerNorm-rows OUTPUT at the rounding level instead (what another summation
    # order inside the kernel does)
    from adell_mri_amd import ops
    of = ops.layernorm_rows_fwd
    def f(*args):
        y, m, r = of(*args)
        return y + 1.2e-7 * torch.randn_like(y) * (torch.rand_like(y) < 0.3), m, r
    ops.layernorm_rows_fwd = f
    b = grads(x0)
else:
    b = grads(x0 * (1 + 1e-7 * torch.randn_like(x0)))
def scale(k):          # the yardstick of tests/test_swin.py
    s = np.abs(g["grad64:" + k]).max()
    if k.endswith(".bias") and ("grad64:" + k[:-5] + ".weight") in g.files:
        s = max(s, 1e-1 * np.abs(g["grad64:" + k[:-5] + ".weight"]).max())
    return s + 1e-12
rows = sorted(((float((a[k] - b[k]).abs().max()) / scale(k), k) for k in a if ("grad64:" + k) in g.files),
              reverse=True)
for e, k in rows[:6]:
    print(f"{e:.3e}  {k}")

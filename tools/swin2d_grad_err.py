"""Per-parameter gradient error of the 2-D SWIN-UNet fixture against its fp64 gradients (the
quantity tests/test_swin.py bounds), largest first: to tell a knife-edge tolerance from a bug."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_swin as T  # noqa: E402

cuda = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "swinunet2d_small"
g = np.load(os.path.join(T.GOLD, name + ".npz"))
net = T.build(name).to(cuda).eval()
x = torch.from_numpy(g["x"]).to(cuda)
prob, _ = net(x)
loss = T.compound_loss(prob, torch.from_numpy(g["y"]).to(cuda))
loss.backward()
rows = []
for k, p in net.named_parameters():
    if ("grad:" + k) not in g.files:
        continue
    ref32, ref64 = g["grad:" + k], g["grad64:" + k]
    scale = np.abs(ref64).max()
    if k.endswith(".bias") and ("grad64:" + k[:-5] + ".weight") in g.files:
        scale = max(scale, 1e-1 * np.abs(g["grad64:" + k[:-5] + ".weight"]).max())
    noise = np.abs(ref32 - ref64).max() / (scale + 1e-12)
    err = np.abs(p.grad.cpu().numpy() - ref64).max() / (scale + 1e-12)
    rows.append((err, noise, k))
for err, noise, k in sorted(rows, reverse=True)[:8]:
    print(f"{err:.3e}  torch-fp32 {noise:.3e}  {k}")

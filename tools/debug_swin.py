"""Compare module outputs of the HIP SWINUNet with a dump of the reference's
(tools/_swin_debug.npz, made in the build container)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_swin import build
ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_swin_debug.npz"))
g = torch.Generator().manual_seed(1234)
x = torch.rand((2, 2, 32, 32, 16), generator=g).cuda()
net = build().cuda().eval()
res = []
def hook(name):
    def f(mod, inp, o):
        if isinstance(o, torch.Tensor) and name in ref.files:
            r = ref[name]
            a = o.detach().cpu().numpy()
            if a.shape != r.shape:
                res.append((name, "shape", a.shape, r.shape))
            else:
                res.append((name, float(np.abs(a - r).max() / (np.abs(r).max() + 1e-30))))
    return f
for name, m in net.named_modules():
    if name:
        m.register_forward_hook(hook(name))
with torch.no_grad():
    net(x, return_logits=True)
for r in res:
    print(*r)

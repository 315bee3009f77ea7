import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from adell_mri_amd import functional as HF, ops
from test_attention_links import build, GOLD
from oracle.torch_ref.unet import compound_loss
name = "unet2d_attention_links"
g = np.load(os.path.join(GOLD, name + ".npz"))
cuda = torch.device("cuda:0")
real_w = ops.conv3d_bwd_weight
real_na = ops.norm_act_bwd
real_dt = ops.norm_act_bwd_from_dt
log = []
def w(x0, dy, *a, **k):
    log.append(("wgrad", tuple(x0.shape), tuple(dy.shape), x0.detach().clone(), dy.detach().clone()))
    return real_w(x0, dy, *a, **k)
def na(x, dout, mean, rstd, *a, **k):
    out = real_na(x, dout, mean, rstd, *a, **k)
    log.append(("adn_bwd", tuple(x.shape), None, dout.detach().clone(), out[0].detach().clone(), x.detach().clone(), None if mean is None else mean.detach().clone(), None if rstd is None else rstd.detach().clone()))
    return out
ops.conv3d_bwd_weight = w
ops.norm_act_bwd = na
def run():
    log.clear()
    net = build(name).to(cuda).eval()
    x = torch.from_numpy(g["x"]).to(cuda)
    prob, _ = net(x)
    loss = compound_loss(prob, torch.from_numpy(g["y"]).to(cuda))
    loss.backward()
    torch.cuda.synchronize()
    return list(log)
HF.FLAGS["no_adn_fuse"] = True
HF.FLAGS["no_lowrank"] = True
a = run()
real_packed = HF._packed
def f(wt, mode):
    if mode == 0:
        old = HF.CONV_PRECISION; HF.CONV_PRECISION = "fp32"
        try: return real_packed(wt, mode)
        finally: HF.CONV_PRECISION = old
    return real_packed(wt, mode)
HF._packed = f
b = run()
rel = lambda u, v: float((u - v).abs().max() / (v.abs().max() + 1e-30))
print(len(a), len(b))
for i, (ea, eb) in enumerate(zip(a, b)):
    if ea[0] == "wgrad":
        print(i, ea[0], ea[1], ea[2], "x0 diff %.2e dy diff %.2e" % (rel(ea[3], eb[3]), rel(ea[4], eb[4])))
    else:
        print(i, ea[0], ea[1], "dout diff %.2e dx diff %.2e x diff %.2e mean %s rstd %s" % (rel(ea[3], eb[3]), rel(ea[4], eb[4]), rel(ea[5], eb[5]),
              None if ea[6] is None else "%.2e" % rel(ea[6], eb[6]), None if ea[7] is None else "%.2e" % rel(ea[7], eb[7])))
print("---- isolate entry 13")
ea, eb = a[13], b[13]
def torch_ref(x, dout):
    xr = x.detach().clone().requires_grad_(True)
    y = torch.relu(torch.nn.functional.instance_norm(xr.double(), eps=1e-5))
    y.backward(dout.double())
    return xr.grad.float()
for tag, e in (("f16x3-forward run", ea), ("fp32-forward run", eb)):
    x, dout, dx, mean, rstd = e[5], e[3], e[4], e[6], e[7]
    ref = torch_ref(x, dout)
    again = real_na(x, dout, mean, rstd, "relu")[0]
    print(tag, "logged dx vs torch fp64 ref %.2e" % rel(dx, ref), "recomputed vs ref %.2e" % rel(again, ref),
          "x range", float(x.min()), float(x.max()), "rstd max", float(rstd.max()), "strides", x.stride(), dout.stride())
    # statistics as the kernel got them vs recomputed from x
    m2 = x.double().mean(dim=(2, 3, 4)); v2 = x.double().var(dim=(2, 3, 4), unbiased=False)
    print("   mean err %.2e rstd err %.2e" % (float((mean.double().view_as(m2) - m2).abs().max()), float((rstd.double().view_as(m2) - (v2 + 1e-5).rsqrt()).abs().max() / (v2 + 1e-5).rsqrt().max())))
xa, xb = ea[5], eb[5]
ma, mb = ea[6].view(2, 16, 1, 1, 1), eb[6].view(2, 16, 1, 1, 1)
sa, sb = (xa - ma) > 0, (xb - mb) > 0
print("mask flips:", int((sa != sb).sum()), "of", sa.numel())
d = (xa - ma).abs()
print("elements with |x - mean| < 1e-5:", int((d < 1e-5).sum()), " < 1e-4:", int((d < 1e-4).sum()), " < 1e-3:", int((d < 1e-3).sum()))
idx = (sa != sb).nonzero()
for i in idx[:5]:
    i = tuple(int(v) for v in i)
    print("  flip at", i, "x-mean a %.3e b %.3e" % (float((xa - ma)[i]), float((xb - mb)[i])), "dout", float(ea[3][i]))
dxa, dxb = ea[4], eb[4]
dd = (dxa - dxb).abs()
print("dx diff: max", float(dd.max()), "at", tuple(int(v) for v in (dd == dd.max()).nonzero()[0]), "median", float(dd.median()), "dx max", float(dxb.abs().max()))

import json, os, sys, torch
sys.path.insert(0, ".")
from adell_mri_amd import _lib, ops
dev = torch.device("cuda:0"); ops.FLAGS["gemm_f16x3"] = True
def timed(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator(device=dev).manual_seed(0)
rows, c = 262144, 96; h = 4 * c
x = torch.randn(rows, c, device=dev, generator=g); hid = torch.randn(rows, h, device=dev, generator=g)
w1 = torch.randn(h, c, device=dev, generator=g) * 0.05; w2 = torch.randn(c, h, device=dev, generator=g) * 0.05
o1 = torch.empty(rows, h, device=dev); o2 = torch.empty(rows, c, device=dev)
out = {"dbg": os.environ.get("ADELL_ROWS_DBG", "0")}
for name, fn, nb in (("expand 96->384", lambda: ops.gemm_f16x3(rows, h, c, x, c, True, w1, c, True, out=o1), 4 * rows * (c + h)),
                     ("reduce 384->96", lambda: ops.gemm_f16x3(rows, c, h, hid, h, True, w2, h, True, out=o2), 4 * rows * (c + h))):
    for label, sw in (("rows", 0), ("tile", 1)):
        with _lib.tuning(gemm_norows=sw):
            us = timed(fn)
        out[f"{name} {label}"] = [round(us, 1), round(nb / us / 1e6, 2)]
print(json.dumps(out))
res = torch.randn(rows, c, device=dev, generator=g); b2 = torch.randn(c, device=dev, generator=g)
for label, sw in (("rows", 0), ("tile", 1)):
    with _lib.tuning(gemm_norows=sw):
        us = timed(lambda: ops.gemm_f16x3(rows, c, h, hid, h, True, w2, h, True, out=o2, bias=b2, residual=res))
    print(label, "reduce + bias + residual", round(us, 1), round(4 * rows * (2 * c + h) / us / 1e6, 2))

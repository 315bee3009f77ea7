"""Rows vs fp32 source of the same values into the same conv: agreement to an ulp, not to the bit
(both the implicit-GEMM instances and the 16-column z-ring kernel; a few hundred of 131 072 outputs
differ by one ulp)."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch
from adell_mri_amd import ops, _lib
cuda = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
N, size, pad = 2, (16, 16, 16), 1
x = ops.ndhwc((torch.randn(N, 16, *size, generator=g) * 1.5).to(cuda))
w = (torch.randn(16, 16, 3, 3, 3, generator=g) * 0.05).to(cuda)
b = torch.randn(16, generator=g).to(cuda)
wp = ops.pack_weight_f16x3(w, 0)
for e in (9, 6, 3, 0, -3):
    r0, s0 = ops.rows_from_f32(x, e)
    v0 = ops.rows_to_f32(r0, s0)
    for no16 in (0, 1):
        with _lib.tuning(igemm_no16=no16):
            y_ref, _ = ops.conv3d_fwd(v0, wp, b, 16, 3, 1, pad, want_stats=True)
            y, _ = ops.conv3d_fwd(r0, wp, b, 16, 3, 1, pad, want_stats=True, rows0=s0)
        d = (y - y_ref).abs()
        print("exp", e, "no16", no16, "equal", bool(torch.equal(y, y_ref)), "maxdiff", float(d.max()), "nz", int((d > 0).sum()), "of", d.numel(),
              "v0==x?", bool(torch.equal(v0, x)))

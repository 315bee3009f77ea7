"""Sustained rate of the forward kernels with split-row sources, in ONE process, windows of
back-to-back launches alternating between the variants (DVFS: compare under load, not in bursts):
  fp32       the one-brick-per-block instance splitting fp32 operands on the vector ALU
  rows       the same instance staging split rows by copy
  ws_rows    the persistent wave-specialised instance, loaders = LDS-DMA only (igemm_ws_rows)
  ws_fp32    the wave-specialised instance with the register-staged halo (igemm_ws, round 2)
usage: rows_sustained.py [C0 C1 Cout size batch]..."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adell_mri_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def run(c0, c1, cout, sz, batch, windows=8, per=40):
    g = torch.Generator().manual_seed(1)
    x0 = ops.ndhwc((torch.randn(batch, c0, sz, sz, sz, generator=g) * 1.5).to(dev))
    x1 = ops.ndhwc(torch.randn(batch, c1, sz, sz, sz, generator=g).to(dev)) if c1 else None
    w = (torch.randn(cout, c0 + c1, 3, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    wp = ops.pack_weight_f16x3(w, 0)
    r0, s0 = ops.rows_from_f32(x0, 9)
    r1, s1 = ops.rows_from_f32(x1, 10) if c1 else (None, None)
    flops = 2.0 * batch * sz ** 3 * (c0 + c1) * cout * 27
    h = _lib.lib()

    def fp32():
        ops.conv3d_fwd(x0, wp, b, cout, 3, 1, 1, x1=x1, want_stats=True)

    def rows():
        ops.conv3d_fwd(r0, wp, b, cout, 3, 1, 1, x1=r1, want_stats=True, rows0=s0, rows1=s1)

    variants = {"fp32": (fp32, {}), "rows": (rows, {}),
                "ws_rows": (rows, {"igemm_ws_rows": 1}), "ws_fp32": (fp32, {"igemm_ws": 1})}
    res = {k: [] for k in variants}
    for _ in range(60):
        rows()
    torch.cuda.synchronize()
    for _ in range(windows):
        for name, (fn, tune) in variants.items():
            for k, v in tune.items():
                h.adell_set_tuning(k.encode(), v)
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(per):
                fn()
            e1.record()
            torch.cuda.synchronize()
            for k in tune:
                h.adell_set_tuning(k.encode(), 0)
            res[name].append(flops / (e0.elapsed_time(e1) / per) / 1e9)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(json.dumps({"layer": f"{c0}+{c1}->{cout}@{sz}^3 x{batch}",
                      "TF_median": {k: round(v, 1) for k, v in med.items()},
                      "windows": {k: [round(t) for t in v] for k, v in res.items()}}))


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:]]
    cases = [args[i:i + 5] for i in range(0, len(args), 5)] or [
        [32, 0, 32, 128, 2], [64, 0, 32, 128, 2], [64, 0, 64, 128, 2], [32, 32, 64, 128, 2],
        [32, 0, 32, 64, 2]]
    for c in cases:
        run(*c)

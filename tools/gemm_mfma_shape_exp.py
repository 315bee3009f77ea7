"""v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 at the same wave tile in the f16x3 GEMM
(two experiment builds of csrc/gemm_f16x3.hip: `-DADELL_GEMM_MFMA16` -> _dbg/libgemm16.so, default
-> _dbg/libgemm32.so): sustained TF/s on random data, big square problems (MFMA-bound regime),
alternating the two builds in one process. usage: gemm_mfma_shape_exp.py [n=4096] [rounds=4]"""
import ctypes
import os
import sys

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
here = os.path.dirname(os.path.abspath(__file__))
libs = {k: ctypes.CDLL(os.path.join(here, "..", "adell_mri_amd", "csrc", "_dbg", f"libgemm{k}.so"))
        for k in ("32", "16")}
dev = torch.device("cuda:0")
A = torch.randn(n, n, device=dev)
B = torch.randn(n, n, device=dev)
C = {k: torch.empty(n, n, device=dev) for k in libs}
st = torch.cuda.current_stream().cuda_stream
vp, l, i = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
for L in libs.values():
    L.adell_gemm_f16x3.argtypes = [i, i, i, vp, l, i, vp, l, i, vp, l, vp, vp, l, vp, vp, vp, vp]


def run(k):
    rc = libs[k].adell_gemm_f16x3(n, n, n, A.data_ptr(), n, 1, B.data_ptr(), n, 1, C[k].data_ptr(), n,
                                  None, None, 0, None, None, None, st)
    assert rc == 0


for k in libs:
    for _ in range(20):
        run(k)
torch.cuda.synchronize()
ref = A.double()[:64] @ B.double().t()
for k in libs:
    err = float((C[k][:64].double() - ref).abs().max() / ref.abs().max())
    print(f"mfma{k}: rel err of the first 64 rows {err:.2e}")
for r in range(rounds):
    for k in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(30):
            run(k)
        e0.record()
        for _ in range(100):
            run(k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print(f"round {r} mfma{k}: {ms:.3f} ms = {2 * n ** 3 / ms / 1e9:.0f} TF", flush=True)

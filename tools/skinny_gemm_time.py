"""The skinny Linear-layer GEMMs of the SWIN-UNet step (config 5: per-voxel / per-patch layers with
2 ... 128 features over 0.5 ... 8.4 M rows) through ops.gemm / ops.gemm_f16x3 as functional.linear
dispatches them: us per call and TB/s of algorithmic bytes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adell_mri_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


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


# (M, N, K, a_kc, b_kc) as the timer tags of tools/secondary_layers.py print them
SHAPES = [(8388608, 8, 2, 1, 1), (8388608, 8, 2, 1, 0), (8388608, 2, 8, 1, 1), (8388608, 2, 8, 1, 0),
          (8, 2, 8388608, 0, 0), (2, 8, 8388608, 0, 0),
          (2097152, 8, 32, 1, 1), (2097152, 8, 32, 1, 0), (2097152, 32, 8, 1, 1), (2097152, 32, 8, 1, 0),
          (2097152, 8, 64, 1, 0), (2097152, 64, 8, 1, 1), (32, 8, 2097152, 0, 0), (64, 8, 2097152, 0, 0),
          (8, 32, 2097152, 0, 0), (8, 64, 2097152, 0, 0),
          (524288, 32, 128, 1, 1), (524288, 32, 128, 1, 0), (524288, 128, 32, 1, 1), (128, 32, 524288, 0, 0),
          (32, 128, 524288, 0, 0)]
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K, akc, bkc in SHAPES:
    A = torch.randn((M, K) if akc else (K, M), device=dev, generator=g)
    B = torch.randn((N, K) if bkc else (K, N), device=dev, generator=g)
    lda, ldb = A.shape[1], B.shape[1]
    ref = (A if akc else A.t()).double()[:4096 if akc else M] @ (B.t() if bkc else B).double() if akc else None
    f16 = ops.gemm_f16x3_ok(M, N, K, A, lda, bool(akc), B, ldb, bool(bkc))
    fn = (lambda: ops.gemm_f16x3(M, N, K, A, lda, bool(akc), B, ldb, bool(bkc))) if f16 else \
         (lambda: ops.gemm(M, N, K, A, lda, bool(akc), B, ldb, bool(bkc)))
    out = fn()
    err = float((out[:4096].double() - ref).abs().max() / ref.abs().max()) if ref is not None else \
        float((out.double() - A.double().t() @ B.double()).abs().max() / (A.double().t() @ B.double()).abs().max())
    us = timed(fn)
    nbytes = 4.0 * (M * K + K * N + M * N)
    print(f"{M}x{N}x{K} {'kc' if akc else 'outer'}/{'kc' if bkc else 'outer'}: {'f16x3' if f16 else 'fp32 '} "
          f"{us:8.1f} us  {nbytes / us / 1e6:5.2f} TB/s  err {err:.1e}")

"""GEMM shapes of one SWIN-UNet step (256x256x128) and their event-timed durations, fp32-MFMA GEMM
and f16x3 GEMM: which Linear layers are on which kernel, and what each costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
import bench
from adell_mri_amd import ops

dev = torch.device("cuda:0")
runs = bench.other_config_builders(dev) if hasattr(bench, "other_config_builders") else None
from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unetr import SWINUNet
from adell_mri_amd.modules.segmentation.losses import CompoundLoss, binary_focal_loss, binary_generalized_dice_loss
from adell_mri_amd.optim import FusedSGD
import importlib.util
spec = importlib.util.spec_from_file_location("bench_swin", os.path.join(os.path.dirname(__file__), "bench_swin.py"))
src = open(spec.origin).read()
log = collections.OrderedDict()
r_g, r_h = ops.gemm, ops.gemm_f16x3


def timed(kind, real, M, N, K, A, lda, a_kc, B, ldb, b_kc, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = real(M, N, K, A, lda, a_kc, B, ldb, b_kc, *a, **k)
    e1.record()
    log.setdefault((kind, M, N, K, int(a_kc), int(b_kc)), []).append((e0, e1))
    return out


ops.gemm = lambda *a, **k: timed("f32", r_g, *a, **k)
ops.gemm_f16x3 = lambda *a, **k: timed("f16x3", r_h, *a, **k)
sys.argv = ["bench_swin.py", "--steps", "2", "--warmup", "1"]
exec(compile(src, spec.origin, "exec"), {"__name__": "__main__", "__file__": spec.origin})
torch.cuda.synchronize()
rows = []
for key, evs in log.items():
    ms = [a.elapsed_time(b) for a, b in evs]
    rows.append((sum(ms) / 3, key, len(evs) / 3, min(ms)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"GEMM time per step (event-timed, includes launch gaps): {tot:.2f} ms")
for t, key, n, mn in rows[:30]:
    kind, M, N, K, akc, bkc = key
    print(f"{t:7.3f} ms/step x{n:4.1f} min {mn * 1e3:7.1f} us  {kind:6s} M={M} N={N} K={K} a_kc={akc} b_kc={bkc}  "
          f"{2.0 * M * N * K / (mn * 1e-3) / 1e12:6.1f} TF  {(M * K + N * K + M * N) * 4 / (mn * 1e-3) / 1e9:7.0f} GB/s")

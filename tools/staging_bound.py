"""Upper bounds, INSIDE the training step, for anything that only makes operand staging cheaper in
the implicit-GEMM kernels (split rows, LDS-DMA halos, the wave-specialised instance): the debug
build's timing switches drop the staging work (results wrong, values stay finite) while the MFMA
stream, the barriers and the epilogues stay. Alternating blocks of steps in one process.
  bash tools/build_dbg.sh; ADELL_HIP_LIBRARY=adell_mri_amd/libadellhip_dbg.so python tools/staging_bound.py"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from adell_mri_amd import _lib  # noqa: E402
from adell_mri_amd import functional as HF  # noqa: E402
from adell_mri_amd.parallel import GradSync  # noqa: E402
from adell_mri_amd.trainer import StepRunner  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
net, _ = bench.build_module(dev, bench.CONFIG)
net.train()
opt = net.configure_optimizers()["optimizer"]
for g in opt.param_groups:
    g["lr"] = 0.0                      # wrong gradients must not walk the weights away
runner = StepRunner(net, opt, GradSync(opt))
batch = bench.synthetic_batch(2, (128, 128, 128), dev, 42)
HF.FLAGS["no_rows"] = True

VARIANTS = {
    "block": dict(igemm_ws=0, igemm_dbg=0, fuse=True),
    "block_no_halo_staging": dict(igemm_ws=0, igemm_dbg=1, fuse=True),
    "block_no_staging": dict(igemm_ws=0, igemm_dbg=3, fuse=True),
    "block_nofuse": dict(igemm_ws=0, igemm_dbg=0, fuse=False),
    "ws_nofuse": dict(igemm_ws=1, igemm_dbg=0, fuse=False),
    "ws_nofuse_no_halo_work": dict(igemm_ws=1, igemm_dbg=1, fuse=False),
    "ws_nofuse_idle_loaders": dict(igemm_ws=1, igemm_dbg=3, fuse=False),
    "block_nofuse_no_staging": dict(igemm_ws=0, igemm_dbg=3, fuse=False),
}


def apply(v):
    h = _lib.lib()
    _lib.check(h.adell_set_tuning(b"igemm_ws", v["igemm_ws"]))
    _lib.check(h.adell_set_tuning(b"igemm_dbg", v["igemm_dbg"]))
    HF.FLAGS["no_adn_fuse"] = not v["fuse"]
    torch.cuda.synchronize()


for _ in range(6):
    runner.train_step(batch)
torch.cuda.synchronize()
res = {k: [] for k in VARIANTS}
for r in range(rounds):
    for name, v in VARIANTS.items():
        apply(v)
        runner.train_step(batch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            loss = runner.train_step(batch)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(round(e0.elapsed_time(e1) / steps, 3))
apply(VARIANTS["block"])
print(json.dumps({k: {"median_ms": statistics.median(v), "blocks": v} for k, v in res.items()}, indent=1))

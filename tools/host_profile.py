"""Host-side cost of a secondary config's training step by Python function (cProfile over 5 steps
after bench.py's warm-up): ADELL_BENCH_ONLY=cfg3_unetr_96 python tools/host_profile.py"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def probe(key, runner, batch):
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        runner.train_step(batch)
    pr.disable()
    torch.cuda.synchronize()
    out = io.StringIO()
    st = pstats.Stats(pr, stream=out)
    st.sort_stats("tottime").print_stats(45)
    print(f"==== {key} (5 steps)")
    print(out.getvalue()[:9000])
    return {}


bench.PROBE = probe
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bench.other_config_runs(dev, 0, 1, torch.cuda.synchronize, lambda v, d: float(v))

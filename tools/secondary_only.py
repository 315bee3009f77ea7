"""bench.py's `secondary` BASELINE-config runs alone (ADELL_BENCH_ONLY=<keys> selects; ADELL_BENCH_NO_GRAPH=1
keeps config 3 eager): one line per workload."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
out = bench.other_config_runs(dev, 0, 1, torch.cuda.synchronize, lambda v, d: float(v))
for k, v in out.items():
    r = v.get("step_record") or {}
    print(k, "ms", round(v.get("ms_per_step", 0), 2), "median", round(v.get("median_ms_per_step", 0), 2),
          "per", r.get("per_step_ms"), "host", r.get("host_ms"), "graph replay", v.get("hip_graph_replay"), v.get("error"))

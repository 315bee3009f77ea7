"""A/B of two BUILDS of libadellhip.so on the bench workload: child processes alternate between the
two libraries on one box (each: 8 warm-up + N timed training steps; ADELL_HIP_LIBRARY selects the
build). usage: ab_lib.py libA.so libB.so [rounds=3] [steps=16]"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch

    import bench
    from adell_mri_amd.parallel import GradSync
    from adell_mri_amd.trainer import StepRunner

    steps = int(sys.argv[2])
    dev = torch.device("cuda:0")
    net, _ = bench.build_module(dev, bench.CONFIG)
    net.train()
    opt = net.configure_optimizers()["optimizer"]
    runner = StepRunner(net, opt, GradSync(opt))
    batch = bench.synthetic_batch(int(net.batch_size), (128, 128, 128), dev, 42)
    for _ in range(8):
        runner.train_step(batch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        runner.train_step(batch)
    e1.record()
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print("MS_PER_STEP", e0.elapsed_time(e1) / steps, "device_allocs", st.get("num_device_alloc"),
          "retries", st.get("num_alloc_retries"), "reserved_GB",
          round(torch.cuda.memory_reserved() / 2 ** 30, 1))
    sys.exit(0)

libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 16
res = {0: [], 1: []}
for r in range(rounds):
    for which in (0, 1):
        env = dict(os.environ, ADELL_HIP_LIBRARY=libs[which])
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(steps)],
                             env=env, capture_output=True, text=True, timeout=300)
        ms = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("MS_PER_STEP")]
        if not ms:
            print(out.stdout[-2000:], out.stderr[-2000:])
            sys.exit(1)
        res[which].append(round(ms[0], 3))
print(json.dumps({"A": sys.argv[1], "B": sys.argv[2], "ms_A": res[0], "ms_B": res[1],
                  "median_A": statistics.median(res[0]), "median_B": statistics.median(res[1])}))

"""Host enqueue time per training step against the GPU time of the same steps (no synchronisation
inside the loop): on the bench workload the host queues a step in ~10 ms and runs ~25 ms per step
ahead of the GPU, so the 45-60 us gaps a rocprofv3 kernel trace shows between the small forward
kernels are the tracer's (un-profiled 36.3 ms/step vs 37.1 profiled), not host starvation."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from adell_mri_amd.parallel import GradSync
from adell_mri_amd.trainer import StepRunner
dev = torch.device("cuda:0")
net, _ = bench.build_module(dev, bench.CONFIG)
net.train()
opt = net.configure_optimizers()["optimizer"]
runner = StepRunner(net, opt, GradSync(opt))
batch = bench.synthetic_batch(int(net.batch_size), (128, 128, 128), dev, 42)
for _ in range(6):
    runner.train_step(batch)
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for i in range(10):
    a = time.perf_counter()
    runner.train_step(batch)
    ts.append((time.perf_counter() - a) * 1e3)
host_total = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
total = (time.perf_counter() - t0) * 1e3
print("host ms per step:", [round(t, 1) for t in ts], "host total", round(host_total, 1), "with sync", round(total, 1))

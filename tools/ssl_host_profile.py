#!/usr/bin/env python
"""Host-side cost of the VICReg / ConvNeXt step (config 4 is near host-bound): wall time per step
without the kernel timer, pure enqueue time, and the cProfile hot spots of the enqueue."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
    from adell_mri_amd.modules.self_supervised.pl import SelfSLConvNeXtPL
    from adell_mri_amd.trainer import StepRunner

    dev = torch.device("cuda", 0)
    adn = get_adn_fn(1, "layer", "gelu", 0.0)
    torch.manual_seed(0)
    net = SelfSLConvNeXtPL(
        aug_image_key_1="a", aug_image_key_2="b", ssl_method="vicreg", stop_gradient=False,
        learning_rate=0.005, weight_decay=0.001, n_epochs=100, batch_size=16, ema=None,
        backbone_args=dict(spatial_dim=3, in_channels=1,
                           structure=[[96, 384, 7, 3], [192, 768, 7, 3], [384, 1536, 7, 9],
                                      [768, 3072, 3, 3]],
                           maxpool_structure=[[2, 2, 2]] * 4),
        projection_head_args=dict(in_channels=768, structure=[1024, 2048, 1024], adn_fn=adn),
        prediction_head_args=dict(in_channels=1024, structure=[2048, 1024], adn_fn=adn)).to(dev)
    net.train()
    runner = StepRunner(net)
    g = torch.Generator().manual_seed(1)
    shape = (16, 1, 64, 64, 64)
    batch = {"a": torch.rand(shape, generator=g).to(dev), "b": torch.rand(shape, generator=g).to(dev)}
    for _ in range(3):
        runner.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        runner.train_step(batch)
    torch.cuda.synchronize()
    print("ms/step (no timer):", (time.perf_counter() - t0) / 10 * 1e3)
    host = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.train_step(batch)
        host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print("host enqueue ms (GPU idle at start):", sorted(host)[len(host) // 2] * 1e3)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        runner.train_step(batch)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()

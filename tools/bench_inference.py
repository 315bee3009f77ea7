#!/usr/bin/env python
"""Whole-volume sliding-window inference throughput: the config-2 U-Net
in eval mode over a 256x256x128 2-channel volume, 128^3 windows at half-window stride, with and
without the test-time flip. Not the headline bench."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.unet import UNet
    from adell_mri_amd.utils.inference import SegmentationInference, window_plan

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = UNet(spatial_dimensions=3, conv_type="regular", link_type="residual",
               upscale_type="transpose", norm_type="instance", padding=1, dropout_param=0.1,
               activation_fn=activation_factory["swish"], in_channels=2, n_classes=2,
               depth=[32, 32, 64, 128, 256], kernel_sizes=[3] * 5, strides=[2] * 5).to(dev).eval()
    x = torch.rand((1, 2, 256, 256, 128), device=dev)
    out = {}
    for flip in (False, True):
        for bs in (1, 4):
            sli = SegmentationInference(base_inference_function=lambda t: net(t)[0],
                                        sliding_window_size=[128, 128, 128], stride=0.5,
                                        n_classes=2, flip=flip, inference_batch_size=bs)
            sli(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                y = sli(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            # the reference's enumeration starts a window at every multiple of the stride and
            # moves the ones that stick out back to the edge: 4 x 4 x 2 = 32 windows here
            nwin = len(window_plan((256, 256, 128), (128,) * 3, (64,) * 3)) * (2 if flip else 1)
            out[f"flip={flip},window_batch={bs}"] = {"s_per_volume": dt, "windows_per_s": nwin / dt}
    print(json.dumps({"workload": "U-Net (config 2) eval, 256x256x128 volume, 128^3 windows, "
                                  "stride 64", "output_shape": list(y.shape), "results": out}))


if __name__ == "__main__":
    main()

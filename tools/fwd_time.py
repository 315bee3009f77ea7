"""Where the time of one sliding-window step goes (crop, forward, accumulate)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unet import UNet
dev = torch.device("cuda", 0)
net = UNet(spatial_dimensions=3, conv_type="regular", link_type="residual", upscale_type="transpose", norm_type="instance", padding=1, dropout_param=0.1, activation_fn=activation_factory["swish"], in_channels=2, n_classes=2, depth=[32, 32, 64, 128, 256], kernel_sizes=[3] * 5, strides=[2] * 5).to(dev).eval()
X = torch.rand((1, 2, 256, 256, 128), device=dev)
total = torch.zeros((1, 1, 256, 256, 128), device=dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
with torch.no_grad():
    ms, crop = t(lambda: X[..., 64:192, 64:192, 0:128]); print("crop view ms", ms)
    ms, cc = t(lambda: torch.cat([crop], 0)); print("cat ms", ms)
    ms, y = t(lambda: net(cc)[0]); print("forward(cat crop) ms", ms)
    ms, y2 = t(lambda: net(X[:, :, :128, :128, :].contiguous())[0]); print("forward(contig) ms", ms)
    ms, _ = t(lambda: total[..., 64:192, 64:192, 0:128].add_(y.squeeze(0).squeeze(0))); print("accumulate ms", ms)
    print(y.shape, y.stride(), cc.stride())

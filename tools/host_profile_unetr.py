#!/usr/bin/env python
"""cProfile of the host side of the UNETR step (BASELINE configs[2]: ~950 launches, ~20 ms of enqueue
time per 22 ms step -- the one configuration that is host-bound)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from adell_mri_amd.modules.activations import activation_factory  # noqa: E402
from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,  # noqa: E402
                                                       binary_generalized_dice_loss)
from adell_mri_amd.modules.segmentation.unetr import UNETR  # noqa: E402
from adell_mri_amd.optim import FusedSGD  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
kw = dict(image_size=[96, 96, 96], patch_size=[16, 16, 16], number_of_blocks=8,
          attention_dim=512, hidden_dim=512, embedding_size=512, n_heads=8,
          return_at=[2, 4, 6], mlp_structure=[1024], dropout_rate=0.1,
          embed_method="linear", spatial_dimensions=3, conv_type="regular",
          link_type="residual", upscale_type="transpose", norm_type="instance", padding=1,
          dropout_param=0.0, activation_fn=activation_factory["leaky_relu"], in_channels=1,
          n_classes=2, depth=[16, 32, 64, 128], kernel_sizes=[3, 3, 3, 3])
net = UNETR(**kw).to(dev).train()
loss_fn = CompoundLoss([(binary_generalized_dice_loss, {"smooth": 1e-5, "eps": 1e-6}),
                        (binary_focal_loss, {"gamma": 0.0, "eps": 1e-6})])
opt = FusedSGD(net.parameters(), lr=5e-3, momentum=0.99, weight_decay=5e-4, nesterov=True)
g = torch.Generator().manual_seed(1)
x = torch.rand((4, 1, 96, 96, 96), generator=g).to(dev)
y = (torch.rand((4, 1, 96, 96, 96), generator=g) > 0.9).float().to(dev)


def step():
    opt.zero_grad()
    prob, _ = net(x)
    loss = torch.stack([t.mean() for t in loss_fn(prob, y)]).mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(34)
# who issues the device-to-device copies / host uploads (rocclr copyBuffer in the kernel trace)
st.print_callers("copy_|contiguous|clone|'to' of|tensor|zeros|full")
# one step under the torch profiler: memcpy / memset records with the op that issued them
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))


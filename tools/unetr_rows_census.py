import sys
sys.path.insert(0, "/root/repo")
import torch
from adell_mri_amd import ops
from adell_mri_amd import functional as HF
from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unetr import UNETR
dev = torch.device("cuda:0")
torch.manual_seed(0)
kw = dict(image_size=[96, 96, 96], patch_size=[16, 16, 16], number_of_blocks=8,
          attention_dim=512, hidden_dim=512, embedding_size=512, n_heads=8,
          return_at=[2, 4, 6], mlp_structure=[1024], dropout_rate=0.1,
          embed_method="linear", spatial_dimensions=3, conv_type="regular",
          link_type="residual", upscale_type="transpose", norm_type="instance", padding=1,
          dropout_param=0.0, activation_fn=activation_factory["leaky_relu"], in_channels=1,
          n_classes=2, depth=[16, 32, 64, 128], kernel_sizes=[3, 3, 3, 3])
net = UNETR(**kw).to(dev).train()
x = torch.rand((2, 1, 96, 96, 96), device=dev)
log = []
r_cf = ops.conv3d_fwd
def cf(x0, wp, bias, Cout, kernel, stride, padding, x1=None, **k):
    log.append((tuple(x0.shape), None if x1 is None else x1.shape[1], Cout, k.get("rows0") is not None, k.get("rows1") is not None, k.get("residual") is not None))
    return r_cf(x0, wp, bias, Cout, kernel, stride, padding, x1=x1, **k)
ops.conv3d_fwd = cf
y = net(x)[0]
for l in log: print(l)
print(HF._ROWS_PLAN)

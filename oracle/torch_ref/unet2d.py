"""TEST INFRASTRUCTURE ONLY -- stock-torch (CPU) restatement of BASELINE configs[0]: the 2-D
U-Net of the reference's testing/test_unet.py:63-72 (``UNet(2, depth=[16, 32, 64],
upscale_type="transpose", padding="same", strides=2, kernel_sizes=3, conv_type="regular",
link_type="identity")`` with the constructor defaults BatchNorm2d + PReLU, 140 748 parameters),
driven by a reference-keyed ``state_dict``.

Layout as adell_mri/modules/segmentation/unet.py:245-258 (conv_block_2d: Conv(in, in, k, stride) ->
ADN(in) -> Conv(in, out, k)), :543-586 (encoder level = block + ADN, strided block + ADN), :415-459
(ConvTranspose2d(k = s, stride s, padding max(s - 2, 0))), :605-655 (decoder block + ADN, head
Conv -> ADN -> Conv 1x1 -> Sigmoid), :751-843 (forward); ADN ordering "NDA" with BatchNorm2d in
training mode (batch statistics), Dropout and PReLU (adn_fn.py:140-152). Pinned against outputs of
the real reference by tests/test_oracle_golden.py (fixture unet2d_cfg1, oracle/make_golden.py).
bench.py times it as the cfg-1 leg of ``cpu_baseline``. The product never imports
this file."""
import statistics
import time

import torch
import torch.nn.functional as F

DEPTH, STRIDES = [16, 32, 64], [2, 2, 2]


class UNet2dOracle:
    def __init__(self, state_dict, dropout_param=0.0):
        self.sd = {k: v.detach().clone().float() for k, v in state_dict.items()}
        self.p = dropout_param
        self.training = True

    def parameters(self):
        return [v for k, v in self.sd.items()
                if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]

    def requires_grad_(self, flag=True):
        for v in self.parameters():
            v.requires_grad_(flag)
        return self

    def conv(self, key, x, stride=1, padding=1):
        return F.conv2d(x, self.sd[key + ".weight"], self.sd.get(key + ".bias"), stride=stride,
                        padding=padding)

    def adn(self, key, x):
        pre = key + ".op."
        x = F.batch_norm(x, None, None, self.sd[pre + "normalization.weight"],
                         self.sd[pre + "normalization.bias"], training=True, eps=1e-5)
        if self.training and self.p > 0:
            x = F.dropout(x, self.p, True)
        return F.prelu(x, self.sd[pre + "activation.weight"])

    def block(self, key, x, stride):
        x = self.conv(key + ".0", x, stride)
        x = self.adn(key + ".1", x)
        return self.conv(key + ".2", x)

    def forward(self, x, return_logits=True):
        L = len(DEPTH)
        enc, cur = [], x
        for i in range(L):
            key = f"encoding_operations.{i}"
            cur = self.adn(key + ".0.1", self.block(key + ".0.0", cur, 1))
            enc.append(cur)
            if i < L - 1:
                cur = self.adn(key + ".1.1", self.block(key + ".1.0", cur, STRIDES[i]))
        for i in range(L - 1):
            s = STRIDES[::-1][1:][i]
            cur = F.conv_transpose2d(cur, self.sd[f"upscale_ops.{i}.weight"],
                                     self.sd[f"upscale_ops.{i}.bias"], stride=s,
                                     padding=max(s - 2, 0))
            cur = torch.cat((cur, enc[-i - 2]), 1)
            key = f"decoding_operations.{i}"
            cur = self.adn(key + ".1", self.block(key + ".0", cur, 1))
        cur = self.conv("final_layer.0", cur)
        cur = self.adn("final_layer.1", cur)
        cur = self.conv("final_layer.2", cur, padding=0)
        return cur if return_logits else torch.sigmoid(cur)


def training_step_seconds(batch=4, size=128, steps=3, warmup=1, seed=0):
    """Median wall time of one training step (forward, dice + focal, backward, SGD-Nesterov
    lr 5e-4 / wd 5e-3: BASELINE.md section 4) on ``batch`` x 1 x size x size inputs."""
    from adell_mri_amd.modules.segmentation.unet import UNet
    from oracle.torch_ref.unet import compound_loss
    from oracle.weights import fill_state_dict

    net = UNet(spatial_dimensions=2, depth=DEPTH, upscale_type="transpose", padding="same",
               strides=STRIDES, kernel_sizes=[3, 3, 3], conv_type="regular",
               link_type="identity", activation_fn=torch.nn.PReLU, dropout_param=0.0)
    ref = UNet2dOracle(fill_state_dict(net.state_dict())).requires_grad_(True)
    opt = torch.optim.SGD(ref.parameters(), lr=5e-4, momentum=0.99, weight_decay=5e-3,
                          nesterov=True)
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((batch, 1, size, size), generator=g)
    y = (torch.rand((batch, 1, size, size), generator=g) > 0.9).float()
    times = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = compound_loss(ref.forward(x, return_logits=False), y)
        loss.backward()
        opt.step()
        if i >= warmup:
            times.append(time.perf_counter() - t0)
    return statistics.median(times)

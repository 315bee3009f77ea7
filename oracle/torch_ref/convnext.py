"""TEST INFRASTRUCTURE ONLY -- functional stock-torch (CPU) restatement of the reference
ConvNeXt network and VICReg loss, driven by a reference-keyed ``state_dict``.

Follows adell_mri/modules/layers/conv_next.py:86-235 (backbone: stem Conv(k=4, stride) +
channels-first LayerNorm, stages of blocks each followed by MaxPool) and :388-452 (heads),
layers/res_blocks.py:516-604 (ConvNeXtBlock3d), layers/res_net.py:278-324
(ProjectionHead: global spatial max, Linear (+ LayerNorm + GELU) layers),
layers/regularization.py:60-92 (LayerNorm eps 1e-6) and
self_supervised/losses/vicreg.py:30-165. Pinned against outputs of the real reference
by tests/test_oracle_golden.py (fixtures: oracle/make_golden.py gen_ssl).
The product never imports this file.
"""
import math

import torch
import torch.nn.functional as F


class ConvNeXtOracle:
    """cfg: structure [[C, inter, k, N], ...], maxpool_structure, first_layer_stride,
    head_adn (True: Linear -> LayerNorm(eps 1e-5) -> GELU between head layers)."""

    def __init__(self, state_dict, cfg):
        self.sd = {k: v.detach().clone().float() for k, v in state_dict.items()}
        self.cfg = cfg

    def parameters(self):
        return list(self.sd.values())

    def requires_grad_(self, flag=True):
        for v in self.sd.values():
            v.requires_grad_(flag)
        return self

    def block(self, key, x):
        sd = self.sd
        w = sd[key + ".dwconv.weight"]
        h = F.conv3d(x, w, sd[key + ".dwconv.bias"], padding=[k // 2 for k in w.shape[2:]],
                     groups=w.shape[0])
        h = h.permute(0, 2, 3, 4, 1)
        h = F.layer_norm(h, (h.shape[-1],), sd[key + ".norm.weight"], sd[key + ".norm.bias"], 1e-6)
        h = F.gelu(F.linear(h, sd[key + ".pwconv1.weight"], sd[key + ".pwconv1.bias"]))
        h = F.linear(h, sd[key + ".pwconv2.weight"], sd[key + ".pwconv2.bias"])
        if key + ".gamma" in sd:
            h = sd[key + ".gamma"] * h
        x = x + h.permute(0, 4, 1, 2, 3)
        if key + ".out_layer.0.weight" in sd:
            x = F.gelu(F.conv3d(x, sd[key + ".out_layer.0.weight"], sd[key + ".out_layer.0.bias"]))
        return x

    def backbone(self, x):
        sd, cfg = self.sd, self.cfg
        x = F.conv3d(x, sd["backbone.input_layer.0.weight"], sd["backbone.input_layer.0.bias"],
                     stride=cfg.get("first_layer_stride", 4))
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + 1e-6)
        x = (sd["backbone.input_layer.1.weight"].reshape(1, -1, 1, 1, 1) * x
             + sd["backbone.input_layer.1.bias"].reshape(1, -1, 1, 1, 1))
        mps = cfg.get("maxpool_structure") or [2] * len(cfg["structure"])
        for i, (st, mp) in enumerate(zip(cfg["structure"], mps)):
            n_blocks = max(st[3], 2)  # conv_next.py:186-202: first + range(1, N-1) + last
            for j in range(n_blocks):
                x = self.block(f"backbone.operations.{i}.{j}", x)
            x = F.max_pool3d(x, mp, mp)
        return x

    def head(self, key, x, n_layers):
        sd = self.sd
        if x.dim() > 2:
            x = x.flatten(start_dim=2).max(-1).values
        for i in range(n_layers - 1):
            k = f"{key}.op.linear_{i}.0"
            x = F.linear(x, sd[k + ".weight"], sd[k + ".bias"])
            if self.cfg.get("head_adn", True):  # get_adn_fn(1, "layer", "gelu", 0): A-D-N order
                nk = f"{key}.op.linear_{i}.1"
                x = self._adn(nk, x)
        k = f"{key}.op.linear_{n_layers - 1}"
        return F.linear(x, sd[k + ".weight"], sd[k + ".bias"])

    def _adn(self, key, x):
        sd = self.sd
        order = self.cfg.get("adn_ordering", "NDA")
        for c in order:
            if c == "N":
                nk = [k for k in sd if k.startswith(key + ".") and k.endswith("weight")]
                w = sd[nk[0]]
                x = F.layer_norm(x, (x.shape[-1],), w, sd[nk[0][:-6] + "bias"], 1e-5)
            elif c == "A":
                x = F.gelu(x)
        return x

    def forward(self, x, ret="projection"):
        x = self.backbone(x)
        if ret == "representation":
            return x
        n_proj = len(self.cfg["projection_structure"])
        x = self.head("projection_head.0", x, n_proj)
        x = F.layer_norm(x, (x.shape[-1],), self.sd["projection_head.1.weight"],
                         self.sd["projection_head.1.bias"], 1e-5)
        if ret == "projection":
            return x
        return self.head("prediction_head", x, len(self.cfg["prediction_structure"]))


def vicreg_loss(x1, x2, min_var=1.0, eps=1e-4, lam=25.0, mu=25.0, nu=0.1):
    """(lam*inv, mu*var, nu*cov) -- vicreg.py:60-165."""
    def var_term(x):
        return F.relu(min_var - torch.sqrt(torch.var(x, 0) + eps)).mean()

    def cov_term(x):
        xc = x - x.mean(0)
        cov = (xc.T @ xc) / (x.shape[0] - 1)
        off = cov - torch.diag(torch.diag(cov))
        return (off / math.sqrt(x.shape[1])).pow(2).sum()

    inv = ((x1 - x2) ** 2).sum() / x1.numel()
    return (lam * inv, mu * (var_term(x1) / 2 + var_term(x2) / 2),
            nu * (cov_term(x1) / 2 + cov_term(x2) / 2))

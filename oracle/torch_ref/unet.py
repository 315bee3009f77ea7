"""TEST INFRASTRUCTURE ONLY -- functional stock-torch (CPU) restatement of the
reference U-Net forward / training step, driven by a reference-keyed
``state_dict``.

It follows adell_mri/modules/segmentation/unet.py:543-655 (how the encoder,
links, upscaling, decoder and head are laid out) and :751-843 (forward), with
ActDropNorm ordering "NDA" (unet.py:697-714; adn_fn.py:140-152) and
ResidualBlock3d (res_blocks.py:150-200). Pinned against outputs of the real
reference by tests/test_oracle_golden.py (fixtures: oracle/make_golden.py).
The product never imports this file.
"""
import torch
import torch.nn.functional as F

_ACTS = {
    "identity": lambda x: x, "swish": F.silu, "relu": F.relu, "gelu": F.gelu,
    "leaky_relu": F.leaky_relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "elu": F.elu,
}


def _t(v, n=3):
    return [v] * n if isinstance(v, int) else list(v)


class UNetOracle:
    """cfg keys: depth, kernel_sizes, strides, padding, norm_type ("instance" |
    "identity"), activation (name), link_type ("identity" | "conv" | "residual"),
    n_classes, dropout_param (only used when training=True)."""

    def __init__(self, state_dict, cfg):
        self.sd = {k: v.detach().clone().float() for k, v in state_dict.items()}
        self.cfg = cfg
        self.training = False

    def parameters(self):
        return list(self.sd.values())

    def requires_grad_(self, flag=True):
        for v in self.sd.values():
            v.requires_grad_(flag)
        return self

    # -- building blocks ---------------------------------------------------
    def conv(self, key, x, stride=1, padding=0):
        w = self.sd[key + ".weight"]
        if padding == "same":
            padding = [k // 2 for k in w.shape[2:]]
        return F.conv3d(x, w, self.sd.get(key + ".bias"), stride=stride, padding=padding)

    def adn(self, x):
        nt = self.cfg.get("norm_type", "instance")
        if nt == "instance":
            x = F.instance_norm(x, eps=1e-5)
        elif nt not in ("identity", None):
            raise NotImplementedError(nt)
        p = self.cfg.get("dropout_param", 0.0)
        if self.training and p > 0:
            x = F.dropout(x, p, True)
        return _ACTS[self.cfg.get("activation", "swish")](x)

    def conv_block(self, key, x, stride, padding):
        x = self.conv(key + ".0", x, stride, padding)
        x = self.adn(x)
        return self.conv(key + ".2", x, 1, padding)

    def residual_block(self, key, x):
        h = self.conv(key + ".op.0", x, 1, "same")
        h = self.adn(h)
        h = self.conv(key + ".op.2", h, 1, "same")
        out = h + x
        if key + ".final_op.weight" in self.sd:
            out = self.conv(key + ".final_op", out)
        return self.adn(out)

    # -- forward -----------------------------------------------------------
    def forward(self, x, return_logits=True):
        enc, cur = self.encode("encoding_operations", x)
        return self.decode(enc, cur, return_logits)

    def encode(self, prefix, x):
        """One encoder tree (unet.py:543-586): the per-level outputs and the bottleneck."""
        c = self.cfg
        depth, strides, ks = c["depth"], c["strides"], c["kernel_sizes"]
        pad = c.get("padding", "same")
        L = len(depth)
        enc = []
        cur = x
        for i in range(L):
            k = _t(ks[min(i, L - 2)] if i == L - 1 else ks[i])
            key = f"{prefix}.{i}"
            p_ = [kk // 2 for kk in k] if pad == "same" else pad
            cur = self.adn(self.conv_block(key + ".0.0", cur, 1, p_))
            enc.append(cur)
            if i < L - 1:
                cur = self.adn(self.conv_block(key + ".1.0", cur, _t(strides[i]),
                                               [kk // 2 for kk in k]))
        return enc, cur

    def decode(self, enc, cur, return_logits=True):
        """Links, upscaling, decoder and head (unet.py:790-843)."""
        c = self.cfg
        depth, strides, ks = c["depth"], c["strides"], c["kernel_sizes"]
        pad = c.get("padding", "same")
        L = len(depth)
        for i in range(L - 1):
            skip = enc[-i - 2]
            lt = c.get("link_type", "identity")
            if lt == "residual":
                skip = self.residual_block(f"link_ops.{i}", skip)
            elif lt == "conv":
                skip = self.adn(self.conv(f"link_ops.{i}.0", skip, 1,
                                          [1, 1, 1] if pad == "same" else pad))
            s = _t(strides[::-1][1:][i])
            cur = F.conv_transpose3d(cur, self.sd[f"upscale_ops.{i}.weight"],
                                     self.sd[f"upscale_ops.{i}.bias"], stride=s,
                                     padding=[max(j - 2, 0) for j in s])
            cur = torch.cat((cur, skip), 1)
            k = _t(ks[-2::-1][i])
            p_ = [kk // 2 for kk in k] if pad == "same" else pad
            cur = self.adn(self.conv_block(f"decoding_operations.{i}.0", cur, 1, p_))
        cur = self.conv("final_layer.0", cur, 1, "same")
        cur = self.adn(cur)
        cur = self.conv("final_layer.2", cur)
        if return_logits:
            return cur
        return torch.sigmoid(cur) if c.get("n_classes", 2) <= 2 else torch.softmax(cur, 1)


# -- loss / optimiser restatements (adell_mri/modules/segmentation/losses.py) ----
def dice_loss(pred, target, smooth=1e-5, eps=1e-6):
    """binary_generalized_dice_loss with weight=1, scale=1 (losses.py:14-54,251-292)."""
    t = target.flatten(2)
    p = pred.flatten(2)
    num = torch.clip(t * p, 0).sum(-1).sum(-1)
    den = torch.clip(t + p + smooth, eps).sum(-1).sum(-1)
    return 1 - 2 * num / den


def focal_loss(pred, target, gamma=1.0, eps=1e-6):
    """binary_focal_loss with alpha=1, threshold=0.5, scale=1 (losses.py:112-164)."""
    e = torch.as_tensor(eps).type_as(pred)
    p = torch.maximum(pred, e).flatten(2)
    q = torch.maximum(1 - p, e)
    t = (target > 0.5).long().flatten(2)
    return (p ** gamma * torch.log(p) * t + q ** gamma * torch.log(q) * (1 - t)).negative().mean(-1)


def compound_loss(pred, target, smooth=1e-5, dice_eps=1e-6, gamma=1.0, focal_eps=1e-6):
    """calculate_loss of UNetBasePL (segmentation/pl.py:218-222) for the
    dice + focal CompoundLoss of u-net-3d-resnet.yaml: mean over the list of
    per-loss batch means, then .mean()."""
    d = dice_loss(pred, target, smooth, dice_eps)
    f = focal_loss(pred, target, gamma, focal_eps)
    return torch.stack([d.mean(), f.mean()]).mean()

"""TEST INFRASTRUCTURE ONLY -- stock-torch (CPU) restatement of the reference's multi-branch
U-Net forward (adell_mri/modules/segmentation/unet.py:1113-1253) for 3-D inputs, driven by a
reference-keyed ``state_dict``: one encoder per branch (:999-1041), concurrent squeeze-and-excite
merges (adell_mri/modules/layers/self_attention.py:21-150) weighted by the branch weights
(:1160-1207), then the plain U-Net decoder. Pinned against outputs of the real reference by
tests/test_oracle_golden.py (fixture brunet3d_two_branch, oracle/make_golden.py). The product
never imports this file."""
import torch
import torch.nn.functional as F

from .unet import UNetOracle


class BrUNetOracle(UNetOracle):
    def cse(self, key, x):
        """ConcurrentSqueezeAndExcite3d: x * sigmoid(conv 1x1x1 C -> 1) + x * sigmoid(MLP(mean))."""
        sd = self.sd
        s = torch.sigmoid(F.conv3d(x, sd[key + ".spatial.op.0.weight"], sd[key + ".spatial.op.0.bias"]))
        m = x.flatten(2).mean(-1)
        h = F.relu(F.linear(m, sd[key + ".channel.op.0.weight"], sd[key + ".channel.op.0.bias"]))
        c = torch.sigmoid(F.linear(h, sd[key + ".channel.op.2.weight"], sd[key + ".channel.op.2.bias"]))
        return x * s + x * c[:, :, None, None, None]

    def merged(self, xs, weights=None):
        """(per-level merged skip tensors, merged bottleneck)."""
        nb = len(xs)
        if weights is None:
            weights = [torch.ones(xs[0].shape[0]) for _ in xs]
        wv = [w.float()[:, None, None, None, None] for w in weights]
        w_sum = sum(wv)
        pre, bott = [], []
        for b in range(nb):
            enc, cur = self.encode(f"encoders.{b}", xs[b])
            pre.append([e * wv[b] for e in enc])
            bott.append(cur * wv[b])
        L = len(self.cfg["depth"])
        bottleneck = sum(self.cse(f"merge_ops.{L - 1}.{b}", bott[b]) / w_sum for b in range(nb))
        skips = [sum(self.cse(f"merge_ops.{i}.{b}", pre[b][i]) / w_sum for b in range(nb))
                 for i in range(L)]
        return skips, bottleneck

    def forward(self, xs, weights=None, return_logits=True):
        skips, bottleneck = self.merged(xs, weights)
        return self.decode(skips, bottleneck, return_logits)

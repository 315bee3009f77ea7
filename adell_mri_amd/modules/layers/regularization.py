"""Regularisation / normalisation layers named by ``norm_fn_dict`` that are not
stock torch modules (mirror of adell_mri/modules/layers/regularization.py)."""
import torch


class LayerNormChannelsFirst(torch.nn.Module):
    """LayerNorm over the channel axis of an [N, C, ...] tensor
    (adell_mri/modules/layers/regularization.py:95-121)."""

    def __init__(self, normalized_shape, eps=1e-6):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(normalized_shape))
        self.bias = torch.nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        """NDHWC memory makes the channel axis the row: one short-row LayerNorm kernel."""
        from ... import functional as HF
        from ... import ops

        if x.dim() == 4:       # 2-D activation: a depth-1 volume
            return self.forward(x.unsqueeze(2)).squeeze(2)
        if x.dim() != 5:
            raise NotImplementedError("HIP LayerNormChannelsFirst: 4-D / 5-D activations only")
        rows = ops.ndhwc(x).permute(0, 2, 3, 4, 1)            # [N, D, H, W, C], contiguous
        return HF.layer_norm(rows, self.weight, self.bias, self.eps).permute(0, 4, 1, 2, 3)


class UOut(torch.nn.Module):
    """U-out (adell_mri/modules/layers/regularization.py:11-57)."""

    def __init__(self, beta: float = 0.0):
        super().__init__()
        self.beta = beta

    def forward(self, X):
        if not (self.training and self.beta != 0.0):
            return X
        from ... import functional as HF

        # X + X * r, r ~ U(-beta, beta) drawn per (item, channel): one broadcast-scale kernel
        r = torch.rand(X.shape[:2], device=X.device) * (2.0 * self.beta) - self.beta
        return HF.scale_per_item_channel(X, 1.0 + r)


class ChannelDropout(torch.nn.Module):
    """Drops whole channels (axis 1) per batch item, without rescaling
    (adell_mri/modules/layers/regularization.py:230-261; the ViT's patch erasing applies it to
    [B, tokens, features]). The mask is drawn on the host exactly as the reference draws it
    (``torch.rand([B, C])``), so a seeded run erases the same channels."""

    def __init__(self, dropout_prob: float, channel_axis: int = 1):
        super().__init__()
        self.dropout_prob = dropout_prob
        self.channel_axis = channel_axis
        if channel_axis != 1:
            raise NotImplementedError("HIP ChannelDropout: channel_axis 1 only")

    def forward(self, X):
        if not (self.dropout_prob > 0 and self.training is True):
            return X
        from ... import functional as HF

        keep = (torch.rand([X.shape[0], X.shape[1]]) > self.dropout_prob).float().to(X.device)
        return HF.scale_per_item_channel(X, keep)


class LayerNorm(torch.nn.Module):
    """LayerNorm over the channel axis for channels_last ([..., C]) or channels_first
    ([N, C, ...]) tensors (adell_mri/modules/layers/regularization.py:60-92; weight and
    bias are [1, C] as in the reference). On the HIP path a channels_first activation is
    NDHWC in memory, so both formats are the same row LayerNorm kernel."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones([1, normalized_shape]))
        self.bias = torch.nn.Parameter(torch.zeros([1, normalized_shape]))
        self.eps = eps
        self.data_format = data_format
        if self.data_format not in ["channels_last", "channels_first"]:
            raise NotImplementedError
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        from ... import functional as HF
        from ... import ops

        w, b = self.weight.reshape(-1), self.bias.reshape(-1)
        if self.data_format == "channels_last":
            return HF.layer_norm(x, w, b, self.eps)
        if x.dim() == 4:   # 2-D network: depth-1 volume
            return self.forward(x.unsqueeze(2)).squeeze(2)
        if x.dim() != 5:
            raise NotImplementedError("channels_first LayerNorm: 4-D / 5-D activations only")
        xr = ops.ndhwc(x).permute(0, 2, 3, 4, 1)            # [N, D, H, W, C] contiguous view
        return HF.layer_norm(xr, w, b, self.eps).permute(0, 4, 1, 2, 3)

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
        raise NotImplementedError("LayerNormChannelsFirst: HIP kernel lands with the UNETR/ConvNeXt rows")


class UOut(torch.nn.Module):
    """U-out (adell_mri/modules/layers/regularization.py:11-57)."""

    def __init__(self, beta: float = 0.0):
        super().__init__()
        self.beta = beta

    def forward(self, X):
        if self.training and self.beta != 0.0:
            raise NotImplementedError("UOut: HIP kernel not implemented yet")
        return X

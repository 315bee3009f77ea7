"""NT-Xent loss of SimCLR (mirror of adell_mri/modules/self_supervised/losses/ntxent.py:11-46)."""
import torch

from .... import functional as HF


class NTXentLoss(torch.nn.Module):
    """Rows of [relu(X1); relu(X2)] against each other: cosine similarities / temperature, every
    row's positive is its other view, the denominator excludes the row itself."""

    def __init__(self, temperature: float = 1.0, apply_relu: bool = True):
        super().__init__()
        self.temperature = temperature
        self.apply_relu = apply_relu

    def forward(self, X1: torch.Tensor, X2: torch.Tensor):
        return HF.pair_loss(X1, X2, "ntxent", self.temperature, self.apply_relu)

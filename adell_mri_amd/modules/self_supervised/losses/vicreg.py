"""VICReg loss (mirror of adell_mri/modules/self_supervised/losses/vicreg.py:30-165).

``forward(X1, X2)`` returns ``(lam * inv, mu * var, nu * cov)`` like the reference; the
three terms and their gradients come from one HIP kernel each way (csrc/ssl.hip), which
uses the B x B Gram matrix of the centred embeddings instead of materialising the D x D
covariance matrix: sum(offdiag(C)^2) = ||Xc Xc^T||_F^2 / (B-1)^2 - sum(diag(C)^2).
"""
from typing import Tuple

import torch

from .... import functional as HF


class VICRegLoss(torch.nn.Module):
    def __init__(self, min_var: float = 1.0, eps: float = 1e-4, lam: float = 25.0,
                 mu: float = 25.0, nu: float = 0.1):
        super().__init__()
        self.min_var = min_var
        self.eps = eps
        self.lam = lam
        self.mu = mu
        self.nu = nu

    def flatten_if_necessary(self, x):
        """[B, C, *spatial] feature maps -> [B, C] spatial means (vicreg.py:138-141), on the
        channel-statistics kernel."""
        if len(x.shape) > 2:
            if x.dim() == 5:
                return HF.channel_mean(x)
            if x.dim() == 4:
                return HF.channel_mean(x.unsqueeze(2))
            return HF.channel_mean(x.unsqueeze(2).unsqueeze(2))
        return x

    def vicreg_loss(self, X1: torch.Tensor, X2: torch.Tensor, adj: float = 1.0):
        """(var_loss, cov_loss, inv_loss), unweighted (vicreg.py:112-136)."""
        terms = HF.vicreg_terms(X1, X2, self.min_var, self.eps)
        return terms[1], terms[2] / adj, terms[0]

    def forward(self, X1: torch.Tensor, X2: torch.Tensor
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        var_loss, cov_loss, inv_loss = self.vicreg_loss(self.flatten_if_necessary(X1),
                                                        self.flatten_if_necessary(X2))
        return self.lam * inv_loss, self.mu * var_loss, self.nu * cov_loss

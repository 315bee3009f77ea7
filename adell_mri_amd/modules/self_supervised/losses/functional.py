"""Functional self-supervised losses (mirror of the pieces of
adell_mri/modules/self_supervised/losses/functional.py that SelfSLBasePL.init_loss selects):
one fused HIP kernel pass each way on the two [B, D] embedding batches."""
import torch

from .... import functional as HF


def _embeddings(x: torch.Tensor) -> torch.Tensor:
    return x.flatten(start_dim=1) if x.dim() > 2 else x


def simsiam_loss(x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """-mean_i cos(x1_i, x2_i) (functional.py:138-150)."""
    return HF.pair_loss(_embeddings(x1), _embeddings(x2), "simsiam")


def byol_loss(x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """2 * simsiam_loss + 2 (functional.py:153-164)."""
    return HF.pair_loss(_embeddings(x1), _embeddings(x2), "byol")

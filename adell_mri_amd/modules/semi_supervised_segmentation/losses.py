"""Losses of the semi-supervised U-Net (adell_mri/modules/semi_supervised_segmentation/losses.py).
Built: ``LocalContrastiveLoss`` (the one the network factory wires, network_factories.py:602-620)
as one fused HIP kernel each way. The anchor / anatomical / nearest-neighbour / pseudo-label
variants of that file are not reached by ``UNetContrastiveSemiSL.training_step`` and raise."""
import numpy as np
import torch

from ... import functional as HF


class LocalContrastiveLoss(torch.nn.Module):
    """losses.py:480-526: per voxel, the B x B cosine similarities between the features of view 2
    (rows) and view 1 (columns) / temperature, soft-maxed over the columns; the loss of item i is
    the mean over voxels of -log(max(softmax[i, i], 1e-8)). Returns [B]."""

    def __init__(self, temperature: float = 0.1, seed: int = 42):
        super().__init__()
        self.temperature = temperature
        self.seed = seed
        self.rng = np.random.default_rng(seed)
        self.eps = torch.as_tensor(1e-8)

    def forward(self, X_1: torch.Tensor, X_2: torch.Tensor,
                anchors: torch.Tensor = None) -> torch.Tensor:
        if X_1.dim() == 4:  # 2-D network: depth-1 volume
            X_1, X_2 = X_1.unsqueeze(2), X_2.unsqueeze(2)
        return HF.loco_loss(X_1, X_2, self.temperature, float(self.eps))


def _not_built(name, where):
    class _Raiser(torch.nn.Module):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError(
                f"{name} ({where}) has no HIP kernel: UNetContrastiveSemiSL.training_step only "
                f"reaches LocalContrastiveLoss")
    _Raiser.__name__ = name
    return _Raiser


LocalContrastiveLossWithAnchors = _not_built("LocalContrastiveLossWithAnchors", "losses.py:529-585")
AnatomicalContrastiveLoss = _not_built("AnatomicalContrastiveLoss", "losses.py:77-251")
NearestNeighbourLoss = _not_built("NearestNeighbourLoss", "losses.py:254-444")
PseudoLabelCrossEntropy = _not_built("PseudoLabelCrossEntropy", "losses.py:447-477")

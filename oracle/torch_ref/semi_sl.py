"""CPU restatement (stock torch, explicit arithmetic) of the local contrastive loss of the
reference's semi-supervised U-Net (TEST INFRASTRUCTURE ONLY;
adell_mri/modules/semi_supervised_segmentation/losses.py:498-526), pinned to
tests/golden/loco_loss.npz by tests/test_oracle_golden.py."""
import torch


def local_contrastive_loss(x1, x2, temperature=0.1, eps=1e-8, cos_eps=1e-8):
    """x1, x2: [B, C, *spatial]. [B] losses: per voxel s, z[i, j] = cos(x2[i, :, s], x1[j, :, s])
    / T (each norm clamped at cos_eps, as torch.nn.functional.cosine_similarity does),
    p = softmax over j, loss[i] = mean_s -log(max(p[i, i], eps))."""
    a = x2.flatten(2)                       # [B, C, S] rows i
    b = x1.flatten(2)                       # [B, C, S] columns j
    an = a / a.norm(dim=1, keepdim=True).clamp_min(cos_eps)
    bn = b / b.norm(dim=1, keepdim=True).clamp_min(cos_eps)
    z = torch.einsum("ics,jcs->ijs", an, bn) / temperature
    p = torch.softmax(z, dim=1)
    diag = torch.diagonal(p, dim1=0, dim2=1).permute(1, 0)   # [B, S]
    return -torch.log(torch.clamp_min(diag, eps)).mean(-1)

"""TEST INFRASTRUCTURE ONLY: torch-CPU restatement of the arithmetic of the MONAI transforms the
reference composes in transform_factory/augmentations.py:19-178 (get_augmentations_unet), used to
check adell_mri_amd/utils/augment.py and csrc/augment.hip.

PARITY UNPINNED: MONAI (pinned by the reference's pyproject: monai >= 1.3) is not installed in the
build container and the reference holds no golden vectors for its augmentation chain, so these
functions restate the published definitions of the transforms:
  * AdjustContrast   (monai/transforms/intensity/array.py): ((x - min) / (max - min + 1e-7)) ** gamma
                     * (max - min) + min over the whole image
  * StdShiftIntensity: x + factor * std(x) (population standard deviation of the whole image)
  * RandRicianNoise  : sqrt((x + n1) ** 2 + n2 ** 2), n1, n2 ~ N(0, std)
  * Affine / Resample: src = A (dst - centre) + centre in voxel coordinates, torch grid_sample
                     (align_corners=False), bilinear / nearest, reflection / border / zeros padding
"""
import torch
import torch.nn.functional as F


def adjust_contrast(x, gamma):
    mn, mx = x.min(), x.max()
    rng = mx - mn
    return ((x - mn) / (rng + 1e-7)) ** gamma * rng + mn


def std_shift(x, factor):
    return x + factor * x.std(unbiased=False)


def rician(x, n1, n2):
    return torch.sqrt((x + n1) ** 2 + n2 ** 2)


def affine_resample(x, theta, linear=True, pad_mode="reflection"):
    """x: [N, C, D, H, W]; theta: [N, 3, 4] over centred (z, y, x) voxel coordinates."""
    N, C, D, H, W = x.shape
    size = torch.tensor([D, H, W], dtype=torch.float64)
    centre = (size - 1) / 2
    zz, yy, xx = torch.meshgrid(*[torch.arange(s, dtype=torch.float64) for s in (D, H, W)],
                                indexing="ij")
    dst = torch.stack([zz, yy, xx], -1) - centre                       # [D, H, W, 3]
    out = []
    for n in range(N):
        A, t = theta[n, :, :3].double(), theta[n, :, 3].double()
        src = dst @ A.T + t + centre                                   # voxel coordinates (z, y, x)
        norm = (2 * src + 1) / size - 1                                # align_corners=False
        grid = norm.flip(-1).unsqueeze(0)                              # grid_sample wants (x, y, z)
        out.append(F.grid_sample(x[n:n + 1].double(), grid, mode="bilinear" if linear else "nearest",
                                 padding_mode=pad_mode, align_corners=False))
    return torch.cat(out).float()

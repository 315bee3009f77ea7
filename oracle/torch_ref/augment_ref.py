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


# ---- round 4: the rest of the factory's vocabulary (still PARITY UNPINNED, see the header) ------------
#   * GaussianSmooth    (monai/networks/layers/simplelayers.py: GaussianFilter, approx="erf",
#                        truncated=4.0): separable convolution with the erf taps, zero padding
#   * RandBiasField     (monai/transforms/intensity/array.py): x * exp(leggrid3d over
#                        linspace(-1, 1, size) of the coefficient cube)
#   * GridDistortion    (monai/transforms/spatial/array.py): per-axis piecewise-linear coordinate ramps,
#                        resampled with border padding; coordinates taken as voxel indices
#   * GibbsNoise        (monai/transforms/intensity/array.py): fftshift(fftn(x)) masked to the sphere
#                        of radius (1 - alpha) * max(size) * sqrt(2) / 2 about (size - 1) / 2, then
#                        ifftn(ifftshift(.)).real
#   * SimulateLowResolution: nearest resize by the zoom factor, trilinear resize back
def gaussian_blur(x, taps_zyx):
    """x: [C, D, H, W]; taps_zyx: three 1-D tap arrays (z, y, x)."""
    out = x.unsqueeze(0).double()
    C = x.shape[0]
    for axis, taps in enumerate(taps_zyx):
        t = torch.as_tensor(taps, dtype=torch.float64)
        shape = [1, 1, 1, 1, 1]
        shape[2 + axis] = t.numel()
        w = t.view(shape).repeat(C, 1, 1, 1, 1)
        pad = [0, 0, 0]
        pad[axis] = (t.numel() - 1) // 2
        out = F.conv3d(out, w, padding=pad, groups=C)
    return out[0].float()


def bias_field(x, cube):
    """x: [C, D, H, W]; cube: [4, 4, 4] Legendre coefficients."""
    import numpy as np
    D, H, W = x.shape[1:]
    coords = [np.linspace(-1.0, 1.0, n, dtype=np.float64) if n > 1 else np.array([-1.0])
              for n in (D, H, W)]
    field = np.polynomial.legendre.leggrid3d(coords[0], coords[1], coords[2],
                                             np.asarray(cube, dtype=np.float64))
    return (x.double() * torch.from_numpy(np.exp(field))).float()


def lut_resample(x, luts, linear=True):
    """x: [C, D, H, W]; luts: (lz, ly, lx) input voxel coordinates per output index; border."""
    size = torch.tensor(x.shape[1:], dtype=torch.float64)
    lz, ly, lx = [torch.as_tensor(v, dtype=torch.float64) for v in luts]
    zz, yy, xx = torch.meshgrid(lz, ly, lx, indexing="ij")
    src = torch.stack([zz, yy, xx], -1)
    src = torch.minimum(torch.maximum(src, torch.zeros(3, dtype=torch.float64)), size - 1)
    norm = (2 * src + 1) / size - 1
    grid = norm.flip(-1).unsqueeze(0)
    return F.grid_sample(x.unsqueeze(0).double(), grid, mode="bilinear" if linear else "nearest",
                         padding_mode="border", align_corners=False)[0].float()


def gibbs(x, alpha):
    """x: [C, D, H, W]."""
    import numpy as np
    dims = (1, 2, 3)
    k = torch.fft.fftshift(torch.fft.fftn(x.double(), dim=dims), dim=dims)
    shape = x.shape[1:]
    r = (1 - alpha) * max(shape) * np.sqrt(2) / 2.0
    grids = torch.meshgrid(*[torch.arange(n, dtype=torch.float64) - (n - 1) / 2 for n in shape],
                           indexing="ij")
    mask = torch.sqrt(sum(g ** 2 for g in grids)) <= r
    return torch.fft.ifftn(torch.fft.ifftshift(k * mask, dim=dims), dim=dims).real.float()


def low_resolution(x, zoom):
    """x: [C, D, H, W]."""
    full = tuple(x.shape[1:])
    small = tuple(int(round(n * zoom)) for n in full)
    down = F.interpolate(x.unsqueeze(0), size=small, mode="nearest")
    return F.interpolate(down, size=full, mode="trilinear", align_corners=False)[0]

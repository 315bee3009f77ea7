"""Device-side batch augmentation: the arithmetic of
`get_augmentations_unet` (adell_mri/transform_factory/augmentations.py:19-178) applied to batches
that already live in HBM, by the HIP passes of csrc/augment.hip -- no worker processes, no host
round trip of the volumes. The reference builds a MONAI transform chain; MONAI is not a
dependency here, so this is not an API mirror of those classes but of the FACTORY: the same
`augment` vocabulary, the same probabilities and parameter ranges, the same order.

Built (pure arithmetic on the device; the whole vocabulary of the factory since round 4):
  "distort"    RandGridDistortiond(distort_limit=0.05, 5 cells per axis): per-axis piecewise-linear
               coordinate tables, trilinear for the image keys / nearest for the others, border padding
  "intensity"  RandAdjustContrastd(gamma=(0.5, 1.5)) then RandStdShiftIntensityd(factors=0.1)
  "blur"       RandGaussianSmoothd (MONAI defaults: sigma ~ U(0.25, 1.5) per axis, prob 0.1, "erf"
               taps truncated at 4 sigma): three 1-D passes, zero padding
  "noise"      RandRicianNoised(std=0.02) (Philox + Box-Muller noise generated in the kernel) and
               RandGibbsNoised(alpha=(0.3, 0.6)): the spectrum (direct DFT per line, any axis length)
               zeroed outside radius (1 - alpha) max(size) sqrt(2) / 2 about its shifted centre
  "rbf"        RandBiasFieldd(degree=3, coeff ~ U(0, 0.1)) on the T2 keys: x * exp(Legendre field)
  "affine"     RandAffined(rotate_range=[pi/8, pi/8, pi/16]), bilinear for the image keys, nearest
               for the others, reflection padding
  "shear"      RandAffined(shear_range=((0.9, 1.1),) * 3)
  "lowres"     RandSimulateLowResolutiond(zoom_range=[0.8, 1.2]): nearest resize by the zoom factor,
               trilinear resize back
  "flip"       RandFlipd(prob=0.25) per axis of `flip_axis`
  "trivial"    OneOf(the above) with probability 1, flips kept outside the choice
PARITY UNPINNED for all of them: MONAI is not installed in the build container and the reference
holds no vectors for its augmentation chain; the kernels follow the published definitions of the
MONAI transforms (restated in oracle/torch_ref/augment_ref.py). Random crops are
`utils.batching.DeviceCropSampler`.

Random decisions and parameters are drawn on the host from a seeded numpy RandomState (as MONAI
does); one draw per batch ITEM, applied to every key of the item. Checked against the torch-CPU
restatement `oracle/torch_ref/augment_ref.py` (tests/test_augment.py)."""
from typing import Dict, Sequence

import numpy as np
import torch

from .. import ops

_BUILT = ("intensity", "noise", "rbf", "affine", "shear", "flip", "blur", "distort", "lowres",
          "trivial")


def gaussian_taps(sigma: float, truncated: float = 4.0) -> np.ndarray:
    """MONAI's gaussian_1d(sigma, truncated, approx="erf"): integral of the Gaussian over each
    voxel, tail = int(max(sigma * truncated, 0.5) + 0.5), not renormalised."""
    from math import erf
    tail = int(max(float(sigma) * truncated, 0.5) + 0.5)
    t = 0.70710678 / abs(float(sigma))
    return np.array([max(0.5 * (erf(t * (k + 0.5)) - erf(t * (k - 0.5))), 0.0)
                     for k in range(-tail, tail + 1)], dtype=np.float32)


def distortion_table(size: int, steps: Sequence[float], num_cells: int = 5) -> np.ndarray:
    """MONAI GridDistortion's per-axis coordinate ramp (monai/transforms/spatial/array.py): the axis
    is cut into ``num_cells`` cells of size // num_cells voxels, cell i is stretched by steps[i],
    the last (partial) cell ends at the axis end; returns the input coordinate, in voxels, of every
    output index."""
    ranges = np.zeros(size, dtype=np.float64)
    cell = size // num_cells
    prev = 0.0
    for idx in range(num_cells + 1):
        start = int(idx * cell)
        end = start + cell
        if end > size:
            end, cur = size, float(size)
        else:
            cur = prev + cell * steps[idx]
        if end > start:
            ranges[start:end] = np.linspace(prev, cur, end - start)
        prev = cur
    return ranges.astype(np.float32)


def bias_coefficients(coeff: Sequence[float], degree: int = 3) -> np.ndarray:
    """The dense (degree + 1)^3 Legendre coefficient cube of MONAI's RandBiasField for three
    dimensions: the drawn coefficients fill the entries with i + j + k <= degree in (i, j, k)
    order."""
    cube = np.zeros((4, 4, 4), dtype=np.float32)
    pts = [(i, j, k) for i in range(degree + 1) for j in range(degree + 1 - i)
           for k in range(degree + 1 - i - j)]
    for (i, j, k), c in zip(pts, coeff):
        cube[i, j, k] = c
    return cube
_VALID = ("intensity", "noise", "rbf", "affine", "shear", "flip", "blur", "distort", "lowres",
          "trivial")


def rotation_matrix(rz: float, ry: float, rx: float) -> np.ndarray:
    """Rotation about the first, second and third spatial axis in turn (MONAI's create_rotate for
    three dimensions: R_first @ R_second @ R_third), as a 3 x 3 matrix over (z, y, x)."""
    c, s = np.cos, np.sin
    r0 = np.array([[1, 0, 0], [0, c(rz), -s(rz)], [0, s(rz), c(rz)]])
    r1 = np.array([[c(ry), 0, s(ry)], [0, 1, 0], [-s(ry), 0, c(ry)]])
    r2 = np.array([[c(rx), -s(rx), 0], [s(rx), c(rx), 0], [0, 0, 1]])
    return r0 @ r1 @ r2


def shear_matrix(coefs: Sequence[float]) -> np.ndarray:
    """MONAI's create_shear for three dimensions; missing coefficients are zero."""
    s = list(coefs) + [0.0] * (6 - len(coefs))
    return np.array([[1, s[0], s[1]], [s[2], 1, s[3]], [s[4], s[5], 1]], dtype=np.float64)


class DeviceAugmenter:
    """``aug = DeviceAugmenter(augment, all_keys, image_keys)``; ``aug(batch)`` takes a dict of
    [N, C, D, H, W] float32 CUDA tensors and returns the augmented dict (new tensors, NDHWC
    memory). ``last_plan`` holds what was drawn for the last batch."""

    def __init__(self, augment: Sequence[str], all_keys: Sequence[str], image_keys: Sequence[str],
                 t2_keys: Sequence[str] = (), flip_axis: Sequence[int] = (0, 1), seed: int = 42):
        for a in augment:
            if a not in _VALID:
                raise NotImplementedError("augment can only contain {}".format(list(_VALID)))
            if a not in _BUILT:
                raise NotImplementedError(
                    f"augmentation {a!r} has no device kernel (built: {list(_BUILT)})")
        self.augment = list(augment)
        self.all_keys = list(all_keys)
        self.image_keys = list(image_keys)
        self.t2_keys = list(t2_keys)
        self.flip_axis = tuple(flip_axis)
        self.trivial = "trivial" in self.augment
        self.prob = 1.0 if self.trivial else 0.2
        self.R = np.random.RandomState(seed)
        self.seed = int(seed)
        self._calls = 0
        self.last_plan = None

    # ---- host side: what happens to each item ----------------------------------------------
    def _transforms(self):
        """The atomic transforms in the order of augmentations.py:52-127 (each "augment" word
        contributes one or two of them; OneOf of "trivial" chooses among the atoms)."""
        atoms = []
        if "distort" in self.augment:
            atoms += ["distort"]
        if "intensity" in self.augment:
            atoms += ["contrast", "stdshift"]
        if "blur" in self.augment:
            atoms += ["blur"]
        if "noise" in self.augment:
            atoms += ["rician", "gibbs"]
        if "rbf" in self.augment and len(self.t2_keys) > 0:
            atoms += ["rbf"]
        if "affine" in self.augment:
            atoms += ["affine"]
        if "shear" in self.augment:
            atoms += ["shear"]
        if "lowres" in self.augment:
            atoms += ["lowres"]
        return atoms

    def plan(self, n_items: int):
        """One dict per batch item: the parameters of the transforms that fire for it."""
        atoms = self._transforms()
        items = []
        for _ in range(n_items):
            if self.trivial:      # Identityd is the first member of the OneOf list
                pick = self.R.randint(len(atoms) + 1)
                chosen = [] if pick == 0 else [atoms[pick - 1]]
            else:
                chosen = [a for a in atoms if a == "blur" or self.R.rand() < self.prob]
            # RandGaussianSmoothd is built without `prob`: MONAI's default 0.1 decides, also inside
            # the OneOf of "trivial" (augmentations.py:78-79)
            if "blur" in chosen and not self.R.rand() < 0.1:
                chosen = [a for a in chosen if a != "blur"]
            it = {}
            if "distort" in chosen:
                it["distort"] = [[float(1.0 + self.R.uniform(-0.05, 0.05)) for _ in range(6)]
                                 for _ in range(3)]
            if "blur" in chosen:
                it["blur"] = [float(self.R.uniform(0.25, 1.5)) for _ in range(3)]
            if "gibbs" in chosen:
                it["gibbs_alpha"] = float(self.R.uniform(0.3, 0.6))
            if "rbf" in chosen:
                it["rbf"] = [float(v) for v in self.R.uniform(0.0, 0.1, 20)]
            if "lowres" in chosen:
                it["lowres"] = float(self.R.uniform(0.8, 1.2))
            if "contrast" in chosen:
                it["gamma"] = float(self.R.uniform(0.5, 1.5))
            if "stdshift" in chosen:
                it["shift"] = float(self.R.uniform(-0.1, 0.1))
            if "rician" in chosen:
                it["noise_std"] = {k: float(self.R.uniform(0.0, 0.02)) for k in self.image_keys}
            if "affine" in chosen:
                ang = [self.R.uniform(-r, r) for r in (np.pi / 8, np.pi / 8, np.pi / 16)]
                it["affine"] = rotation_matrix(*ang)
            if "shear" in chosen:
                it["shear"] = shear_matrix([self.R.uniform(0.9, 1.1) for _ in range(3)])
            it["flips"] = (tuple(a for a in self.flip_axis if self.R.rand() < 0.25)
                           if "flip" in self.augment else ())
            items.append(it)
        return items

    # ---- device side ---------------------------------------------------------------------------
    @staticmethod
    def _theta(mats, device):
        rows = np.zeros((len(mats), 12), dtype=np.float32)
        for i, m in enumerate(mats):
            rows[i].reshape(3, 4)[:, :3] = m
        return torch.from_numpy(rows).to(device)

    def _resample(self, batch, mats, which):
        """One affine pass over every key for the items whose matrix is not None."""
        if all(m is None for m in mats):
            return batch
        full = [np.eye(3) if m is None else m for m in mats]
        out = {}
        for k, v in batch.items():
            theta = self._theta(full, v.device)
            linear = k in self.image_keys and which != "flip"
            out[k] = ops.affine_sample(v, theta, linear=linear,
                                       pad_mode="zeros" if which == "flip" else "reflection")
        return out

    def _distort(self, batch, plan):
        if not any("distort" in it for it in plan):
            return batch
        out = {}
        for k, v in batch.items():
            D, H, W = v.shape[2:]
            rows = np.zeros((len(plan), D + H + W), dtype=np.float32)
            for i, it in enumerate(plan):
                steps = it.get("distort", [[1.0] * 6] * 3)
                rows[i] = np.concatenate([distortion_table(n, st) for n, st in zip((D, H, W), steps)])
            out[k] = ops.axis_lut_sample(v, torch.from_numpy(rows).to(v.device),
                                         linear=k in self.image_keys)
        return out

    def _blur(self, x, plan):
        if not any("blur" in it for it in plan):
            return x
        for axis in range(3):
            taps = [gaussian_taps(it["blur"][axis]) if "blur" in it else np.ones(1, np.float32)
                    for it in plan]
            R = max((len(t) - 1) // 2 for t in taps)
            rows = np.zeros((len(plan), 2 * R + 1), dtype=np.float32)
            for i, t in enumerate(taps):
                r = (len(t) - 1) // 2
                rows[i, R - r:R + r + 1] = t
            x = ops.axis_filter(x, torch.from_numpy(rows).to(x.device), axis)
        return x

    def _gibbs(self, x, plan):
        if not any("gibbs_alpha" in it for it in plan):
            return x
        size = max(x.shape[2:])
        # an item that does not fire keeps its whole spectrum: radius past the farthest corner
        rad = [(1.0 - it["gibbs_alpha"]) * size * np.sqrt(2.0) / 2.0 if "gibbs_alpha" in it
               else 2.0 * size for it in plan]
        return ops.gibbs_lowpass(x, torch.tensor(rad, dtype=torch.float32, device=x.device))

    def _bias(self, x, plan):
        if not any("rbf" in it for it in plan):
            return x
        rows = np.zeros((len(plan), 64), dtype=np.float32)
        for i, it in enumerate(plan):
            if "rbf" in it:
                rows[i] = bias_coefficients(it["rbf"]).reshape(-1)
        return ops.bias_field(x, torch.from_numpy(rows).to(x.device))

    def _lowres(self, x, plan):
        if not any("lowres" in it for it in plan):
            return x
        outs = []
        for i, it in enumerate(plan):
            xi = x[i:i + 1]
            if "lowres" in it:
                full = tuple(xi.shape[2:])
                small = tuple(int(round(n * it["lowres"])) for n in full)
                xi = ops.resize_linear(ops.interp_nearest(xi, small), full)
            outs.append(ops.ndhwc(xi))
        return ops.ndhwc(torch.cat(outs, 0))

    def __call__(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        keys = [k for k in self.all_keys if k in batch]
        batch = {k: ops.ndhwc(batch[k]) for k in keys}
        first = batch[keys[0]]
        if not first.is_cuda or first.dtype != torch.float32 or first.dim() != 5:
            raise ops.AdellHipError("DeviceAugmenter: [N, C, D, H, W] float32 CUDA tensors only")
        n = first.shape[0]
        plan = self.plan(n)
        self.last_plan = plan
        self._calls += 1
        batch = self._distort(batch, plan)
        # intensity: contrast pass (needs min / max of the item), then shift (needs the std of what
        # the contrast pass produced) and noise together
        for k in self.image_keys:
            if k not in batch:
                continue
            x = batch[k]
            if any("gamma" in it for it in plan):
                st = ops.item_stats(x).cpu().numpy()          # [n, 4]: tiny, wanted on the host
                rows = np.zeros((n, 8), dtype=np.float32)
                for i, it in enumerate(plan):
                    if "gamma" in it:
                        rows[i, :3] = (st[i, 0], st[i, 1] - st[i, 0], it["gamma"])
                x = ops.aug_intensity(x, torch.from_numpy(rows).to(x.device))
            rows = np.zeros((n, 8), dtype=np.float32)
            if any("shift" in it for it in plan):
                st = ops.item_stats(x).cpu().numpy()
                for i, it in enumerate(plan):
                    if "shift" in it:
                        rows[i, 3] = it["shift"] * st[i, 3]
            for i, it in enumerate(plan):
                rows[i, 4] = it.get("noise_std", {}).get(k, 0.0)
            # (the factory's order: contrast, shift, BLUR, Rician, Gibbs -- augmentations.py:64-95)
            if np.any(rows[:, 3] != 0.0):
                srows = rows.copy()
                srows[:, 4] = 0.0
                x = ops.aug_intensity(x, torch.from_numpy(srows).to(x.device))
                rows[:, 3] = 0.0
            x = self._blur(x, plan)
            if np.any(rows[:, 4] > 0.0):
                x = ops.aug_intensity(x, torch.from_numpy(rows).to(x.device), seed=self.seed,
                                      rng_offset=self._calls * 131 + self.image_keys.index(k))
            x = self._gibbs(x, plan)
            if k in self.t2_keys:
                x = self._bias(x, plan)
            batch[k] = x
        batch = self._resample(batch, [it.get("affine") for it in plan], "affine")
        batch = self._resample(batch, [it.get("shear") for it in plan], "shear")
        for k in self.image_keys:
            if k in batch:
                batch[k] = self._lowres(batch[k], plan)
        flips = []
        for it in plan:
            if it["flips"]:
                d = np.ones(3)
                for a in it["flips"]:
                    d[a] = -1.0
                flips.append(np.diag(d))
            else:
                flips.append(None)
        return self._resample(batch, flips, "flip")

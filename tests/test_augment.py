"""Device-side augmentation (factory: transform_factory/augmentations.py:19-178):
the HIP passes of csrc/augment.hip against the torch-CPU restatement (oracle/torch_ref/augment_ref.py)
and the factory-level driver's plan / composition."""
import numpy as np
import pytest
import torch

from adell_mri_amd.utils.augment import DeviceAugmenter, rotation_matrix, shear_matrix
from oracle.torch_ref import augment_ref as ref


def test_factory_vocabulary_and_plan_statistics():
    with pytest.raises(NotImplementedError):
        DeviceAugmenter(["sharpen"], ["image"], ["image"])
    for word in ("blur", "distort", "lowres", "rbf"):         # the whole vocabulary is built (round 4)
        DeviceAugmenter([word], ["image"], ["image"])
    assert DeviceAugmenter(["rbf"], ["image"], ["image"])._transforms() == []   # no T2 keys: nothing
    full = DeviceAugmenter(["intensity", "noise", "rbf", "affine", "shear", "blur", "distort",
                            "lowres"], ["image"], ["image"], t2_keys=["image"], seed=5)
    # the order of augmentations.py:52-127
    assert full._transforms() == ["distort", "contrast", "stdshift", "blur", "rician", "gibbs",
                                  "rbf", "affine", "shear", "lowres"]
    plan = full.plan(4000)
    frac = lambda key: np.mean([key in it for it in plan])  # noqa: E731
    for key in ("distort", "gibbs_alpha", "rbf", "lowres"):
        assert 0.17 < frac(key) < 0.23, key
    assert 0.07 < frac("blur") < 0.13                        # MONAI's default prob of 0.1
    al = [it["gibbs_alpha"] for it in plan if "gibbs_alpha" in it]
    assert 0.3 <= min(al) and max(al) <= 0.6
    aug = DeviceAugmenter(["intensity", "noise", "affine", "shear", "flip"], ["image", "mask"],
                          ["image"], seed=1)
    plan = aug.plan(4000)
    frac = lambda key: np.mean([key in it for it in plan])  # noqa: E731
    for key in ("gamma", "shift", "noise_std", "affine", "shear"):
        assert 0.17 < frac(key) < 0.23, key                  # prob 0.2 each, independent
    assert 0.22 < np.mean([0 in it["flips"] for it in plan]) < 0.28
    gam = [it["gamma"] for it in plan if "gamma" in it]
    assert 0.5 <= min(gam) and max(gam) <= 1.5
    tri = DeviceAugmenter(["trivial", "intensity", "affine"], ["image"], ["image"], seed=2)
    plan = tri.plan(3000)
    fired = [sum(k in it for k in ("gamma", "shift", "affine")) for it in plan]
    assert max(fired) == 1 and 0.2 < np.mean([f == 0 for f in fired]) < 0.3   # OneOf incl. identity
    R = rotation_matrix(0.3, -0.2, 0.1)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(R), 1.0)
    assert np.allclose(shear_matrix([0.5, 0.25, 2.0]), [[1, 0.5, 0.25], [2.0, 1, 0], [0, 0, 1]])


@pytest.mark.gpu
def test_item_stats_and_intensity_pass_match_restatement(cuda):
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(0)
    x = torch.rand(3, 2, 9, 10, 11, generator=g) * 3.0 - 0.5
    xd = ops.ndhwc(x.to(cuda))
    st = ops.item_stats(xd).cpu()
    for i in range(3):
        want = torch.tensor([x[i].min(), x[i].max(), x[i].mean(), x[i].std(unbiased=False)])
        assert torch.allclose(st[i], want, rtol=1e-5, atol=1e-6)
    gammas, factors = [0.6, 0.0, 1.4], [0.05, -0.1, 0.0]
    rows = torch.zeros(3, 8)
    for i in range(3):
        rows[i, :3] = torch.tensor([st[i, 0], st[i, 1] - st[i, 0], gammas[i]])
    y = ops.aug_intensity(xd, rows.to(cuda))
    st2 = ops.item_stats(y).cpu()
    rows2 = torch.zeros(3, 8)
    rows2[:, 3] = torch.tensor(factors) * st2[:, 3]
    z = ops.aug_intensity(y, rows2.to(cuda)).cpu()
    for i in range(3):
        want = x[i] if gammas[i] == 0.0 else ref.adjust_contrast(x[i], gammas[i])
        want = ref.std_shift(want, factors[i])
        assert torch.allclose(z[i], want, rtol=2e-5, atol=2e-5), i


@pytest.mark.gpu
def test_rician_noise_statistics_and_determinism(cuda):
    from adell_mri_amd import ops

    x = torch.zeros(2, 1, 32, 32, 32, device=cuda)
    rows = torch.zeros(2, 8)
    rows[:, 4] = torch.tensor([0.02, 0.0])
    a = ops.aug_intensity(x, rows.to(cuda), seed=7, rng_offset=1)
    b = ops.aug_intensity(x, rows.to(cuda), seed=7, rng_offset=1)
    c = ops.aug_intensity(x, rows.to(cuda), seed=7, rng_offset=2)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.equal(a[1], x[1])                                  # std 0: untouched
    r = a[0].flatten().double().cpu()                               # Rayleigh(sigma) on a zero image
    sigma = 0.02
    assert abs(float(r.mean()) - sigma * np.sqrt(np.pi / 2)) < 0.01 * sigma
    assert abs(float((r * r).mean()) - 2 * sigma ** 2) < 0.02 * sigma ** 2
    # on a non-zero image: sqrt((x + n1)^2 + n2^2) >= |x| - |n|, and the mean bias is ~ sigma^2 / 2x
    y = ops.aug_intensity(torch.ones_like(x), rows.to(cuda), seed=3, rng_offset=5)[0]
    assert abs(float(y.mean()) - (1.0 + sigma ** 2 / 2)) < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("linear", [True, False])
@pytest.mark.parametrize("pad_mode", ["reflection", "border", "zeros"])
def test_affine_resampling_matches_grid_sample(cuda, linear, pad_mode):
    from adell_mri_amd import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 2, 12, 14, 10, generator=g)
    mats = [rotation_matrix(0.35, -0.2, 0.15), shear_matrix([0.3, -0.2, 0.1]),
            np.diag([1.0, -1.0, 1.0])]
    theta = torch.zeros(3, 3, 4)
    for i, m in enumerate(mats):
        theta[i, :, :3] = torch.from_numpy(m).float()
    theta[0, :, 3] = torch.tensor([0.4, -1.3, 0.25])
    got = ops.affine_sample(x.to(cuda), theta.reshape(3, 12).to(cuda), linear=linear,
                            pad_mode=pad_mode).cpu()
    want = ref.affine_resample(x, theta, linear=linear, pad_mode=pad_mode)
    if linear:
        assert torch.allclose(got, want, rtol=1e-4, atol=2e-5)
    else:
        # nearest: a coordinate within rounding of a half-way point may pick the other neighbour
        assert float((got != want).float().mean()) < 2e-3
    assert torch.equal(got[2], x[2].flip(2)) or linear            # an exact flip in nearest mode


@pytest.mark.gpu
def test_augmenter_composes_the_passes(cuda):
    keys = ["image", "mask"]
    aug = DeviceAugmenter(["intensity", "affine", "flip"], keys, ["image"], flip_axis=(0, 1), seed=5)
    g = torch.Generator().manual_seed(1)
    batch = {"image": torch.rand(6, 2, 16, 16, 12, generator=g).to(cuda),
             "mask": (torch.rand(6, 1, 16, 16, 12, generator=g) > 0.7).float().to(cuda)}
    # force every transform for item 0, none for item 1
    forced = aug.plan(6)
    forced[0] = {"gamma": 0.7, "shift": 0.05, "affine": rotation_matrix(0.2, 0.1, -0.1), "flips": (1,)}
    forced[1] = {"flips": ()}
    aug.plan = lambda n: forced
    out = aug(batch)
    x, m = batch["image"].cpu(), batch["mask"].cpu()
    assert torch.equal(out["image"][1].cpu(), x[1]) and torch.equal(out["mask"][1].cpu(), m[1])
    want = ref.std_shift(ref.adjust_contrast(x[0], 0.7), 0.05)[None]
    th = torch.zeros(1, 3, 4)
    th[0, :, :3] = torch.from_numpy(forced[0]["affine"]).float()
    want = ref.affine_resample(want, th, linear=True, pad_mode="reflection")
    want = want.flip(3)                                            # spatial axis 1
    assert torch.allclose(out["image"][0].cpu(), want[0], rtol=1e-4, atol=5e-5)
    wm = ref.affine_resample(m[0:1], th, linear=False, pad_mode="reflection").flip(3)
    assert float((out["mask"][0].cpu() != wm[0]).float().mean()) < 5e-3
    assert set(np.unique(out["mask"].cpu().numpy())) <= {0.0, 1.0}   # labels stay labels


@pytest.mark.gpu
def test_blur_bias_distort_gibbs_lowres_match_restatement(cuda):
    """Round 4: the remaining transforms of the factory against the torch-CPU restatement of the MONAI
    definitions (parity unpinned: MONAI absent), on a ragged non-power-of-two volume."""
    from adell_mri_amd import ops
    from adell_mri_amd.utils.augment import (bias_coefficients, distortion_table, gaussian_taps)

    g = torch.Generator().manual_seed(3)
    x = torch.rand((2, 2, 12, 20, 18), generator=g)
    xd = ops.ndhwc(x.to(cuda))
    # blur: item 0 smoothed with three sigmas, item 1 untouched (unit impulse)
    sig = (0.4, 1.5, 0.9)
    y = xd
    for axis in range(3):
        t = gaussian_taps(sig[axis])
        R = (len(t) - 1) // 2
        rows = np.zeros((2, 2 * R + 1), dtype=np.float32)
        rows[0] = t
        rows[1, R] = 1.0
        y = ops.axis_filter(y, torch.from_numpy(rows).to(cuda), axis)
    want = ref.gaussian_blur(x[0], [gaussian_taps(s) for s in sig])
    assert float((y[0].cpu() - want).abs().max()) < 1e-5
    assert torch.equal(y[1].cpu(), x[1])
    # bias field
    cube = bias_coefficients(np.linspace(0.01, 0.1, 20))
    rows = np.zeros((2, 64), dtype=np.float32)
    rows[0] = cube.reshape(-1)
    y = ops.bias_field(xd, torch.from_numpy(rows).to(cuda))
    assert float((y[0].cpu() - ref.bias_field(x[0], cube)).abs().max()) < 2e-5
    assert torch.equal(y[1].cpu(), x[1])
    # grid distortion, both interpolation modes
    steps = [[1.03, 0.96, 1.0, 1.05, 0.95, 1.0], [0.97, 1.02, 1.04, 0.98, 1.0, 1.0],
             [1.0, 1.05, 0.95, 1.02, 0.99, 1.0]]
    luts = [distortion_table(n, st) for n, st in zip((12, 20, 18), steps)]
    ident = [np.arange(n, dtype=np.float32) for n in (12, 20, 18)]
    rows = np.stack([np.concatenate(luts), np.concatenate(ident)])
    for linear in (True, False):
        y = ops.axis_lut_sample(xd, torch.from_numpy(rows).to(cuda), linear=linear)
        want = ref.lut_resample(x[0], luts, linear=linear)
        if linear:
            assert float((y[0].cpu() - want).abs().max()) < 2e-5
        else:       # ties of the rounding may fall either way on a handful of voxels
            assert float((y[0].cpu() != want).float().mean()) < 5e-3
        assert torch.equal(y[1].cpu(), x[1])
    # Gibbs: a k-space low-pass on one item, the other keeps its whole spectrum
    alpha = 0.45
    rad = torch.tensor([(1 - alpha) * 20 * np.sqrt(2) / 2, 40.0], dtype=torch.float32, device=cuda)
    y = ops.gibbs_lowpass(xd, rad)
    assert float((y[0].cpu() - ref.gibbs(x[0], alpha)).abs().max()) < 2e-5
    assert float((y[1].cpu() - x[1]).abs().max()) < 2e-5
    # low resolution
    for zoom in (0.8, 1.17):
        small = tuple(int(round(n * zoom)) for n in (12, 20, 18))
        y = ops.resize_linear(ops.interp_nearest(xd[:1], small), (12, 20, 18))
        assert float((y[0].cpu() - ref.low_resolution(x[0], zoom)).abs().max()) < 1e-5


@pytest.mark.gpu
def test_augmenter_runs_the_whole_vocabulary(cuda):
    g = torch.Generator().manual_seed(1)
    batch = {"image": torch.rand((3, 1, 16, 24, 20), generator=g).to(cuda),
             "mask": (torch.rand((3, 1, 16, 24, 20), generator=g) > 0.8).float().to(cuda)}
    aug = DeviceAugmenter(["intensity", "noise", "rbf", "affine", "shear", "flip", "blur",
                           "distort", "lowres"], ["image", "mask"], ["image"], t2_keys=["image"],
                          seed=9)
    seen = set()
    for _ in range(12):
        out = aug(batch)
        assert out["image"].shape == batch["image"].shape and torch.isfinite(out["image"]).all()
        assert set(out["mask"].unique().tolist()) <= {0.0, 1.0}      # labels stay labels
        for it in aug.last_plan:
            seen.update(k for k in it if k != "flips")
    assert {"distort", "gamma", "gibbs_alpha", "rbf", "lowres", "affine"} <= seen
    tri = DeviceAugmenter(["trivial", "blur", "noise", "lowres"], ["image"], ["image"], seed=2)
    out = tri({"image": batch["image"]})
    assert torch.isfinite(out["image"]).all()

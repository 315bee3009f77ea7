"""Batch assembly: collate mirrors on the CPU, the device crop sampler
against plain slicing + torch.flip on the GPU."""
import pytest
import torch

from adell_mri_amd.utils.batching import (DeviceCropSampler, safe_collate, safe_collate_crops,
                                          unpack_crops)


def test_safe_collate_stacks_what_it_can():
    a = {"image": torch.zeros(1, 4, 4), "boxes": torch.zeros(2, 4), "name": "a"}
    b = {"image": torch.ones(1, 4, 4), "boxes": torch.zeros(3, 4), "name": "b"}
    out = safe_collate([a, b])
    assert tuple(out["image"].shape) == (2, 1, 4, 4)
    assert isinstance(out["boxes"], list) and len(out["boxes"]) == 2      # ragged: stays a list
    assert out["name"] == ["a", "b"]
    out = safe_collate([[torch.zeros(2), torch.zeros(3)], [torch.ones(2), torch.ones(3)]])
    assert tuple(out[0].shape) == (2, 2) and tuple(out[1].shape) == (2, 3)
    crops = [[{"image": torch.zeros(1, 2)}, {"image": torch.ones(1, 2)}], [{"image": torch.ones(1, 2)}]]
    assert len(unpack_crops(crops)) == 3
    assert tuple(safe_collate_crops(crops)["image"].shape) == (3, 1, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,crop,flip_axes", [((2, 20, 17, 13), (8, 8, 8), (0, 1, 2)),
                                                  ((1, 16, 16, 16), (16, 16, 16), (2,)),
                                                  ((3, 30, 22), (16, 8), (0, 1))])
def test_device_crops_equal_slicing_and_flip(cuda, shape, crop, flip_axes):
    g = torch.Generator().manual_seed(0)
    img = torch.randn(shape, generator=g).to(cuda)
    msk = (torch.rand((1, *shape[1:]), generator=g) > 0.5).float().to(cuda)
    sampler = DeviceCropSampler(crop, flip_axes=flip_axes, flip_prob=0.5, seed=3)
    ref = DeviceCropSampler(crop, flip_axes=flip_axes, flip_prob=0.5, seed=3)
    plan = ref.plan(list(shape[1:]), 6)
    out = sampler({"image": img, "mask": msk}, 6)
    assert tuple(out["image"].shape) == (6, shape[0], *crop)
    assert tuple(out["mask"].shape) == (6, 1, *crop)
    assert any(f for _, f in plan)                       # the case does flip something
    for i, (origin, flips) in enumerate(plan):
        index = (slice(None),) + tuple(slice(o, o + c) for o, c in zip(origin, crop))
        for key, src in (("image", img), ("mask", msk)):
            want = src[index]
            if flips:
                want = torch.flip(want, [a + 1 for a in flips])
            assert torch.equal(out[key][i], want), (key, i)

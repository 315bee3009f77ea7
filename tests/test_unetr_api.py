"""CPU-side conformance of the UNETR mirror: state_dict keys / shapes equal the
reference's (recorded in the golden fixture)."""
import os

import numpy as np
import torch

from adell_mri_amd.modules.activations import activation_factory
from adell_mri_amd.modules.segmentation.unetr import UNETR
from cases import UNETR_CASES

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build(kw):
    kw = dict(kw)
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    return UNETR(**kw)


import pytest  # noqa: E402


@pytest.mark.parametrize("name", ["unetr3d_small", "unetr2d_small"])
def test_unetr_state_dict_keys_and_shapes_equal_reference(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    net = build(UNETR_CASES[name])
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["param_keys"]]
    shapes = {str(k): tuple(int(i) for i in str(s).split(",")) for k, s in
              zip(g["param_keys"], g["param_shapes"])}
    for k, v in sd.items():
        assert tuple(v.shape) == shapes[k], k


def test_unetr_token_rearrangement_matches_einops():
    import einops

    net = build(UNETR_CASES["unetr3d_small"])
    emb = net.vit.embedding
    x = torch.randn(2, 1, 32, 32, 32)
    ref = einops.rearrange(x, "b c (h x) (w y) (d z) -> b (h w d) (x y z c)", x=8, y=8, z=8)
    assert torch.equal(emb._to_tokens(x), ref)
    t = torch.randn(2, 64, 512)
    ref = einops.rearrange(
        t, "b (h w d) (x s1 y s2 z s3 c) -> b (c s1 s2 s3) (h x) (w y) (d z)",
        h=4, w=4, d=4, x=2, y=2, z=2, s1=4, s2=4, s3=4, c=1)
    assert torch.equal(emb._from_tokens(t, [4, 4, 4]), ref)
    assert torch.equal(emb._from_tokens(emb._to_tokens(x), [1, 1, 1]), x)

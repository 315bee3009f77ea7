"""Whole-volume inference operators: the reference's own tests
(testing/test_segmentation_inference_pl.py:21-52: identity round trip, shapes, value range) on
the mirror, the mirror and the plain restatement (oracle/inference_ref.py) against outputs of the
REAL reference operators (tests/golden/inference_ops.npz), and the HIP U-Net through the operators
against the restatement."""
import numpy as np
import pytest
import torch

from adell_mri_amd.utils.inference import (FlippedInference, SegmentationInference,
                                           SlidingWindowSegmentation, TensorListReduction,
                                           window_plan)
from oracle import inference_ref

h, w, d, c = 32, 32, 32, 1


def test_sliding_window_inference_identity():
    net = torch.nn.Identity()
    x = torch.rand([1, c, h, w, d], generator=torch.Generator().manual_seed(0))
    sli = SlidingWindowSegmentation([8, 8, 8], lambda t: net.forward(t), n_classes=1)
    out = sli(x)
    assert list(out.shape) == [1, 1, h, w, d]
    assert out.max() <= 1 and out.min() >= 0
    assert torch.all(torch.isclose(out, x))


def test_segmentation_inference_identity_with_flip():
    net = torch.nn.Identity()
    x = torch.rand([1, c, h, w, d], generator=torch.Generator().manual_seed(1))
    keep = x.clone()
    sli = SegmentationInference(base_inference_function=lambda t: net.forward(t),
                                sliding_window_size=[8, 8, 8], n_classes=1, flip=True)
    out = sli(x)
    assert list(out.shape) == [1, 1, h, w, d]
    assert torch.all(torch.isclose(out, x))
    assert torch.equal(x, keep)   # the input is not modified


@pytest.mark.parametrize("shape,window,stride", [((20, 17, 13), (8, 8, 8), (5, 5, 5)),
                                                 ((16, 16, 16), (8, 8, 8), (8, 8, 8)),
                                                 ((9, 30, 11), (8, 16, 8), (3, 7, 8))])
def test_window_plan_matches_restatement(shape, window, stride):
    assert window_plan(shape, window, stride) == inference_ref.windows_3d(shape, window, stride)


@pytest.mark.parametrize("batch", [1, 3, 4])
def test_overlapping_windows_match_restatement(batch):
    """A position-dependent 'network' (2 output channels) over ragged overlapping windows, window
    batches of several sizes, dict input."""
    g = torch.Generator().manual_seed(2)
    x = torch.randn([2, 3, 20, 17, 13], generator=g)
    wgt = torch.randn([2, 3], generator=g)

    def net(t):   # 1x1x1 conv + a nonlinearity: not translation-trivial once windows overlap
        t = t["image"] if isinstance(t, dict) else t
        return torch.tanh(torch.einsum("oc,bcxyz->boxyz", wgt, t)) * t.sum(1, keepdim=True)

    sli = SlidingWindowSegmentation([8, 8, 8], net, n_classes=2, stride=[5, 6, 7],
                                    inference_batch_size=batch)
    out = sli({"image": x})
    ref = inference_ref.sliding_window_3d(
        x.numpy().astype(np.float64), lambda a: net(torch.from_numpy(a).float()).numpy(),
        (8, 8, 8), (5, 6, 7), 2)
    np.testing.assert_allclose(out.numpy(), ref, rtol=1e-5, atol=1e-6)


def test_flipped_inference_matches_restatement():
    g = torch.Generator().manual_seed(3)
    x = torch.randn([1, 2, 6, 7, 8], generator=g)
    ramp = torch.arange(8.0).view(1, 1, 1, 1, 8)

    def net(t):
        return t * ramp   # not flip-equivariant along the last axis

    out = FlippedInference(net, flips=[(2,), (4,)])(x)
    ref = inference_ref.flipped(x.numpy().astype(np.float64),
                                lambda a: net(torch.from_numpy(np.ascontiguousarray(a)).float()).numpy(),
                                [(2,), (4,)])
    np.testing.assert_allclose(out.numpy(), ref, rtol=1e-6, atol=1e-6)


def test_mc_dropout_and_reduction_shapes():
    drop = torch.nn.Dropout(0.5).train()
    x = torch.ones([1, c, 16, 16, 16])
    sli = SegmentationInference(base_inference_function=lambda t: torch.sigmoid(drop(t)),
                                sliding_window_size=[8, 8, 8], n_classes=1, flip=True,
                                mc_iterations=3)
    out = sli(x)
    assert list(out.shape) == [1, 2, 16, 16, 16]        # mean and standard deviation
    assert out.max() <= 1 and out.min() >= 0
    red = TensorListReduction(postproc_fn=lambda t: t * 2)
    two = SegmentationInference(base_inference_function=[lambda t: t, lambda t: 3 * t],
                                sliding_window_size=[8, 8, 8], n_classes=1, reduction=red)
    assert torch.allclose(two(x), 4 * x)


def test_unbatched_input_and_stride_fraction():
    x = torch.rand([c, 16, 16, 16], generator=torch.Generator().manual_seed(4))
    sli = SegmentationInference(base_inference_function=lambda t: t, sliding_window_size=[8, 8, 8],
                                stride=0.5, n_classes=1)
    assert sli.stride == [4, 4, 4]
    out = sli(x)
    assert list(out.shape) == [1, 16, 16, 16] and torch.allclose(out, x)


# ---- pinned to outputs of the REAL reference operators (oracle/make_golden_inference.py) -------------
def _golden():
    import os

    return np.load(os.path.join(os.path.dirname(__file__), "golden", "inference_ops.npz"))


def _build(kind, kw, net_factory):
    if kind == "sliding":
        return SlidingWindowSegmentation(kw["window"], net_factory(kw["n_classes"]),
                                         kw["n_classes"], kw["stride"], kw["batch"])
    if kind == "flip":
        return FlippedInference(net_factory(kw["n_out"]), flips=kw["flips"],
                                flip_keys=kw.get("flip_keys"))
    return SegmentationInference(base_inference_function=net_factory(1),
                                 sliding_window_size=kw["window"], stride=kw["stride"],
                                 inference_batch_size=kw["batch"], n_classes=kw["n_classes"],
                                 flip=kw["flip"])


def test_mirror_matches_reference_outputs():
    """SlidingWindowSegmentation (ragged overlapping windows, window batches 1 / 2 / 3 / 4, dict
    input with a non-tensor entry, 2 classes, default stride, 2-D), FlippedInference (tensor, dict +
    flip_keys) and SegmentationInference (fractional stride + flip) against what the reference's own
    classes returned for the same inputs and the same closed-form network."""
    from cases import INFERENCE_CASES, inference_net

    g = _golden()
    for name, (shape, kind, kw) in INFERENCE_CASES.items():
        x = torch.from_numpy(g[name + "/x"])
        assert tuple(x.shape) == shape
        X = {"image": x, "meta": "not a tensor"} if kw.get("as_dict") else x
        y = _build(kind, kw, inference_net)(X)
        want = torch.from_numpy(g[name + "/y"])
        assert y.shape == want.shape, name
        assert torch.allclose(y, want, rtol=1e-6, atol=1e-6), (name, float((y - want).abs().max()))


def test_restatement_matches_reference_outputs():
    """oracle/inference_ref.py (the checker of the GPU parity test below) against the same
    fixtures: its 3-D sliding window and its flip averaging."""
    from cases import INFERENCE_CASES, inference_net

    g = _golden()

    def as_np(net):
        return lambda a: net(torch.from_numpy(np.ascontiguousarray(a)).float()).numpy()

    for name in ("sw3d_ragged_b1", "sw3d_ragged_b3", "sw3d_default_stride"):
        kw = INFERENCE_CASES[name][2]
        stride = kw["stride"] if kw["stride"] is not None else kw["window"]
        ref = inference_ref.sliding_window_3d(g[name + "/x"].astype(np.float64),
                                              as_np(inference_net(kw["n_classes"])), kw["window"],
                                              stride, kw["n_classes"])
        np.testing.assert_allclose(ref, g[name + "/y"], rtol=1e-5, atol=1e-6)
    kw = INFERENCE_CASES["flip_tensor"][2]
    ref = inference_ref.flipped(g["flip_tensor/x"].astype(np.float64),
                                as_np(inference_net(kw["n_out"])), kw["flips"])
    np.testing.assert_allclose(ref, g["flip_tensor/y"], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_device_inputs_match_reference_outputs(cuda):
    """The same operators with the volume, the windows and the accumulators on the GPU."""
    from cases import INFERENCE_CASES, inference_net

    def net_on_device(n_out):
        net = inference_net(n_out)

        def fn(X):   # the closed-form network is host arithmetic; its inputs / outputs live on the GPU
            host = ({k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in X.items()}
                    if isinstance(X, dict) else X.cpu())
            return net(host).to(cuda)
        return fn

    g = _golden()
    for name, (shape, kind, kw) in INFERENCE_CASES.items():
        x = torch.from_numpy(g[name + "/x"]).to(cuda)
        X = {"image": x, "meta": "not a tensor"} if kw.get("as_dict") else x
        y = _build(kind, kw, net_on_device)(X)
        assert y.is_cuda
        want = torch.from_numpy(g[name + "/y"])
        assert torch.allclose(y.cpu(), want, rtol=1e-6, atol=1e-6), name


@pytest.mark.gpu
def test_sliding_window_unet_matches_oracle_composition(cuda):
    """The HIP U-Net over overlapping windows of a ragged volume == the torch-CPU oracle U-Net
    composed by the plain restatement."""
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.unet import UNet
    from oracle.torch_ref.unet import UNetOracle
    from oracle.weights import tensor_for
    kw = dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
              upscale_type="transpose", norm_type="instance", padding=1, dropout_param=0.0,
              in_channels=2, n_classes=2, depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3)
    net = UNet(activation_fn=activation_factory["swish"], **kw)
    sd = {k: torch.from_numpy(tensor_for(k, v.shape)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net = net.to(cuda).eval()
    ref = UNetOracle(sd, dict(depth=kw["depth"], kernel_sizes=kw["kernel_sizes"],
                              strides=kw["strides"], padding=1, norm_type="instance",
                              activation="swish", link_type="residual", n_classes=2,
                              dropout_param=0.0))
    x = torch.rand([1, 2, 40, 24, 36], generator=torch.Generator().manual_seed(5))
    sli = SegmentationInference(base_inference_function=lambda t: net(t)[0],
                                sliding_window_size=[16, 16, 16], stride=[12, 8, 10], n_classes=2,
                                flip=True, inference_batch_size=4)
    out = sli(x.to(cuda)).cpu().numpy()

    def ref_fn(a):
        with torch.no_grad():
            return ref.forward(torch.from_numpy(np.ascontiguousarray(a)).float(),
                               return_logits=False).numpy()

    want = inference_ref.flipped(
        x.numpy(), lambda a: inference_ref.sliding_window_3d(a, ref_fn, (16,) * 3, (12, 8, 10), 1),
        [(2,)])
    assert out.shape == want.shape
    np.testing.assert_allclose(out, want, rtol=1e-4, atol=1e-5)

"""Generate tests/golden/*.npz from the REAL reference (run in the build
container only; /root/reference never travels to the GPU box).

Import recipe: the reference's package __init__ files pull
in monai/lightning, which are not installed, so empty package stubs with the
right __path__ are registered first and the leaf modules are imported directly.

    python oracle/make_golden.py
"""
import os
import sys
import types

import numpy as np

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REF = os.environ.get("ADELL_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for name, path in [
    ("adell_mri", "adell_mri"), ("adell_mri.modules", "adell_mri/modules"),
    ("adell_mri.modules.layers", "adell_mri/modules/layers"),
    ("adell_mri.modules.segmentation", "adell_mri/modules/segmentation"),
    ("adell_mri.utils", "adell_mri/utils"),
    ("adell_mri.modules.self_supervised", "adell_mri/modules/self_supervised"),
    ("adell_mri.modules.self_supervised.losses", "adell_mri/modules/self_supervised/losses"),
    ("adell_mri.modules.semi_supervised_segmentation",
     "adell_mri/modules/semi_supervised_segmentation"),
]:
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(REF, path)]
    sys.modules[name] = m
import einops.layers.torch  # noqa: E402,F401  (the reference uses it via bare `import einops`)
import torch  # noqa: E402

from adell_mri.modules.activations import activation_factory  # noqa: E402
from adell_mri.modules.layers.adn_fn import get_adn_fn  # noqa: E402
from adell_mri.modules.layers.res_blocks import ResidualBlock3d  # noqa: E402
from adell_mri.modules.segmentation.losses import (  # noqa: E402
    binary_focal_loss, binary_generalized_dice_loss)
from adell_mri.modules.segmentation.unet import BrUNet, UNet  # noqa: E402
from adell_mri.modules.segmentation.unetr import SWINUNet, UNETR  # noqa: E402
from adell_mri.modules.segmentation.unetpp import UNetPlusPlus  # noqa: E402
from adell_mri.modules.layers.linear_blocks import MultiHeadSelfAttention  # noqa: E402
from adell_mri.modules.layers.vit import TransformerBlock  # noqa: E402
from adell_mri.modules.layers.res_net import ResNet  # noqa: E402

from adell_mri.modules.layers.conv_next import ConvNeXt  # noqa: E402
from adell_mri.modules.layers.res_blocks import ConvNeXtBlock3d  # noqa: E402
from adell_mri.modules.self_supervised.losses.vicreg import VICRegLoss  # noqa: E402

from oracle.weights import fill_state_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

UNET_CASES = {
    # name: (constructor kwargs, input shape, input distribution)
    "unet3d_cfg2_tiny": (dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
                              upscale_type="transpose", norm_type="instance", padding=1,
                              dropout_param=0.15, activation_fn="swish", in_channels=2,
                              n_classes=2, depth=[4, 4, 8, 16, 32], kernel_sizes=[3] * 5,
                              strides=[2] * 5), (1, 2, 32, 32, 32), "uniform"),
    "unet3d_cfg2_small": (dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
                               upscale_type="transpose", norm_type="instance", padding=1,
                               dropout_param=0.15, activation_fn="swish", in_channels=2,
                               n_classes=2, depth=[8, 8, 16], kernel_sizes=[3] * 3,
                               strides=[2] * 3), (2, 2, 16, 16, 16), "normal"),
    "unet3d_identity_links_relu": (dict(spatial_dimensions=3, conv_type="regular",
                                        link_type="identity", upscale_type="transpose",
                                        norm_type="instance", padding=1, dropout_param=0.0,
                                        activation_fn="relu", in_channels=1, n_classes=2,
                                        depth=[8, 16, 32], kernel_sizes=[3] * 3,
                                        strides=[2] * 3), (1, 1, 16, 16, 16), "uniform"),
    "unet3d_conv_links_gelu": (dict(spatial_dimensions=3, conv_type="regular", link_type="conv",
                                    upscale_type="transpose", norm_type="instance", padding=1,
                                    dropout_param=0.0, activation_fn="gelu", in_channels=3,
                                    n_classes=2, depth=[8, 16, 32], kernel_sizes=[3] * 3,
                                    strides=[2] * 3), (1, 3, 16, 16, 16), "normal"),
}


ATTENTION_LINK_CASES = {
    # link_type="attention" (unet.py:473-481: SelfAttentionBlock over [16, 16, 1] patches of every
    # skip tensor; the reference tests it for shapes only, testing/test_unet.py:204-235)
    "unet3d_attention_links": (dict(spatial_dimensions=3, conv_type="regular",
                                    link_type="attention", upscale_type="transpose",
                                    norm_type="instance", padding="same", dropout_param=0.0,
                                    activation_fn="swish", in_channels=1, n_classes=2,
                                    depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3,
                                    _grad64=True),
                               (2, 1, 64, 64, 8), "uniform"),
    "unet2d_attention_links": (dict(spatial_dimensions=2, conv_type="regular",
                                    link_type="attention", upscale_type="transpose",
                                    norm_type="instance", padding="same", dropout_param=0.0,
                                    activation_fn="gelu", in_channels=2, n_classes=2,
                                    depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3,
                                    _grad64=True),
                               (2, 2, 64, 96), "uniform"),
}


SAE_CASES = {
    # conv_type="sae" (unet.py:375-397): conv block + ConcurrentSqueezeAndExcite in every level
    "unet3d_sae": (dict(spatial_dimensions=3, conv_type="sae", link_type="identity",
                        upscale_type="transpose", norm_type="instance", padding=1,
                        dropout_param=0.0, activation_fn="swish", in_channels=2, n_classes=2,
                        depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
                   (2, 2, 16, 16, 16), "uniform"),
    "unet2d_sae": (dict(spatial_dimensions=2, conv_type="sae", link_type="conv",
                        upscale_type="transpose", norm_type="instance", padding=1,
                        dropout_param=0.0, activation_fn="relu", in_channels=1, n_classes=2,
                        depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
                   (2, 1, 32, 32), "normal"),
}
ASP_CASES = {
    # conv_type="asp" (unet.py:399-413, multi_resolution.py:291-416): every encoder op is an atrous
    # spatial pyramid (rates 1 and 2: dilated conv -> ADN -> depthwise-separable conv -> ADN per
    # rate, outputs concatenated) that IGNORES kernel size, stride and padding -- the encoder never
    # downsamples and the decoder crops its upsampled tensors to the skips; decoder ops are "sae"
    "unet3d_asp": (dict(spatial_dimensions=3, conv_type="asp", link_type="identity",
                        upscale_type="transpose", norm_type="instance", padding=1,
                        dropout_param=0.0, activation_fn="swish", in_channels=2, n_classes=2,
                        depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
                   (2, 2, 12, 10, 9), "uniform"),
    # (no 2-D case: AtrousSpatialPyramidPooling2d cannot be constructed -- standard_blocks.py:78
    # reads self.paddign)
}
DEPTHWISE_CASES = {
    # conv_type="depthwise" (unet.py:292-307): Conv(groups = channels, k, stride) -> ADN -> 1x1 conv.
    # The constructor default padding="same" (the 1x1 conv takes the SAME padding argument, so an
    # integer padding > 0 would grow every tensor by 2 p per block in the reference)
    "unet3d_depthwise": (dict(spatial_dimensions=3, conv_type="depthwise", link_type="identity",
                              upscale_type="transpose", norm_type="instance", padding="same",
                              dropout_param=0.0, activation_fn="swish", in_channels=2,
                              n_classes=2, depth=[8, 16, 32], kernel_sizes=[3] * 3,
                              strides=[2] * 3), (2, 2, 16, 16, 16), "normal"),
    # 2-D: the same block on Conv2d (depthwise_conv_block_2d, unet.py:276-290)
    "unet2d_depthwise": (dict(spatial_dimensions=2, conv_type="depthwise", link_type="identity",
                              upscale_type="transpose", norm_type="instance", padding="same",
                              dropout_param=0.0, activation_fn="relu", in_channels=1, n_classes=2,
                              depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
                         (2, 1, 32, 48), "uniform"),
}


UNET2D_CASES = {
    # BASELINE configs[0]: the 2-D U-Net of the reference's own testing/test_unet.py:63-72 with
    # the constructor defaults (BatchNorm2d, PReLU), 140 748 parameters. train() for batch
    # statistics; dropout_param (default 0.1) is 0 because torch's dropout stream cannot be
    # reproduced
    "unet2d_cfg1": (dict(spatial_dimensions=2, depth=[16, 32, 64], upscale_type="transpose",
                         padding="same", strides=[2, 2, 2], kernel_sizes=[3, 3, 3],
                         conv_type="regular", link_type="identity", activation_fn="prelu",
                         dropout_param=0.0, _train=True), (2, 1, 64, 64), "uniform"),
    # the constructor defaults of the reference (unet.py:43-68): upscale_type="upsample" with
    # bilinear interpolation, identity links, BatchNorm2d, PReLU
    "unet2d_upsample": (dict(spatial_dimensions=2, depth=[8, 16, 32], padding="same",
                             strides=[2, 2, 2], kernel_sizes=[3, 3, 3], activation_fn="prelu",
                             dropout_param=0.0, _train=True), (2, 1, 40, 48), "uniform"),
    # 2-D residual skip links (ResidualBlock2d, res_blocks.py:13-105) + skip conditioning input
    "unet2d_residual_links": (dict(spatial_dimensions=2, depth=[8, 16, 32], padding=1,
                                   strides=[2, 2, 2], kernel_sizes=[3, 3, 3],
                                   upscale_type="transpose", norm_type="instance",
                                   activation_fn="swish", dropout_param=0.0,
                                   link_type="residual", in_channels=2), (2, 2, 32, 40), "uniform"),
    # conv_type="resnet" in 2-D: ResidualBlock2d encoder blocks followed by MaxPool2d
    # (res_block_conv_2d, unet.py:309-342)
    "unet2d_resnet_blocks": (dict(spatial_dimensions=2, depth=[8, 16, 32], padding=1,
                                  strides=[2, 2, 2], kernel_sizes=[3, 3, 3], conv_type="resnet",
                                  upscale_type="transpose", norm_type="instance",
                                  activation_fn="swish", dropout_param=0.0,
                                  link_type="identity", in_channels=1), (2, 1, 32, 48), "uniform"),
    # skip conditioning (X_skip_layer resized to every skip resolution and concatenated,
    # unet.py:796-801) + deep supervision heads (unet.py:657-683, 836-841)
    "unet3d_skipcond_deepsup": (dict(spatial_dimensions=3, depth=[8, 16, 32], padding=1,
                                     strides=[2, 2, 2], kernel_sizes=[3, 3, 3],
                                     upscale_type="transpose", norm_type="instance",
                                     activation_fn="swish", dropout_param=0.0,
                                     link_type="conv", in_channels=1, skip_conditioning=1,
                                     deep_supervision=True), (1, 1, 24, 24, 24), "uniform"),
    # tabular feature conditioning (unet.py:716-740, 803-810): Linear -> BatchNorm1d -> swish ->
    # Linear -> BatchNorm1d -> sigmoid gates on every skip connection; train() for the batch
    # statistics of the BatchNorm1d layers over 4 items
    "unet3d_feature_cond": (dict(spatial_dimensions=3, depth=[8, 16, 32], padding=1,
                                 strides=[2, 2, 2], kernel_sizes=[3, 3, 3],
                                 upscale_type="transpose", norm_type="instance",
                                 activation_fn="swish", dropout_param=0.0, link_type="identity",
                                 in_channels=1, feature_conditioning=5, _train=True),
                            (4, 1, 16, 16, 16), "uniform"),
    # 3-D: 1x1x1 conv + trilinear Upsample, anisotropic stride at the deepest level
    "unet3d_upsample": (dict(spatial_dimensions=3, depth=[8, 16, 32], padding=1,
                             strides=[2, 2, [2, 2, 1]], kernel_sizes=[3, 3, 3],
                             upscale_type="upsample", interpolation="trilinear",
                             norm_type="instance", activation_fn="swish", dropout_param=0.0,
                             link_type="identity", in_channels=2), (1, 2, 16, 24, 12), "uniform"),
}


UNETR_CASES = {
    "unetr3d_small": (dict(image_size=[32, 32, 32], patch_size=[8, 8, 8], number_of_blocks=4,
                           return_at=[1, 2], embedding_size=64, attention_dim=64, hidden_dim=64,
                           n_heads=4, mlp_structure=[128], spatial_dimensions=3,
                           link_type="identity", upscale_type="transpose", norm_type="instance",
                           padding=1, dropout_param=0.0, activation_fn="swish", in_channels=1,
                           n_classes=2, depth=[8, 16, 32], kernel_sizes=[3, 3, 3]),
                      (2, 1, 32, 32, 32), "uniform"),
}


UNETR_CASES["unetr3d_feature_cond"] = (
    # tabular feature gates on the UNETR skip tensors (unetr.py:217-218, 356-358, 399-405);
    # eval(): the MLP blocks of the ViT carry the default dropout of get_adn_fn (0.1), whose
    # random stream cannot be reproduced, so the BatchNorm1d gates use their running statistics
    dict(image_size=[16, 16, 16], patch_size=[4, 4, 4], number_of_blocks=2, return_at=[1],
         embedding_size=32, attention_dim=32, hidden_dim=32, n_heads=2, mlp_structure=[64],
         spatial_dimensions=3, link_type="identity", upscale_type="transpose",
         norm_type="instance", padding=1, dropout_param=0.0, activation_fn="swish", in_channels=1,
         n_classes=2, depth=[8, 16], kernel_sizes=[3, 3], feature_conditioning=5),
    (4, 1, 16, 16, 16), "uniform")
UNETR_CASES["unetr2d_small"] = (
    # 2-D UNETR (unetr.py:104 of the HIP mirror raised for this before round 2)
    dict(image_size=[32, 48], patch_size=[8, 8], number_of_blocks=4, return_at=[1, 2],
         embedding_size=64, attention_dim=64, hidden_dim=64, n_heads=4, mlp_structure=[128],
         spatial_dimensions=2, link_type="identity", upscale_type="transpose",
         norm_type="instance", padding=1, dropout_param=0.0, activation_fn="swish", in_channels=2,
         n_classes=2, depth=[8, 16, 32], kernel_sizes=[3, 3, 3]), (2, 2, 32, 48), "uniform")


UNETPP_CASES = {
    "unetpp3d_small": (dict(spatial_dimensions=3, conv_type="regular", upscale_type="transpose",
                            norm_type="instance", padding=1, dropout_param=0.0,
                            activation_fn="swish", in_channels=2, n_classes=2,
                            depth=[8, 8, 16, 32], kernel_sizes=[3] * 4, strides=[2] * 4,
                            _cls="unetpp"), (1, 2, 16, 16, 16), "uniform"),
    # 2-D U-Net++ (DenseBlock 2-D, standard_blocks.py:284-376)
    "unetpp2d_small": (dict(spatial_dimensions=2, conv_type="regular", upscale_type="transpose",
                            norm_type="instance", padding=1, dropout_param=0.0,
                            activation_fn="swish", in_channels=1, n_classes=2,
                            depth=[8, 8, 16], kernel_sizes=[3] * 3, strides=[2] * 3,
                            _cls="unetpp"), (2, 1, 24, 40), "uniform"),
}


SWIN_CASES = {
    # BASELINE config 5 (unet-swin.yaml) in miniature: convolutional embedding, 8^3 windows of
    # 4^3 patches (8 tokens), shifts [0, 1], anisotropic first stride, conv links
    "swinunet3d_small": (dict(image_size=[32, 32, 16], patch_size=[4, 4, 4],
                              window_size=[8, 8, 8], shift_sizes=[[0, 1], [0, 1], [0, 1]],
                              embedding_size=[16, 32, 64], n_heads=4, dropout_rate=0.0,
                              embed_method="convolutional", mlp_structure=4.0,
                              spatial_dimensions=3, conv_type="regular", link_type="conv",
                              upscale_type="transpose", norm_type="instance", padding="same",
                              dropout_param=0.0, activation_fn="leaky_relu", in_channels=2,
                              n_classes=2, depth=[8, 16, 32], kernel_sizes=[3, 3, 3],
                              strides=[[2, 2, 1], 2, 2], _cls="swin"),
                         (2, 2, 32, 32, 16), "uniform"),
    # the 2-D SWIN-UNet the reference's own tests build (testing/test_swin_unet.py:15, 43-161):
    # linear embedding, 8 x 8 windows of 4 x 4 patches, anisotropic first stride
    "swinunet2d_small": (dict(image_size=[32, 64], patch_size=[4, 4], window_size=[8, 8],
                              shift_sizes=[[0, 1], [0, 1], [0, 1]], embedding_size=[16, 32, 64],
                              n_heads=4, dropout_rate=0.0, embed_method="linear",
                              mlp_structure=4.0, spatial_dimensions=2, conv_type="regular",
                              link_type="conv", upscale_type="transpose", norm_type="instance",
                              padding="same", dropout_param=0.0, activation_fn="leaky_relu",
                              in_channels=2, n_classes=2, depth=[8, 16, 32],
                              kernel_sizes=[3, 3, 3], strides=[[2, 1], 2, 2], _cls="swin"),
                         (2, 2, 32, 64), "uniform"),
}


BACKBONE_CASES = {
    # BASELINE config 2b in miniature: ResNet backbone (7^3 stem, k=5 / k=3 residual stages,
    # batch norm) repackaged as U-Net encoder, anisotropic pooling [2,2,1]
    "unet3d_resnet_backbone": (dict(spatial_dimensions=3, upscale_type="transpose",
                                    link_type="identity", norm_type="instance", padding=1,
                                    dropout_param=0.0, activation_fn="swish", in_channels=2,
                                    n_classes=2, _cls="backbone",
                                    _structure=[[8, 8, 5, 2], [16, 16, 3, 2]],
                                    _maxpool=[[2, 2, 1], [2, 2, 2]]),
                               (2, 2, 16, 16, 8), "uniform"),
}


def make_backbone_unet(kw):
    """entrypoints/segmentation/train.py:672-765 in miniature."""
    kw = dict(kw)
    kw.pop("_cls")
    structure, mpl = kw.pop("_structure"), kw.pop("_maxpool")
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    res = ResNet(dict(spatial_dim=3, in_channels=kw["in_channels"], structure=structure,
                      maxpool_structure=mpl, res_type="resnet",
                      adn_fn=get_adn_fn(3, "batch", "swish", 0.0)))
    bb = res.backbone
    res_ops = [bb.input_layer, *bb.operations]
    pool_ops = [bb.first_pooling, *bb.pooling_operations]
    enc = torch.nn.ModuleList([torch.nn.ModuleList([a, b]) for a, b in zip(res_ops, pool_ops)])
    kw["depth"] = [structure[0][0], *[x[0] for x in structure]]
    kw["kernel_sizes"] = [3 for _ in kw["depth"]]
    kw["strides"] = [2, *mpl]
    net = UNet(encoding_operations=enc, **kw)
    net.load_state_dict(fill_state_dict(net.state_dict()))
    return net


def make_unet(kw):
    if kw.get("_cls") == "backbone":
        return make_backbone_unet(kw)
    kw = dict(kw)
    kw.pop("_train", None)
    kw.pop("_grad64", None)
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    cls = {"unetpp": UNetPlusPlus, "swin": SWINUNet}.get(
        kw.pop("_cls", None), UNETR if "patch_size" in kw else UNet)
    net = cls(**kw)
    net.load_state_dict(fill_state_dict(net.state_dict()))
    return net


def gen_unet(name, kw, shape, dist):
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(shape, generator=g) if dist == "uniform" else torch.randn(shape, generator=g)
    y = (torch.rand((shape[0], 1, *shape[2:]), generator=g) > 0.9).float()
    net = make_unet(kw)
    # eval(): dropout off, instance norm unaffected; the backbone case runs in train() so that
    # its BatchNorm layers use batch statistics (dropout_param is 0 there)
    net = net.train() if (kw.get("_cls") == "backbone" or kw.get("_train")) else net.eval()
    out = {"x": x.numpy(), "y": y.numpy()}
    # forward parity target: logits (north_star: within 1e-4 rel)
    extra = {}
    if kw.get("feature_conditioning"):   # tabular features gating the skip connections
        fc = torch.randn((shape[0], kw["feature_conditioning"]), generator=g)
        out["x_fc"] = fc.numpy()
        extra["X_feature_conditioning"] = fc
    if kw.get("skip_conditioning"):      # extra image channels concatenated to every skip tensor
        sk = torch.rand((shape[0], kw["skip_conditioning"], *shape[2:]), generator=g)
        out["x_skip"] = sk.numpy()
        extra["X_skip_layer"] = sk
    logits = net(x, return_logits=True, **extra)[0]
    out["logits"] = logits.detach().numpy()
    res = net(x, **extra)
    prob = res[0]
    if len(res) == 3 and isinstance(res[2], list):  # U-Net++ auxiliary heads
        for i, a in enumerate(res[2]):
            out[f"aux{i}"] = a.detach().numpy()
    out["prob"] = prob.detach().numpy()
    # training-step arithmetic (segmentation/pl.py:218-222,284-317): dice + focal
    d = binary_generalized_dice_loss(prob, y, smooth=1e-5, eps=1e-6)
    f = binary_focal_loss(prob, y, gamma=1.0, eps=1e-6)
    loss = torch.stack([d.mean(), f.mean()]).mean()
    out["dice"] = d.detach().numpy()
    out["focal"] = f.detach().numpy()
    out["loss"] = loss.detach().numpy()
    net.zero_grad()
    loss.backward()
    sd_keys = [k for k, _ in net.named_parameters()]
    for k, p in net.named_parameters():
        if p.grad is not None:  # parameters the forward never touches have no gradient
            out["grad:" + k] = p.grad.numpy().copy()
    if kw.get("_cls") == "swin" or kw.get("_grad64"):
        # the per-voxel LayerNorm over 2 channels is ill-conditioned: fp32 gradients of the
        # reference itself carry up to ~1e-2 relative noise, so the fp64 gradients of the same
        # network are stored as the parity target (the test scales its tolerance by the
        # reference's own fp32-vs-fp64 difference)
        net64 = make_unet(kw).eval().double()
        for m in net64.modules():
            if getattr(m, "attention_mask", None) is not None:
                m.attention_mask = m.attention_mask.double()
        prob64 = net64(x.double())[0]
        d64 = binary_generalized_dice_loss(prob64, y.double(), smooth=1e-5, eps=1e-6)
        f64 = binary_focal_loss(prob64, y.double(), gamma=1.0, eps=1e-6)
        torch.stack([d64.mean(), f64.mean()]).mean().backward()
        for k, p in net64.named_parameters():
            if p.grad is not None:
                out["grad64:" + k] = p.grad.numpy().copy()
    out["param_shapes"] = np.array([",".join(map(str, p.shape)) for _, p in net.named_parameters()])
    # one and two SGD-Nesterov steps as configured at segmentation/pl.py:563-569
    opt = torch.optim.SGD(net.parameters(), lr=5e-4, momentum=0.99, weight_decay=5e-3,
                          nesterov=True)
    opt.step()
    if "patch_size" not in kw and "_cls" not in kw:
        for k, p in net.named_parameters():
            out["step1:" + k] = p.detach().numpy().copy()
    out["param_keys"] = np.array(sd_keys)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "params", sum(p.numel() for p in net.parameters()), "loss", float(loss))


BRUNET_CASES = {
    # multi-branch U-Net (unet.py:846-1253): two encoders, concurrent squeeze-and-excite merges.
    # (constructor kwargs, per-branch input shape, missing[(branch, item), ...])
    "brunet3d_two_branch": (dict(spatial_dimensions=3, n_input_branches=2, depth=[8, 16, 32],
                                 upscale_type="transpose", padding=1, strides=[2, 2, 2],
                                 kernel_sizes=[3, 3, 3], conv_type="regular",
                                 link_type="identity", norm_type="instance",
                                 activation_fn="swish", dropout_param=0.0, in_channels=1),
                            (2, 1, 16, 24, 16), []),
    # 2-D, conv links, inputs missing for some items (fix_input zero-fills them and the branch
    # weights drop them from the merge, unet.py:1094-1111, 1160-1207)
    "brunet2d_missing_inputs": (dict(spatial_dimensions=2, n_input_branches=2, depth=[8, 16, 32],
                                     upscale_type="transpose", padding=1, strides=[2, 2, 2],
                                     kernel_sizes=[3, 3, 3], conv_type="regular",
                                     link_type="conv", norm_type="instance",
                                     activation_fn="swish", dropout_param=0.0, in_channels=2),
                                (3, 2, 32, 40), [(0, 2), (1, 1)]),
}


def gen_brunet(name, kw, shape, missing):
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(4321)
    nb = kw["n_input_branches"]
    items = [[torch.rand(shape[1:], generator=g) for _ in range(shape[0])] for _ in range(nb)]
    for b, i in missing:
        items[b][i] = None
    y = (torch.rand((shape[0], 1, *shape[2:]), generator=g) > 0.9).float()
    kw = dict(kw)
    kw["activation_fn"] = activation_factory[kw["activation_fn"]]
    net = BrUNet(**kw)
    net.load_state_dict(fill_state_dict(net.state_dict()))
    net = net.eval()
    if missing:
        xs, ws = BrUNet.fix_input(items)
    else:
        xs, ws = [torch.stack(it, 0) for it in items], None
    out = {"y": y.numpy()}
    for b in range(nb):
        out[f"x{b}"] = xs[b].numpy()
        if ws is not None:
            out[f"w{b}"] = ws[b].numpy()
    out["logits"] = net(xs, ws, return_logits=True)[0].detach().numpy()
    out["bottleneck"] = net(xs, ws, return_bottleneck=True)[2].detach().numpy()
    prob = net(xs, ws)[0]
    out["prob"] = prob.detach().numpy()
    d = binary_generalized_dice_loss(prob, y, smooth=1e-5, eps=1e-6)
    f = binary_focal_loss(prob, y, gamma=1.0, eps=1e-6)
    loss = torch.stack([d.mean(), f.mean()]).mean()
    out["loss"] = loss.detach().numpy()
    net.zero_grad()
    loss.backward()
    for k, p in net.named_parameters():
        if p.grad is not None:
            out["grad:" + k] = p.grad.numpy().copy()
    out["param_keys"] = np.array([k for k, _ in net.named_parameters()])
    out["param_shapes"] = np.array([",".join(map(str, p.shape)) for _, p in net.named_parameters()])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "params", sum(p.numel() for p in net.parameters()), "loss", float(loss))


def gen_blocks():
    g = torch.Generator().manual_seed(7)
    out = {}
    x = torch.randn((2, 8, 8, 8, 8), generator=g)
    adn = get_adn_fn(3, "instance", "swish", 0.0)
    blk = ResidualBlock3d(8, 3, out_channels=8, adn_fn=adn).eval()
    blk.load_state_dict(fill_state_dict(blk.state_dict()))
    out["res_x"] = x.numpy()
    out["res_y"] = blk(x).detach().numpy()
    blk2 = ResidualBlock3d(8, 3, inter_channels=4, out_channels=6, adn_fn=adn).eval()
    blk2.load_state_dict(fill_state_dict(blk2.state_dict()))
    out["res2_y"] = blk2(x).detach().numpy()
    for act in ["swish", "relu", "gelu", "leaky_relu", "sigmoid", "tanh", "elu"]:
        m = get_adn_fn(3, "instance", act, 0.0)(8).eval()
        out["adn_" + act] = m(x).detach().numpy()
    p = torch.rand((2, 1, 8, 8, 8), generator=g)
    t = (torch.rand((2, 1, 8, 8, 8), generator=g) > 0.7).float()
    out["loss_p"], out["loss_t"] = p.numpy(), t.numpy()
    out["loss_dice"] = binary_generalized_dice_loss(p, t, smooth=1e-5, eps=1e-6).numpy()
    out["loss_focal"] = binary_focal_loss(p, t, gamma=1.0, eps=1e-6).numpy()
    out["loss_focal_g2"] = binary_focal_loss(p, t, gamma=2.0, eps=1e-6).numpy()
    # token blocks (linear_blocks.py:248-417, vit.py:884-1002)
    xt = torch.randn((2, 24, 32), generator=g)
    mha = MultiHeadSelfAttention(32, 32, 48, 32, n_heads=4).eval()
    mha.load_state_dict(fill_state_dict(mha.state_dict()))
    out["tok_x"] = xt.numpy()
    out["mha_y"] = mha(xt).detach().numpy()
    tb = TransformerBlock(32, 32, 32, n_heads=4, mlp_structure=[64]).eval()
    tb.load_state_dict(fill_state_dict(tb.state_dict()))
    out["tb_y"] = tb(xt).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **out)
    print("blocks ok")


VIT_TOKEN_KW = dict(image_size=[16, 16, 16], patch_size=[8, 8, 8], in_channels=2, number_of_blocks=2,
                    attention_dim=32, hidden_dim=32, embedding_size=32, n_heads=4, dropout_rate=0.0,
                    mlp_structure=[64])


def gen_vit_tokens():
    """ViT with the LinearEmbedding options of vit.py:389-881 that UNETR does not use: class token,
    registers, the fixed sinusoidal table (learnable_embedding False) and patch erasing
    (vit.py:1731-1736, 1793-1794): outputs, every parameter gradient, the initial table, and a
    seeded training-mode forward with erased patches (all other dropouts off, so the erasing mask
    is the only consumer of the host generator)."""
    from adell_mri.modules.layers.vit import ViT

    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 2, 16, 16, 16), generator=g)
    adn = get_adn_fn(1, "identity", "gelu", 0.0)
    out = {"x": x.numpy()}
    net = ViT(**VIT_TOKEN_KW, adn_fn=adn, use_class_token=True, n_registers=2,
              learnable_embedding=False).eval()
    out["pos_init"] = net.embedding.positional_embedding.detach().numpy().copy()
    net.load_state_dict(fill_state_dict(net.state_dict()))
    out["state_keys"] = np.array(list(net.state_dict().keys()))
    y, hidden = net(x, return_at=[0])
    wgt = torch.from_numpy(np.asarray(
        np.random.default_rng(3).uniform(-1, 1, size=tuple(y.shape)), dtype=np.float32))
    (y * wgt).sum().backward()
    out["y"], out["hidden0"], out["wgt"] = y.detach().numpy(), hidden[0].detach().numpy(), wgt.numpy()
    keys = []
    for k, p_ in net.named_parameters():
        if p_.grad is not None:
            out["grad:" + k] = p_.grad.numpy()
            keys.append(k)
    out["grad_keys"] = np.array(keys)
    # patch erasing, learnable table, class token only
    net2 = ViT(**VIT_TOKEN_KW, adn_fn=adn, use_class_token=True, patch_erasing=0.4).train()
    net2.load_state_dict(fill_state_dict(net2.state_dict()))
    torch.manual_seed(5)
    y2, _ = net2(x)
    out["y_erased"] = y2.detach().numpy()
    torch.manual_seed(5)
    out["erase_mask"] = (torch.rand([2, 9]) > 0.4).numpy()
    # channel tokens (vit.py:484-487, 566-571, 622-645): every channel of a patch is a token
    net3 = ViT(**VIT_TOKEN_KW, adn_fn=adn, channel_to_token=True).eval()
    net3.load_state_dict(fill_state_dict(net3.state_dict()))
    out["c2t_state_keys"] = np.array(list(net3.state_dict().keys()))
    y3, _ = net3(x)
    w3 = torch.from_numpy(np.asarray(
        np.random.default_rng(4).uniform(-1, 1, size=tuple(y3.shape)), dtype=np.float32))
    net3.zero_grad()
    (y3 * w3).sum().backward()
    out["c2t_y"], out["c2t_w"] = y3.detach().numpy(), w3.numpy()
    keys3 = []
    for k, p_ in net3.named_parameters():
        if p_.grad is not None:
            out["c2t_grad:" + k] = p_.grad.numpy()
            keys3.append(k)
    out["c2t_grad_keys"] = np.array(keys3)
    np.savez_compressed(os.path.join(OUT, "vit_tokens.npz"), **out)
    print("vit_tokens ok", y.shape, float(y.abs().mean()), out["erase_mask"].mean(), "channel tokens",
          tuple(y3.shape))


SSL_CASE = dict(
    backbone_args=dict(spatial_dim=3, in_channels=1, structure=[[8, 16, 3, 2], [16, 32, 7, 2]],
                       maxpool_structure=[2, 2]),
    projection_head_args=dict(in_channels=16, structure=[32, 24]),
    prediction_head_args=dict(in_channels=24, structure=[32, 24]))
SSL_GAIN = 3.0
SSL_OPT = dict(lr=1e-3, weight_decay=5e-3, eps=1e-8)


def gen_ssl():
    """ConvNeXt blocks, VICReg loss and one VICReg training step of a small 3-D ConvNeXt
    (BASELINE config 4 in miniature; self_supervised/pl.py:904-972 with
    ssl_method='vicreg', stop_gradient=False), AdamW step and EMA update."""
    g = torch.Generator().manual_seed(11)
    out = {}
    x = torch.randn((2, 8, 6, 6, 6), generator=g)
    out["blk_x"] = x.numpy()
    for tag, (k, oc) in {"k3": (3, 8), "k7": (7, 12)}.items():
        blk = ConvNeXtBlock3d(8, k, 16, oc)
        blk.load_state_dict(fill_state_dict(blk.state_dict()))
        xin = x.clone().requires_grad_(True)
        y = blk(xin)
        r = torch.randn(y.shape, generator=g)
        (y * r).sum().backward()
        out[f"blk_{tag}_y"], out[f"blk_{tag}_r"] = y.detach().numpy(), r.numpy()
        out[f"blk_{tag}_dx"] = xin.grad.numpy()
        for n, p in blk.named_parameters():
            out[f"blk_{tag}_grad:{n}"] = p.grad.numpy().copy()
    # VICReg loss on its own (losses/vicreg.py:30-165)
    e1 = torch.randn((6, 40), generator=g).requires_grad_(True)
    e2 = (0.5 * e1.detach() + 0.7 * torch.randn((6, 40), generator=g)).requires_grad_(True)
    crit = VICRegLoss()
    terms = crit(e1, e2)
    sum(terms).backward()
    out["vic_x1"], out["vic_x2"] = e1.detach().numpy(), e2.detach().numpy()
    out["vic_terms"] = torch.stack(terms).detach().numpy()
    out["vic_dx1"], out["vic_dx2"] = e1.grad.numpy(), e2.grad.numpy()
    # one training step
    adn1 = get_adn_fn(1, "layer", "gelu", 0.0)
    kw = {k: dict(v) for k, v in SSL_CASE.items()}
    kw["projection_head_args"]["adn_fn"] = adn1
    kw["prediction_head_args"]["adn_fn"] = adn1
    net = ConvNeXt(**kw)
    net.load_state_dict(fill_state_dict(net.state_dict(), gain=SSL_GAIN))
    net.train()
    # adell_mri/utils/utils.py imports monai (absent here), so the EMA arithmetic of
    # utils.py:484-487 is restated: shadow.sub_((1 - decay) * (shadow - param))
    shadow = {k: p.detach().clone() for k, p in net.named_parameters()}
    # structured volumes (a different spatial pattern per sample) so that the embeddings
    # differ across the batch; pure noise is averaged away by the 4^3 stem
    zz, yy, xx = torch.meshgrid(*[torch.arange(32.0)] * 3, indexing="ij")
    x1 = torch.stack([torch.sin((b + 1) * 0.35 * zz) * torch.cos((b + 2) * 0.23 * yy)
                      + 0.03 * (b - 1.5) * xx for b in range(4)])[:, None]
    x1 = x1 + 0.2 * torch.rand(x1.shape, generator=g)
    x2 = (x1 + 0.3 * torch.randn(x1.shape, generator=g)).flip(2)
    out["x1"], out["x2"] = x1.numpy(), x2.numpy()
    out["representation"] = net(x1, ret="representation").detach().numpy()
    y1 = net(x1, ret="prediction")
    y2 = net(x2, ret="projection")
    out["y1"], out["y2"] = y1.detach().numpy(), y2.detach().numpy()
    losses = crit(y1, y2)
    loss = sum(losses)
    out["losses"] = torch.stack(losses).detach().numpy()
    out["loss"] = loss.detach().numpy()
    loss.backward()
    for k, p in net.named_parameters():
        out["grad:" + k] = p.grad.numpy().copy()
    decay, no_decay = [], []
    for k, p in net.named_parameters():
        (no_decay if "normalization" in k else decay).append(p)
    opt = torch.optim.AdamW(decay + no_decay, **SSL_OPT)
    opt.step()
    for k, p in net.named_parameters():
        out["step1:" + k] = p.detach().numpy().copy()
        shadow[k].sub_((1 - 0.99) * (shadow[k] - p.detach()))
        out["ema1:" + k] = shadow[k].numpy().copy()
    out["param_keys"] = np.array([k for k, _ in net.named_parameters()])
    np.savez_compressed(os.path.join(OUT, "ssl_convnext_small.npz"), **out)
    print("ssl ok: loss", float(loss.detach()), "terms", [float(t.detach()) for t in losses],
          "y1 std over batch", float(y1.detach().std(0).mean()))

FULL_CFG2 = dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
                 upscale_type="transpose", norm_type="instance", padding=1, dropout_param=0.15,
                 activation_fn="swish", in_channels=2, n_classes=2, depth=[32, 32, 64, 128, 256],
                 kernel_sizes=[3] * 5, strides=[2] * 5)
FULL_CASES = {
    # BASELINE config 2 at the size bench.py runs ("Full-size configs: only
    # statistics + 64 sampled voxels"). name: (kwargs, input shape, with gradients?)
    "unet3d_cfg2_full": (FULL_CFG2, (1, 2, 128, 128, 128), True),
    # north_star's 256x256x128 variant: forward only (the eager reference's backward at this
    # size did not finish in 15 minutes / 2 CPU-hours in the build container and was stopped)
    "unet3d_cfg2_full_256x256x128": (FULL_CFG2, (1, 2, 256, 256, 128), False),
    # BASELINE config 2b (SURVEY.md 8(a) row a12, 8(d) "measure it second"): the same YAML with the
    # ResNet of sample_configs/ssl-resnet.yaml:5-6 as encoder, assembled as
    # entrypoints/segmentation/train.py:672-734 does (make_backbone_unet): 41.8 M parameters, odd
    # 65^3 / 33x33x65 / 17x17x65 / 9x9x33 maps from the padded max-pools, crop_to_size live in every
    # decoder level; logits, loss and every parameter gradient at 1 x 2 x 128^3
    "unet3d_cfg2b_full": (dict({k: v for k, v in FULL_CFG2.items()
                                if k not in ("depth", "kernel_sizes", "strides")},
                               _cls="backbone",
                               _structure=[[64, 64, 5, 2], [128, 128, 3, 2], [256, 256, 3, 2],
                                           [512, 512, 3, 2]],
                               _maxpool=[[2, 2, 1], [2, 2, 1], [2, 2, 2], [2, 2, 2]]),
                          (1, 2, 128, 128, 128), True),
}
from oracle.fullsize import FULL_SEED, full_inputs, sample_positions, zlib_crc  # noqa: E402


def gen_full(name, kw, shape, with_grads, net=None):
    """Reference UNet.forward (unet.py:751-843) + dice/focal + backward at the benchmark's size,
    eval() (dropout off; instance norm is mode-independent). Stores statistics and sampled values
    only: logits mean/std/min/max + 64 sampled voxels + a 128-entry line, loss terms, and for
    every parameter the gradient's L2 norm, absolute maximum and 16 sampled entries."""
    import time
    torch.manual_seed(0)
    x, y = full_inputs(shape)
    net = (make_unet(kw) if net is None else net).eval()
    out = {"seed": np.array(FULL_SEED), "shape": np.array(shape),
           "x_checksum": np.array([float(x.double().sum()), float(y.double().sum())])}
    t0 = time.time()
    if with_grads:
        logits = net(x, return_logits=True)[0]
    else:
        with torch.no_grad():
            logits = net(x, return_logits=True)[0]
    print(name, "forward", round(time.time() - t0, 1), "s")
    lg = logits.detach()
    flat = lg.reshape(-1)
    pos = sample_positions(flat.numel(), 64, 1)
    out["logit_stats"] = np.array([float(lg.double().mean()), float(lg.double().std()),
                                   float(lg.min()), float(lg.max()),
                                   float(lg.double().abs().mean())])
    out["logit_pos"], out["logit_val"] = pos, flat[pos].numpy().copy()
    # one full line through the middle of the volume and the eight corners (edge handling)
    mid = [s // 2 for s in shape[2:]]
    out["logit_line"] = lg[0, 0, mid[0], mid[1], :].numpy().copy()
    out["logit_corners"] = lg[0, 0][::shape[2] - 1, ::shape[3] - 1, ::shape[4] - 1].numpy().copy()
    prob = torch.sigmoid(logits)
    d = binary_generalized_dice_loss(prob, y, smooth=1e-5, eps=1e-6)
    f = binary_focal_loss(prob, y, gamma=1.0, eps=1e-6)
    loss = torch.stack([d.mean(), f.mean()]).mean()
    out["dice"], out["focal"] = d.detach().numpy(), f.detach().numpy()
    out["loss"] = loss.detach().numpy()
    if with_grads:
        # the reference's own probability output (final_layer's Sigmoid, unet.py:641-655) equals
        # sigmoid(logits); the loss is taken on it as pl.py:284-317 does
        t0 = time.time()
        net.zero_grad()
        loss.backward()
        print(name, "backward", round(time.time() - t0, 1), "s")
        keys = []
        for k, p in net.named_parameters():
            if p.grad is None:
                continue
            gflat = p.grad.reshape(-1)
            gp = sample_positions(gflat.numel(), 16, zlib_crc(k))
            out["gnorm:" + k] = np.array([float(gflat.double().norm()), float(gflat.abs().max())])
            out["gpos:" + k], out["gval:" + k] = gp, gflat[gp].numpy().copy()
            keys.append(k)
        out["grad_keys"] = np.array(keys)
    out["param_keys"] = np.array([k for k, _ in net.named_parameters()])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "params", sum(p.numel() for p in net.parameters()), "loss", float(loss),
          "logit stats", out["logit_stats"])


# BASELINE configs 3 and 5 at their full sizes (round 4): the REAL reference classes built with the
# keyword arguments the factory splats (as gen_surface does), name-keyed weights, the gen_full record.
FULL_SEG_CASES = {
    # name: (net_type, sample YAML, image size, image keys, input shape, gradients?, patch override)
    "unetr_cfg3_full": ("unetr", "unetr.yaml", [96, 96, 96], 1, (1, 1, 96, 96, 96), True,
                        [16, 16, 16]),
    # forward only: the eager reference's backward at this size does not fit the build container
    "swinunet_cfg5_full": ("swin", "unet-swin.yaml", [256, 256, 128], 2, (1, 2, 256, 256, 128),
                           False, None),
}


def reference_seg_network(net_type, fname, size, n_keys, patch=None):
    import yaml
    classes = {"unet": UNet, "unetpp": UNetPlusPlus, "unetr": UNETR, "swin": SWINUNet}
    with open(os.path.join(REF, "sample_configs", fname)) as fh:
        cfg = yaml.safe_load(fh)
    cfg["in_channels"] = n_keys * cfg.pop("n_channels")
    cfg["activation_fn"] = activation_factory[cfg["activation_fn"]]
    cfg.setdefault("spatial_dimensions", 3)
    for k in ("learning_rate", "batch_size", "weight_decay", "loss_fn"):
        cfg.pop(k)
    model_kw = dict(n_classes=2, bottleneck_classification=False, skip_conditioning=0,
                    feature_conditioning=0, feature_conditioning_params=None,
                    deep_supervision=False)
    cfg["image_size"] = size[:cfg["spatial_dimensions"]]
    if net_type == "unetr":
        cfg["patch_size"] = (patch or cfg["patch_size"])[:cfg["spatial_dimensions"]]
    torch.manual_seed(0)
    net = classes[net_type](**model_kw, **cfg)
    net.load_state_dict(fill_state_dict(net.state_dict()))
    return net


def gen_full_seg(name):
    net_type, fname, size, n_keys, shape, wg, patch = FULL_SEG_CASES[name]
    gen_full(name, None, shape, wg, net=reference_seg_network(net_type, fname, size, n_keys, patch))


def gen_full_convnext():
    """BASELINE config 4 at full width: the reference ConvNeXt of configs/ssl-3d-convnext.yaml (= its
    sample_configs/ssl-2d-convnext.yaml lifted to three dimensions) on four 64^3
    crops per view: representation, both heads, the three VICReg terms and per parameter the
    gradient norm / maximum / 16 sampled entries."""
    import time
    import yaml
    with open(os.path.join(ROOT, "configs", "ssl-3d-convnext.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    adn3 = get_adn_fn(3, cfg["norm_fn"], cfg["act_fn"], 0.0)
    adn1 = get_adn_fn(1, cfg["norm_fn"], cfg["act_fn"], 0.0)
    bb = {k: v for k, v in cfg["backbone_args"].items() if k != "res_type"}
    bb["adn_fn"] = adn3
    kw = dict(backbone_args=bb,
              projection_head_args=dict(cfg["projection_head_args"], adn_fn=adn1),
              prediction_head_args=dict(cfg["prediction_head_args"], adn_fn=adn1))
    torch.manual_seed(0)
    net = ConvNeXt(**kw)
    # norm scales centred at 1: a NON-collapsed state (batch std of y1 0.15, covariance term 0.03;
    # with the plain filler: 0.004 and 5e-8, i.e. the covariance term and its gradient untested)
    net.load_state_dict(fill_state_dict(net.state_dict(), gain=SSL_GAIN, norm_weight_offset=1.0))
    net.train()
    zz, yy, xx = torch.meshgrid(*[torch.arange(64.0)] * 3, indexing="ij")
    g = torch.Generator().manual_seed(FULL_SEED)
    x1 = torch.stack([torch.sin((b + 1) * 0.21 * zz) * torch.cos((b + 2) * 0.13 * yy)
                      + 0.02 * (b - 1.5) * xx for b in range(4)])[:, None]
    x1 = x1 + 0.2 * torch.rand(x1.shape, generator=g)
    x2 = (x1 + 0.3 * torch.randn(x1.shape, generator=g)).flip(2)
    out = {"x_checksum": np.array([float(x1.double().sum()), float(x2.double().sum())])}
    t0 = time.time()
    with torch.no_grad():
        out["representation"] = net(x1, ret="representation").numpy()
    y1 = net(x1, ret="prediction")
    y2 = net(x2, ret="projection")
    out["y1"], out["y2"] = y1.detach().numpy(), y2.detach().numpy()
    losses = VICRegLoss()(y1, y2)
    loss = sum(losses)
    out["losses"] = torch.stack(losses).detach().numpy()
    loss.backward()
    print("convnext_cfg4_full fwd+bwd", round(time.time() - t0, 1), "s")
    keys = []
    for k, p in net.named_parameters():
        if p.grad is None:
            continue
        gflat = p.grad.reshape(-1)
        gp = sample_positions(gflat.numel(), 16, zlib_crc(k))
        out["gnorm:" + k] = np.array([float(gflat.double().norm()), float(gflat.abs().max())])
        out["gpos:" + k], out["gval:" + k] = gp, gflat[gp].numpy().copy()
        keys.append(k)
    out["grad_keys"] = np.array(keys)
    out["param_keys"] = np.array([k for k, _ in net.named_parameters()])
    np.savez_compressed(os.path.join(OUT, "convnext_cfg4_full.npz"), **out)
    print("convnext_cfg4_full params", sum(p.numel() for p in net.parameters()), "losses",
          out["losses"], "y1 std over batch", float(y1.detach().std(0).mean()))


def gen_full_fp64(name, kw, shape):
    """The same network, inputs and loss as gen_full in DOUBLE precision: the yardstick that tells
    the reference's own fp32 summation noise (over 2 M voxels per channel) from an error of the HIP
    path. Writes <name>_fp64.npz: per parameter the gradient's L2 norm / absolute maximum and the
    entries at the SAME sampled positions as the fp32 fixture, plus the logit statistics."""
    import time
    torch.manual_seed(0)
    x, y = full_inputs(shape)
    net = make_unet(kw).eval().double()
    t0 = time.time()
    logits = net(x.double(), return_logits=True)[0]
    print(name, "fp64 forward", round(time.time() - t0, 1), "s")
    lg = logits.detach()
    flat = lg.reshape(-1)
    pos = sample_positions(flat.numel(), 64, 1)
    out = {"logit_stats": np.array([float(lg.mean()), float(lg.std()), float(lg.min()),
                                    float(lg.max()), float(lg.abs().mean())]),
           "logit_pos": pos, "logit_val": flat[pos].numpy().copy()}
    prob = torch.sigmoid(logits)
    d = binary_generalized_dice_loss(prob, y.double(), smooth=1e-5, eps=1e-6)
    f = binary_focal_loss(prob, y.double(), gamma=1.0, eps=1e-6)
    loss = torch.stack([d.mean(), f.mean()]).mean()
    out["loss"] = loss.detach().numpy()
    t0 = time.time()
    loss.backward()
    print(name, "fp64 backward", round(time.time() - t0, 1), "s")
    keys = []
    for k, p in net.named_parameters():
        if p.grad is None:
            continue
        gflat = p.grad.reshape(-1)
        gp = sample_positions(gflat.numel(), 16, zlib_crc(k))
        out["gnorm:" + k] = np.array([float(gflat.norm()), float(gflat.abs().max())])
        out["gpos:" + k], out["gval:" + k] = gp, gflat[gp].numpy().copy()
        keys.append(k)
    out["grad_keys"] = np.array(keys)
    np.savez_compressed(os.path.join(OUT, name + "_fp64.npz"), **out)
    print(name, "fp64 loss", float(loss))


SURFACE_CASES = {
    # name: (net_type, sample YAML, image size the entrypoint would pass, number of image keys)
    "unet": ("unet", "u-net-3d-resnet.yaml", None, 2),
    "unetpp": ("unetpp", "u-net-3d-resnet.yaml", None, 2),
    "unetr": ("unetr", "unetr.yaml", [96, 96, 96], 1),
    "swin": ("swin", "unet-swin.yaml", [256, 256, 128], 2),
}


def gen_surface():
    """What ``get_segmentation_network`` (network_factories.py:493-701) makes of every sample
    YAML: the factory itself imports Lightning (absent), so the reference's *network classes*
    are built with the keyword arguments the factory would splat -- the YAML through
    ``parse_config_unet``'s documented steps (config_parsing.py:30-58; ``n_channels`` read as
    ``in_channels``) plus the factory's model-side boilerplate (:587-604) -- and the
    state_dict key / shape lists are stored. tests/test_config_surface.py compares the HIP
    package's factory output with them."""
    import json
    import yaml
    classes = {"unet": UNet, "unetpp": UNetPlusPlus, "unetr": UNETR, "swin": SWINUNet}
    out = {}
    for name, (net_type, fname, size, n_keys) in SURFACE_CASES.items():
        with open(os.path.join(REF, "sample_configs", fname)) as fh:
            cfg = yaml.safe_load(fh)
        cfg["in_channels"] = n_keys * cfg.pop("n_channels")
        cfg["activation_fn"] = activation_factory[cfg["activation_fn"]]
        cfg.setdefault("spatial_dimensions", 3)
        train_only = {k: cfg.pop(k) for k in ("learning_rate", "batch_size", "weight_decay",
                                              "loss_fn")}
        model_kw = dict(n_classes=2, bottleneck_classification=False, skip_conditioning=0,
                        feature_conditioning=0, feature_conditioning_params=None)
        if net_type != "unetpp":
            model_kw["deep_supervision"] = False
        if net_type in ("unet", "unetpp"):
            model_kw["encoding_operations"] = None
        if net_type in ("unetr", "swin"):
            cfg["image_size"] = size[:cfg["spatial_dimensions"]]
        if net_type == "unetr":
            cfg["patch_size"] = cfg["patch_size"][:cfg["spatial_dimensions"]]
        torch.manual_seed(0)
        net = classes[net_type](**model_kw, **cfg)
        sd = net.state_dict()
        out[name] = {"yaml": fname, "n_keys": n_keys, "size": size,
                     "train_only": {k: v for k, v in train_only.items() if k != "loss_fn"},
                     "loss_fn": train_only["loss_fn"],
                     "n_parameters": sum(p.numel() for p in net.parameters()),
                     "state_dict": [[k, list(v.shape)] for k, v in sd.items()]}
        print(name, out[name]["n_parameters"], len(sd))
    with open(os.path.join(OUT, "factory_surface.json"), "w") as fh:
        json.dump(out, fh, indent=0)

from oracle.make_golden_cases import LOSS_CASES  # noqa: E402


def gen_losses():
    """The loss_factory members beyond binary dice / focal (utils/utils.py:39-59), values and
    gradients with respect to the probabilities, from the reference's own functions."""
    from adell_mri.modules.segmentation import losses as L
    g = torch.Generator().manual_seed(77)
    out = {}
    logits = torch.randn((2, 3, 6, 7, 5), generator=g) * 2
    cls = torch.randint(0, 3, (2, 6, 7, 5), generator=g)
    onehot = torch.nn.functional.one_hot(cls, 3).permute(0, 4, 1, 2, 3).float()
    pb = torch.sigmoid(torch.randn((2, 1, 6, 7, 5), generator=g) * 2)
    tb = (torch.rand((2, 1, 6, 7, 5), generator=g) > 0.7).float()
    r = torch.rand((2,), generator=g) + 0.5
    out.update(logits=logits.numpy(), cls=cls.numpy(), pb=pb.numpy(), tb=tb.numpy(), r=r.numpy())
    for name, (fn, kw, kind) in LOSS_CASES.items():
        if kind == "binary":
            p, t = pb.clone().requires_grad_(True), tb
        else:
            p = torch.softmax(logits, 1).detach().requires_grad_(True)
            t = onehot if kind == "onehot" else cls
        import copy
        val = getattr(L, fn)(p, t, **copy.deepcopy(kw))   # (the hybrid losses write into their dicts)
        (val * r).sum().backward()
        out[name + ":value"], out[name + ":grad"] = val.detach().numpy(), p.grad.numpy().copy()
        print(name, val.detach().numpy())
    np.savez_compressed(os.path.join(OUT, "losses_mc.npz"), **out)


SEMISL_CASES = {
    # UNetSemiSL (semi_supervised_segmentation/unet.py) in the shape of BASELINE config 2
    "unet3d_semisl": (dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
                           upscale_type="transpose", norm_type="instance", padding=1,
                           dropout_param=0.0, activation_fn="swish", in_channels=2, n_classes=2,
                           depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
                      (2, 2, 16, 16, 16)),
    "unet2d_semisl": (dict(spatial_dimensions=2, conv_type="regular", link_type="identity",
                           upscale_type="transpose", norm_type="instance", padding=1,
                           dropout_param=0.0, activation_fn="swish", in_channels=1, n_classes=2,
                           depth=[8, 16], kernel_sizes=[3] * 2, strides=[2] * 2),
                      (3, 1, 24, 32)),
}


def gen_semisl():
    """UNetContrastiveSemiSL.training_step arithmetic (semi_supervised_segmentation/pl.py:371-450)
    from the reference's own UNetSemiSL and LocalContrastiveLoss: supervised dice + focal on the
    annotated batch plus 0.01 * mean(LoCo(features(x_1), linear_transformation(features(x_2))))
    with the stop-gradient teacher (ema=None, stop_gradient=True); and the loss alone on random
    features, with the gradient for both arguments."""
    from adell_mri.modules.semi_supervised_segmentation.losses import LocalContrastiveLoss
    from adell_mri.modules.semi_supervised_segmentation.unet import UNetSemiSL

    for name, (kw, shape) in SEMISL_CASES.items():
        torch.manual_seed(0)
        g = torch.Generator().manual_seed(4321)
        kw = dict(kw)
        kw["activation_fn"] = activation_factory[kw["activation_fn"]]
        net = UNetSemiSL(**kw)
        net.load_state_dict(fill_state_dict(net.state_dict()))
        net.eval()
        x, x1, x2 = (torch.rand(shape, generator=g) for _ in range(3))
        y = (torch.rand((shape[0], 1, *shape[2:]), generator=g) > 0.9).float()
        out = {"x": x.numpy(), "x1": x1.numpy(), "x2": x2.numpy(), "y": y.numpy()}
        prob = net(x)[0]
        d = binary_generalized_dice_loss(prob, y, smooth=1e-5, eps=1e-6)
        f = binary_focal_loss(prob, y, gamma=1.0, eps=1e-6)
        sup = torch.stack([d.mean(), f.mean()]).mean()
        f1 = net.forward_features(X=x1)
        with torch.no_grad():
            f2 = net.forward_features(X=x2, apply_linear_transformation=True)
        loco = LocalContrastiveLoss(seed=42)(f1, f2)
        ssl = loco.mean() * 0.01
        total = sup + ssl
        net.zero_grad()
        total.backward()
        out.update(prob=prob.detach().numpy(), features_1=f1.detach().numpy(),
                   features_2=f2.numpy(), loco=loco.detach().numpy(),
                   sup=sup.detach().numpy(), ssl=ssl.detach().numpy(),
                   total=total.detach().numpy())
        for k, p in net.named_parameters():
            if p.grad is not None:
                out["grad:" + k] = p.grad.numpy().copy()
        out["param_keys"] = np.array([k for k, _ in net.named_parameters()])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "sup", float(sup), "ssl", float(ssl), "loco", loco.detach().numpy())
    # the loss on its own
    g = torch.Generator().manual_seed(99)
    out = {}
    for tag, shape, temp in (("a", (3, 8, 4, 5, 6), 0.1), ("b", (2, 32, 6, 6, 6), 0.1),
                             ("c", (4, 12, 7, 9), 0.5), ("d", (1, 4, 3, 3, 3), 0.1)):
        a = (torch.randn(shape, generator=g) * 2).requires_grad_(True)
        b = (torch.randn(shape, generator=g) + 0.3 * a.detach()).requires_grad_(True)
        if tag == "a":   # a voxel whose features are all zero: the clamp of cosine_similarity
            with torch.no_grad():
                a[0, :, 0, 0, 0] = 0.0
        r = torch.rand((shape[0],), generator=g) + 0.5
        val = LocalContrastiveLoss(temperature=temp)(a, b)
        (val * r).sum().backward()
        out.update({f"{tag}:x1": a.detach().numpy(), f"{tag}:x2": b.detach().numpy(),
                    f"{tag}:r": r.numpy(), f"{tag}:temperature": np.float32(temp),
                    f"{tag}:value": val.detach().numpy(), f"{tag}:grad1": a.grad.numpy().copy(),
                    f"{tag}:grad2": b.grad.numpy().copy()})
        print("loco", tag, val.detach().numpy())
    np.savez_compressed(os.path.join(OUT, "loco_loss.npz"), **out)


SSL2D_CASE = dict(
    # the reference's sample_configs/ssl-2d-convnext.yaml in miniature (spatial_dim 2)
    backbone_args=dict(spatial_dim=2, in_channels=1, structure=[[8, 16, 7, 2], [16, 32, 3, 2]],
                       maxpool_structure=[[2, 2], [2, 2]], first_layer_stride=4),
    projection_head_args=dict(in_channels=16, structure=[32, 24]),
    prediction_head_args=dict(in_channels=24, structure=[32, 24]))


def gen_ssl2d():
    """2-D ConvNeXt (conv_next.py:86-235 with spatial_dim=2, ConvNeXtBlock2d res_blocks.py:429-513):
    a block on its own (output, input gradient, parameter gradients) and the network's three heads
    with the VICReg loss and every parameter gradient."""
    from adell_mri.modules.layers.res_blocks import ConvNeXtBlock2d
    g = torch.Generator().manual_seed(21)
    out = {}
    x = torch.randn((2, 8, 10, 12), generator=g)
    out["blk_x"] = x.numpy()
    for tag, (k, oc) in {"k3": (3, 8), "k7": (7, 12)}.items():
        blk = ConvNeXtBlock2d(8, k, 16, oc)
        blk.load_state_dict(fill_state_dict(blk.state_dict()))
        xin = x.clone().requires_grad_(True)
        y = blk(xin)
        r = torch.randn(y.shape, generator=g)
        (y * r).sum().backward()
        out[f"blk_{tag}_y"], out[f"blk_{tag}_r"] = y.detach().numpy(), r.numpy()
        out[f"blk_{tag}_dx"] = xin.grad.numpy()
        for n, p in blk.named_parameters():
            out[f"blk_{tag}_grad:{n}"] = p.grad.numpy().copy()
    adn1 = get_adn_fn(1, "layer", "gelu", 0.0)
    kw = {k: dict(v) for k, v in SSL2D_CASE.items()}
    kw["projection_head_args"]["adn_fn"] = adn1
    kw["prediction_head_args"]["adn_fn"] = adn1
    net = ConvNeXt(**kw)
    net.load_state_dict(fill_state_dict(net.state_dict(), gain=SSL_GAIN))
    net.train()
    yy, xx = torch.meshgrid(torch.arange(64.0), torch.arange(48.0), indexing="ij")
    x1 = torch.stack([torch.sin((b + 1) * 0.31 * yy) * torch.cos((b + 2) * 0.17 * xx)
                      + 0.02 * (b - 1.5) * xx for b in range(4)])[:, None]
    x1 = x1 + 0.2 * torch.rand(x1.shape, generator=g)
    x2 = (x1 + 0.3 * torch.randn(x1.shape, generator=g)).flip(2)
    out["x1"], out["x2"] = x1.numpy(), x2.numpy()
    out["representation"] = net(x1, ret="representation").detach().numpy()
    y1, y2 = net(x1, ret="prediction"), net(x2, ret="projection")
    out["y1"], out["y2"] = y1.detach().numpy(), y2.detach().numpy()
    losses = VICRegLoss()(y1, y2)
    sum(losses).backward()
    out["losses"] = torch.stack(losses).detach().numpy()
    for k, p in net.named_parameters():
        out["grad:" + k] = p.grad.numpy().copy()
    out["param_keys"] = np.array([k for k, _ in net.named_parameters()])
    out["param_shapes"] = np.array([",".join(map(str, p.shape)) for _, p in net.named_parameters()])
    np.savez_compressed(os.path.join(OUT, "ssl_convnext2d_small.npz"), **out)
    print("ssl2d ok: terms", [float(t.detach()) for t in losses], "representation",
          tuple(out["representation"].shape))


def gen_resnet2d():
    """2-D ResNet (res_net.py:51-396 with spatial_dim=2) as the VICReg wrapper builds it from a 2-D
    backbone configuration: heads, loss terms, every parameter gradient."""
    g = torch.Generator().manual_seed(41)
    adn = get_adn_fn(2, "batch", "swish", 0.0)
    adn1 = get_adn_fn(1, "layer", "gelu", 0.0)
    net = ResNet(dict(spatial_dim=2, in_channels=1, structure=[[8, 8, 5, 2], [16, 16, 3, 2]],
                      maxpool_structure=[[2, 2], [2, 2]], res_type="resnet", adn_fn=adn),
                 dict(in_channels=16, structure=[32, 24], adn_fn=adn1),
                 dict(in_channels=24, structure=[32, 24], adn_fn=adn1))
    net.load_state_dict(fill_state_dict(net.state_dict(), gain=SSL_GAIN))
    net.train()
    yy, xx = torch.meshgrid(torch.arange(40.0), torch.arange(48.0), indexing="ij")
    x1 = torch.stack([torch.sin((b + 1) * 0.31 * yy) * torch.cos((b + 2) * 0.17 * xx)
                      + 0.02 * (b - 1.5) * xx for b in range(4)])[:, None]
    x1 = x1 + 0.2 * torch.rand(x1.shape, generator=g)
    x2 = (x1 + 0.3 * torch.randn(x1.shape, generator=g)).flip(2)
    out = {"x1": x1.numpy(), "x2": x2.numpy()}
    out["representation"] = net(x1, ret="representation").detach().numpy()
    y1, y2 = net(x1, ret="prediction"), net(x2, ret="projection")
    out["y1"], out["y2"] = y1.detach().numpy(), y2.detach().numpy()
    losses = VICRegLoss()(y1, y2)
    sum(losses).backward()
    out["losses"] = torch.stack(losses).detach().numpy()
    for k, p in net.named_parameters():
        out["grad:" + k] = p.grad.numpy().copy()
    out["param_keys"] = np.array([k for k, _ in net.named_parameters()])
    out["param_shapes"] = np.array([",".join(map(str, p.shape)) for _, p in net.named_parameters()])
    np.savez_compressed(os.path.join(OUT, "ssl_resnet2d_small.npz"), **out)
    print("resnet2d ok: terms", [float(t.detach()) for t in losses], "representation",
          tuple(out["representation"].shape))


def gen_pair_losses():
    """simsiam_loss / byol_loss / NTXentLoss (the non-VICReg choices of SelfSLBasePL.init_loss,
    self_supervised/pl.py:202-212) from the reference's own code: values and both gradients."""
    from adell_mri.modules.self_supervised.losses.functional import byol_loss, simsiam_loss
    from adell_mri.modules.self_supervised.losses.ntxent import NTXentLoss
    g = torch.Generator().manual_seed(31)
    out = {}
    cases = [("simsiam_a", "simsiam", (6, 40), 1.0, False), ("byol_a", "byol", (5, 33), 1.0, False),
             ("simsiam_zero_row", "simsiam", (4, 16), 1.0, False),
             ("ntxent_relu", "ntxent", (6, 40), 0.5, True),
             ("ntxent_norelu", "ntxent", (4, 24), 0.1, False),
             ("ntxent_t1", "ntxent", (3, 8), 1.0, True), ("byol_b1", "byol", (1, 12), 1.0, False)]
    for tag, kind, shape, temp, relu in cases:
        a = torch.randn(shape, generator=g).requires_grad_(True)
        b = (0.5 * a.detach() + 0.8 * torch.randn(shape, generator=g)).requires_grad_(True)
        if tag == "simsiam_zero_row":
            with torch.no_grad():
                a[1] = 0.0   # the clamp of cosine_similarity
        fn = {"simsiam": simsiam_loss, "byol": byol_loss,
              "ntxent": NTXentLoss(temperature=temp, apply_relu=relu)}[kind]
        val = fn(a, b)
        val.backward()
        out.update({f"{tag}:kind": np.array(kind), f"{tag}:temperature": np.float32(temp),
                    f"{tag}:relu": np.array(relu), f"{tag}:x1": a.detach().numpy(),
                    f"{tag}:x2": b.detach().numpy(), f"{tag}:value": val.detach().numpy(),
                    f"{tag}:grad1": a.grad.numpy().copy(), f"{tag}:grad2": b.grad.numpy().copy()})
        print(tag, float(val))
    np.savez_compressed(os.path.join(OUT, "ssl_pair_losses.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "resnet2d":
        gen_resnet2d()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pairloss":
        gen_pair_losses()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ssl2d":
        gen_ssl2d()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "semisl":
        gen_semisl()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "losses":
        gen_losses()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "unetr_fc":
        gen_unet("unetr3d_feature_cond", *UNETR_CASES["unetr3d_feature_cond"])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "twod":
        gen_unet("unetr2d_small", *UNETR_CASES["unetr2d_small"])
        gen_unet("unetpp2d_small", *UNETPP_CASES["unetpp2d_small"])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "surface":
        gen_surface()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "full64":
        gen_full_fp64("unet3d_cfg2_full", *FULL_CASES["unet3d_cfg2_full"][:2])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fullseg":
        for name in FULL_SEG_CASES:
            if len(sys.argv) > 2 and sys.argv[2] != name:
                continue
            gen_full_seg(name)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fullssl":
        gen_full_convnext()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        for name, (kw, shape, wg) in FULL_CASES.items():
            if len(sys.argv) > 2 and sys.argv[2] != name:
                continue
            gen_full(name, kw, shape, wg)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ssl":
        gen_ssl()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vit":
        gen_vit_tokens()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "depthwise":
        for name, (kw, shape, dist) in DEPTHWISE_CASES.items():
            gen_unet(name, kw, shape, dist)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "unet2d":
        for name, (kw, shape, dist) in UNET2D_CASES.items():
            gen_unet(name, kw, shape, dist)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "brunet":
        for name, (kw, shape, missing) in BRUNET_CASES.items():
            gen_brunet(name, kw, shape, missing)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sae":
        for name, (kw, shape, dist) in SAE_CASES.items():
            gen_unet(name, kw, shape, dist)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "asp":
        for name, (kw, shape, dist) in ASP_CASES.items():
            gen_unet(name, kw, shape, dist)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attention":
        for name, (kw, shape, dist) in ATTENTION_LINK_CASES.items():
            gen_unet(name, kw, shape, dist)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "swin":
        for name, (kw, shape, dist) in SWIN_CASES.items():
            if len(sys.argv) > 2 and sys.argv[2] != name:
                continue
            gen_unet(name, kw, shape, dist)
        sys.exit(0)
    for name, (kw, shape, dist) in {**UNET_CASES, **UNET2D_CASES, **UNETR_CASES, **UNETPP_CASES,
                                    **BACKBONE_CASES, **SWIN_CASES}.items():
        gen_unet(name, kw, shape, dist)
    for name, (kw, shape, missing) in BRUNET_CASES.items():
        gen_brunet(name, kw, shape, missing)
    gen_blocks()
    gen_ssl()
